mkdir -p gpurun_out/r4
F="--no-train --no-train-files --no-variants --no-fresh-batch --no-cpu-baseline --parity-pairs 0 --head-epochs 0 --steps 40 --warmup 5"
for w in 1 0 1 0 1 0; do
  DIM_WINO_WIDE_FLUSH=$w python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wide=$w', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done > gpurun_out/r4/ab_wide.log 2>&1
cat gpurun_out/r4/ab_wide.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r4/gpu_suite.log 2>&1; tail -5 gpurun_out/r4/gpu_suite.log
