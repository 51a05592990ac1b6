mkdir -p gpurun_out/r4
F="--no-train --no-train-files --no-variants --no-fresh-batch --no-cpu-baseline --parity-pairs 4 --head-epochs 0 --steps 40 --warmup 5"
for w in 1 0 1 0; do
  DIM_WINO_SPLIT=$w python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split=$w', d['value'], d['ms_per_step'], d['roofline']['achieved'], d.get('parity'))"
done > gpurun_out/r4/ab_split.log 2>&1
cat gpurun_out/r4/ab_split.log
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_refine.py tests/test_gpu_fullsize.py -x -q -k "wino or refine or batch16 or loop" > gpurun_out/r4/gpu_split_tests.log 2>&1; tail -5 gpurun_out/r4/gpu_split_tests.log
