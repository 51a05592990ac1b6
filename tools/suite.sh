mkdir -p gpurun_out/r4
timeout -k 10 100 python tools/conv1_time.py > gpurun_out/r4/conv1_split.log 2>&1; tail -4 gpurun_out/r4/conv1_split.log
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r4/gpu_suite.log 2>&1; tail -3 gpurun_out/r4/gpu_suite.log
timeout -k 10 600 python bench.py --no-train --no-train-files > gpurun_out/r4/bench_split.json 2> gpurun_out/r4/bench_split.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/bench_split.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['frac'], d.get('f32_pipe'), d['parity']['max_step_err'], d['parity']['ok'])
for k,v in d['roofline']['all_kernels'].items(): print(k[:60], v['ms_per_forward'])
PY
