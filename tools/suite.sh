mkdir -p gpurun_out/r4
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r4/gpu_suite.log 2>&1; tail -8 gpurun_out/r4/gpu_suite.log
timeout -k 10 600 python bench.py --no-train --no-train-files > gpurun_out/r4/bench_split.json 2> gpurun_out/r4/bench_split.err; tail -c 1500 gpurun_out/r4/bench_split.json
