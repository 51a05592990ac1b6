mkdir -p gpurun_out/r4
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r4/gpu_suite.log 2>&1; tail -3 gpurun_out/r4/gpu_suite.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4/smoke.log 2>&1; tail -2 gpurun_out/r4/smoke.log
timeout -k 10 200 python tools/conv6_time.py > gpurun_out/r4/conv6_time.log 2>&1; tail -10 gpurun_out/r4/conv6_time.log
