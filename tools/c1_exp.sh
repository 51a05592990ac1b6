for v in 0 1 2 3 4 8 15; do echo -n "exp $v: "; DIM_HIP_LIB=$GRAFT_REPO_ROOT/gpurun_exp/libdeepim_hip_exp$v.so timeout -k 10 100 python tools/conv1_time.py 2>/dev/null | tr '\n' ' '; echo; done
