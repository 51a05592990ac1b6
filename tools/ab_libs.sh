# same-box A/B of two builds of the library (gpurun_exp/libdeepim_hip_exp{old,new}.so): plane GEMMs (+ correctness of new), whole loop
mkdir -p gpurun_out/r4
F="--no-train --no-train-files --no-variants --no-fresh-batch --no-cpu-baseline --parity-pairs 0 --head-epochs 0 --steps 40 --warmup 5"
DIM_HIP_LIB=$GRAFT_REPO_ROOT/gpurun_exp/libdeepim_hip_expnew.so timeout -k 10 200 python tools/split_check.py 2>/dev/null
for rep in 1 2; do for v in old new; do
  export DIM_HIP_LIB=$GRAFT_REPO_ROOT/gpurun_exp/libdeepim_hip_exp$v.so
  echo -n "$v: gemm "; timeout -k 10 100 python tools/split_time.py lib 2>/dev/null | tr '\n' ' '
  echo -n " loop "; timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
