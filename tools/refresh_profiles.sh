# Regenerates the per-round evidence under gpurun_out/ on the 1-GPU box (copy what is to be judged into profiles/ afterwards):
#   bench line (un-profiled), rocprofv3 --kernel-trace --stats of the same command, two separate --pmc passes (FETCH_SIZE, WRITE_SIZE).
# usage: ROUND=r02 bash tools/refresh_profiles.sh
set -e
ROUND=${ROUND:-r04}
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 500 python bench.py > $R/${ROUND}_bench.json 2> $R/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/prof_${ROUND} -o ${ROUND} --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-train --no-fresh-batch --no-variants --head-epochs 0 --parity-pairs 0 > $R/${ROUND}_bench_under_rocprof.json 2> $R/prof.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $R/pmc_fetch -o f --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-fresh-batch --no-variants --head-epochs 0 --parity-pairs 0 --profile-steps 1 > /dev/null 2> $R/pmc1.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $R/pmc_write -o w --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-fresh-batch --no-variants --head-epochs 0 --parity-pairs 0 --profile-steps 1 > /dev/null 2> $R/pmc2.err
find $R/prof_${ROUND} $R/pmc_fetch $R/pmc_write -name "*.csv" | head -20

# per-kernel HBM traffic summary (gfx950 correction) + one bf16 training iteration in launch order
python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py $(find $R/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $R/pmc_write -name "*counter_collection.csv" | head -1) $R/${ROUND}_pmc_traffic.json > /dev/null
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/prof_${ROUND}_train -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py 16 bf16 > $R/${ROUND}_train_bf16_iteration.json 2> $R/prof_train.err
python3 $GRAFT_REPO_ROOT/tools/trace_iter.py $(find $R/prof_${ROUND}_train -name "*kernel_trace.csv" | head -1) > $R/${ROUND}_train_bf16_iteration_trace.txt
cp $(find $R/prof_${ROUND}_train -name "*kernel_stats.csv" | head -1) $R/${ROUND}_train_bf16_kernel_stats.csv
cp $(find $R/prof_${ROUND} -name "*kernel_stats.csv" | head -1) $R/${ROUND}_bench_kernel_stats.csv
cp $(find $R/pmc_fetch -name "*counter_collection.csv" | head -1) $R/${ROUND}_pmc_bench_FETCH_SIZE.csv
cp $(find $R/pmc_write -name "*counter_collection.csv" | head -1) $R/${ROUND}_pmc_bench_WRITE_SIZE.csv
rm -rf $R/prof_${ROUND} $R/prof_${ROUND}_train $R/pmc_fetch $R/pmc_write
ls -la $R/${ROUND}_*
