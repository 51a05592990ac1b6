# Regenerates the per-round evidence under gpurun_out/ on the 1-GPU box (copy what is to be judged into profiles/ afterwards):
#   bench line (un-profiled), rocprofv3 --kernel-trace --stats of the same command, two separate --pmc passes (FETCH_SIZE, WRITE_SIZE).
# usage: ROUND=r02 bash tools/refresh_profiles.sh
set -e
ROUND=${ROUND:-r02}
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 500 python bench.py > $R/${ROUND}_bench.json 2> $R/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/prof_${ROUND} -o ${ROUND} --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-train --no-fresh-batch > $R/${ROUND}_bench_under_rocprof.json 2> $R/prof.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $R/pmc_fetch -o f --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-fresh-batch --profile-steps 1 > /dev/null 2> $R/pmc1.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $R/pmc_write -o w --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-fresh-batch --profile-steps 1 > /dev/null 2> $R/pmc2.err
find $R/prof_${ROUND} $R/pmc_fetch $R/pmc_write -name "*.csv" | head -20
