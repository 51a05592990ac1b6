set -e
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 400 python bench.py > $R/r01_bench.json 2> $R/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/prof_r01 -o r01 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $R/r01_bench_under_rocprof.json 2> $R/prof.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $R/pmc_fetch -o f --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 > /dev/null 2> $R/pmc1.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $R/pmc_write -o w --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 > /dev/null 2> $R/pmc2.err
find $R/prof_r01 $R/pmc_fetch $R/pmc_write -name "*.csv" | head -20
