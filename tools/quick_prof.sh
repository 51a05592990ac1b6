# kernel-trace stats of the refinement loop only (no training, no variants): gpurun_out/r4/quick_kernel_stats.csv
set -e
R=$GRAFT_REPO_ROOT/gpurun_out/r4
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/prof_quick -o q --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-train --no-train-files --no-fresh-batch --no-variants --head-epochs 0 --parity-pairs 0 --steps 20 --warmup 3 > $R/quick_bench.json 2> $R/quick_prof.err
cp $(find $R/prof_quick -name "*kernel_stats.csv" | head -1) $R/quick_kernel_stats.csv
rm -rf $R/prof_quick
head -16 $R/quick_kernel_stats.csv | cut -c1-150
