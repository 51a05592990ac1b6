# PMC passes over ONE command (after --): fabric bytes and SQ busy / wait counters per kernel.  usage: bash tools/pmc_one.sh OUTDIR FILTER -- python3 tools/x.py ...
# (separate --pmc passes, program directly after --: the guide's rocprofv3 recipe)
set -e
OUT=$1; FLT=$2; shift 3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp -d $OUT/p$i -o p --output-format csv -- "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_kernels.py "$FLT" $(find $OUT -name "*counter_collection.csv")
