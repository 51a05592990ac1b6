# SQ counters of one bf16 training iteration, one rocprofv3 --pmc pass per counter group (gpurun_out/pmc_train_*).
# usage: bash tools/pmc_train.sh [f32|bf16]
set -e
DT=${1:-bf16}
R=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp -d $R/pmc_train_$i -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py 16 $DT > $R/pmc_train_$i.log 2>&1
done
find $R/pmc_train_* -name "*counter_collection.csv"
