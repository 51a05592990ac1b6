#!/bin/bash
# Build one library per DIM_SPLIT_EXP value (timing experiments on wino_gemm_split_kernel; see wino_gemm_split.hip) into
# gpurun_exp/libdeepim_hip_expN.so.  Usage: tools/split_exp.sh 0 1 2 ...   then on the GPU box: python tools/split_time.py 0 1 2 ...
set -e
cd "$(dirname "$0")/../mx-deepim_amd/csrc"
make -s
mkdir -p ../../gpurun_exp
# FILE=conv.hip MACRO=DIM_C1_EXP tools/split_exp.sh 1 2 ... : the same for another source file / experiment macro
FILE=${FILE:-wino_gemm_split.hip}
MACRO=${MACRO:-DIM_SPLIT_EXP}
OTHERS=$(ls build/*.o | grep -v "/${FILE%.hip}.o")
# a value "oN" builds -DDIM_SPLIT_OPT=N (schedule options) instead of an experiment; "eXoN" both
for v in "$@"; do
  e=0; o=0
  case $v in
    e*o*) e=${v#e}; e=${e%o*}; o=${v#*o};;
    o*) o=${v#o};;
    *) e=$v;;
  esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -D$MACRO=$e -DDIM_SPLIT_OPT=$o $EXTRA -c $FILE -o /tmp/wgs_exp_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_exp/libdeepim_hip_exp$v.so $OTHERS /tmp/wgs_exp_$v.o
  echo built exp $v
done
