#!/bin/bash
# Leading-dimension study (study build): GEMM microbench at the bench's M with padded activation rows / output rows
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
export GENIE_HIP_LIBRARY=$R/1xgpt_amd/libgenie_hip_study.so
OUT=$R/gpurun_out/${1:-r03}_pad_study.txt; : > $OUT
for rep in 1 2; do
for cfg in "0 0" "64 0" "0 32" "64 32" "0 64"; do
  set -- $cfg
  echo "== pad_a=$1 pad_c=$2 (rep $rep)" >> $OUT
  python tools/bench_gemm.py --rows 491520 --prec f16x3 bf16 --pad-a $1 --pad-c $2 --shapes 1536:512 512:512 2048:512 512:2048 1024:512 2>/dev/null | grep TFLOP >> $OUT
done
done
cat $OUT
