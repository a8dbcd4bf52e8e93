#!/bin/bash
# De-phased CUs (GENIE_PP_STAGGER = number of phase groups; study build): GEMM microbench at the bench's M, both precisions
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
export GENIE_HIP_LIBRARY=$R/1xgpt_amd/libgenie_hip_study.so
OUT=$R/gpurun_out/${1:-r03}_stagger.txt; : > $OUT
for rep in 1 2; do
for s in 0 2 4 8; do
  echo "== GENIE_PP_STAGGER=$s (rep $rep)" >> $OUT
  GENIE_PP_STAGGER=$s GENIE_PP_STAGGER_MIN_TILES=1024 python tools/bench_gemm.py --rows 491520 --prec bf16 f16x3 --shapes 1536:512 512:512 2048:512 512:2048 2>/dev/null | grep TFLOP >> $OUT
done
done
cat $OUT
