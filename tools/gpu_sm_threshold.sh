#!/bin/bash
# which kernel should take the narrow one-frame-pass GEMMs (N = 512: proj, fc2) at 1,024-4,096 rows: gemm16_sm (in-workgroup split-K,
# operands straight from L2) or gemm16_nt (128x128 LDS tiles)?  study build: GENIE_GEMM16_SM_MAX = largest M*N gemm16_sm accepts
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/${1:-r03}_sm_threshold.txt; : > $OUT
export GENIE_HIP_LIBRARY=$R/1xgpt_amd/libgenie_hip_study.so
for mx in 1048576 1048575 524287 262143; do
  for prec in f16x3 bf16; do
    for acc in 0 1; do
      echo "== GENIE_GEMM16_SM_MAX=$mx $prec acc=$acc" >> $OUT
      GENIE_GEMM16_SM_MAX=$mx python tools/bench_gemm_small.py --prec $prec --clips 2 4 8 --acc $acc 2>/dev/null | grep -E "proj|fc2" >> $OUT
    done
  done
done
cat $OUT
