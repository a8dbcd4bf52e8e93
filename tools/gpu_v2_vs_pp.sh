#!/bin/bash
# Study build: the multi-workgroup-per-CU kernels (gemm16_v2 256x128 / gemm16_nt 128x128, two-accumulator f16x3) against the
# persistent 256x256 gemm16_pp at the bench's M
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
export GENIE_HIP_LIBRARY=$R/1xgpt_amd/libgenie_hip_study.so
OUT=$R/gpurun_out/${1:-r03}_v2_vs_pp.txt; : > $OUT
for rep in 1 2; do
echo "== pp (rep $rep)" >> $OUT
python tools/bench_gemm.py --rows 491520 --prec f16x3 bf16 --shapes 1536:512 512:512 2048:512 512:2048 2>/dev/null | grep TFLOP >> $OUT
echo "== GENIE_GEMM16_PP=0 -> gemm16_v2 (rep $rep)" >> $OUT
GENIE_GEMM16_PP=0 python tools/bench_gemm.py --rows 491520 --prec f16x3 bf16 --shapes 1536:512 512:512 2048:512 512:2048 2>/dev/null | grep TFLOP >> $OUT
echo "== GENIE_GEMM16_PP=0 GENIE_GEMM16_V1=1 -> gemm16_nt (rep $rep)" >> $OUT
GENIE_GEMM16_PP=0 GENIE_GEMM16_V1=1 python tools/bench_gemm.py --rows 491520 --prec f16x3 bf16 --shapes 1536:512 512:512 2048:512 512:2048 2>/dev/null | grep TFLOP >> $OUT
done
cat $OUT
