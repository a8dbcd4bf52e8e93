#!/bin/bash
# same-box A/B of the mid-size 16-bit GEMMs' epilogue (gemm16_v2 / gemm16_nt; prebuilt 1xgpt_amd/lib_ab_old.so / lib_ab_new.so):
# tests with the new library, one-frame-pass GEMM sizes with and without the residual accumulate, generate, one training step;
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/${1:-r03}_midsize_epi_ab.txt; : > $OUT
GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_new.so python -m pytest tests/test_hip_bf16.py tests/test_hip_f16x3.py tests/test_hip_configs.py tests/test_hip_train.py tests/test_hip_harness.py -m gpu -x -q 2>&1 | tail -3 >> $OUT
for rep in 1 2; do
 for v in old new; do
  export GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_$v.so
  for prec in f16x3 bf16; do
  echo "== $v $prec (rep $rep)" >> $OUT
  python tools/bench_gemm_small.py --prec $prec --clips 8 16 32 2>/dev/null | grep TFLOP >> $OUT
  echo "== $v $prec residual accumulate (rep $rep)" >> $OUT
  python tools/bench_gemm_small.py --prec $prec --clips 8 16 32 --acc 1 2>/dev/null | grep TFLOP >> $OUT
  done
 done
done
for v in old new old new; do
  export GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_$v.so
  echo "== $v generate" >> $OUT
  python tools/bench_generate.py --batches 1 8 16 --steps 2 2>/dev/null | grep "^{'schedule" >> $OUT
  echo "== $v train bf16 8 clips" >> $OUT
  python tools/bench_train.py --precision bf16 --batch 8 --steps 3 2>/dev/null | tail -1 >> $OUT
done
unset GENIE_HIP_LIBRARY
tail -30 $OUT
