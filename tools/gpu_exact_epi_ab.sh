#!/bin/bash
# same-box A/B of the exact-precision GEMM's epilogue (prebuilt 1xgpt_amd/lib_ab_old.so / lib_ab_new.so): correctness tests with the
# new library, microbench of the three epilogue flavours at the reuse path's M, the exact-precision evaluate at 64 clips
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/${1:-r03}_exact_epi_ab.txt; : > $OUT
GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_new.so python -m pytest tests/test_hip_parity.py tests/test_hip_prefix_reuse.py tests/test_hip_configs.py tests/test_hip_bench_config.py tests/test_hip_train.py -m gpu -x -q 2>&1 | tail -3 >> $OUT
for rep in 1 2; do
 for v in old new; do
  export GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_$v.so
  echo "== $v plain (rep $rep)" >> $OUT
  python tools/bench_gemm.py --rows 61440 --prec exact --shapes 1536:512 1024:512 2>/dev/null | grep TFLOP >> $OUT
  echo "== $v residual accumulate (rep $rep)" >> $OUT
  python tools/bench_gemm.py --rows 61440 --prec exact --acc 1 --shapes 512:512 512:2048 2>/dev/null | grep TFLOP >> $OUT
  echo "== $v GELU (rep $rep)" >> $OUT
  python tools/bench_gemm.py --rows 61440 --prec exact --gelu 1 --shapes 2048:512 2>/dev/null | grep TFLOP >> $OUT
 done
done
for rep in 1 2; do
 for v in old new; do
  GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_$v.so python bench.py --precision exact --batch 64 --no-cpu-baseline --no-train-leg --no-secondary --no-board-sampler --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'exact evaluate rep$rep', round(d['value'],1), 'frames/s  GEMM', round(d['roofline']['achieved'],1), 'TF  frac', round(d['roofline']['frac'],3), ' ce', d['ce'])" >> $OUT
 done
done
cat $OUT
