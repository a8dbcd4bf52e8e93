#!/bin/bash
# same-box A/B of the exact-precision GEMM's tile order (lib_ab_old.so = row-major tile ids, lib_ab_new.so = XCD-aware groups of
# 8 row tiles) and of the exact evaluate's batch (64 clips = 7.5 "rounds" of 1,024 resident workgroups in the N = 512 GEMMs, 128 = 15.0)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/${1:-r03}_exact_tile_order_ab.txt; : > $OUT
GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_new.so python -m pytest tests/test_hip_parity.py tests/test_hip_prefix_reuse.py tests/test_hip_bench_config.py -m gpu -x -q 2>&1 | tail -2 >> $OUT
for rep in 1 2; do
 for v in old new; do
  export GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_$v.so
  echo "== $v (rep $rep)" >> $OUT
  python tools/bench_gemm.py --rows 245760 --prec exact --shapes 1536:512 512:512 2>/dev/null | grep TFLOP >> $OUT
  python tools/bench_gemm.py --rows 245760 --prec exact --acc 1 --shapes 512:512 512:2048 2>/dev/null | grep TFLOP >> $OUT
  python tools/bench_gemm.py --rows 245760 --prec exact --gelu 1 --shapes 2048:512 2>/dev/null | grep TFLOP >> $OUT
 done
done
for b in 64 128; do
 for v in old new; do
  GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_$v.so python bench.py --precision exact --batch $b --no-cpu-baseline --no-train-leg --no-secondary --no-board-sampler --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'exact evaluate batch $b', round(d['value'],1), 'frames/s  GEMM', round(d['roofline']['achieved'],1), 'TF  frac', round(d['roofline']['frac'],3), ' ce', d['ce'])" >> $OUT
 done
done
cat $OUT
