#!/bin/bash
# same-box A/B of gemm16_sm's epilogue operands (lib_ab_old.so: residual and bias read after the reduce; lib_ab_new.so: requested at
# the top of the kernel)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/${1:-r03}_sm_prefetch_ab.txt; : > $OUT
GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_new.so python -m pytest tests/test_hip_configs.py tests/test_hip_bf16.py tests/test_hip_f16x3.py tests/test_hip_harness.py -m gpu -x -q 2>&1 | tail -2 >> $OUT
for v in old new old new; do
  export GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_$v.so
  echo "== $v" >> $OUT
  python tools/bench_gemm_small.py --prec f16x3 --clips 1 4 --acc 1 2>/dev/null | grep -E "proj|fc2" >> $OUT
  python tools/bench_gemm_small.py --prec bf16 --clips 1 4 --acc 1 2>/dev/null | grep -E "proj|fc2" >> $OUT
  python tools/bench_generate.py --batches 1 4 --steps 2 --schedules kv_cache 2>/dev/null | grep "^{'schedule" | cut -c1-140 >> $OUT
done
cat $OUT
