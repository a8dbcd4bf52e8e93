#!/bin/bash
# same-box A/B of LayerNorm at one-frame-pass sizes (lib_ab_old.so = 4 rows per wave always, lib_ab_new.so = 1 row per wave up to 16,384 rows)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/${1:-r03}_ln_small_ab.txt; : > $OUT
GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_new.so python -m pytest tests/test_hip_parity.py tests/test_hip_configs.py tests/test_hip_bf16.py tests/test_hip_f16x3.py -m gpu -x -q 2>&1 | tail -2 >> $OUT
for v in old new old new; do
  export GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_$v.so
  echo "== $v" >> $OUT
  python tools/bench_generate.py --batches 4 8 16 --steps 2 --schedules kv_cache 2>/dev/null | grep "^{'schedule" | cut -c1-140 >> $OUT
done
cat $OUT
