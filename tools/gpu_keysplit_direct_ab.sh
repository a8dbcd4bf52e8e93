#!/bin/bash
# same-box A/B of the small-batch spatial attention (lib_ab_old.so = staged key-split kernel, lib_ab_new.so = attn_spatial_keysplit_kernel:
# fragments straight from the qkv rows): the tests that run one-frame passes, then generate at 1 / 2 / 4 clips
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/${1:-r03}_keysplit_direct_ab.txt; : > $OUT
GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_new.so python -m pytest tests/test_hip_prefix_reuse.py tests/test_hip_configs.py tests/test_hip_f16x3.py tests/test_hip_bf16.py tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -2 >> $OUT
for v in old new old new; do
  export GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_$v.so
  echo "== $v" >> $OUT
  python tools/bench_generate.py --batches 1 2 4 --steps 2 --schedules kv_cache 2>/dev/null | grep "^{'schedule" | cut -c1-140 >> $OUT
  python tools/bench_generate.py --batches 1 --steps 8 --schedules kv_cache 2>/dev/null | grep "^{'schedule" | cut -c1-140 >> $OUT
done
cat $OUT
