#!/bin/bash
# same-box A/B of the 2-way K split of fc2 in one-frame passes at 2,048-4,096 rows (lib_ab_old.so: one 128x128-tile GEMM with the fused
# residual epilogue; lib_ab_new.so: two K halves into slabs + ordered combine): tests with the new library, generate, config 5
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/${1:-r03}_fc2_splitk_ab.txt; : > $OUT
GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_new.so python -m pytest tests/test_hip_prefix_reuse.py tests/test_hip_configs.py tests/test_hip_f16x3.py tests/test_hip_bf16.py tests/test_hip_parity.py tests/test_hip_harness.py -m gpu -x -q 2>&1 | tail -2 >> $OUT
for v in old new old new; do
  export GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_$v.so
  echo "== $v" >> $OUT
  python tools/bench_generate.py --batches 1 8 12 16 --steps 2 --schedules kv_cache 2>/dev/null | grep "^{'schedule" | cut -c1-140 >> $OUT
  python tools/bench_generate.py --batches 16 --steps 2 --schedules kv_cache --precision bf16 2>/dev/null | grep "^{'schedule" | cut -c1-140 | sed 's/^/bf16 /' >> $OUT
  python tools/bench_e2e.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('e2e 8 clips: sample', round(d['generate_frames_per_sec'],1), 'end to end', round(d['end_to_end_generated_frames_per_sec'],1))" >> $OUT
done
cat $OUT
