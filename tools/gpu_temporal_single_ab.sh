#!/bin/bash
# same-box A/B of the KV-cache decode's temporal attention (lib_ab_old.so = t + 1 serial 64-lane reductions, lib_ab_new.so = all scores in
# one pass): generate / harness tests with the new library, then generate at 1 / 8 / 16 clips and config 5 end to end
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/${1:-r03}_temporal_single_ab.txt; : > $OUT
GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_new.so python -m pytest tests/test_hip_configs.py tests/test_hip_harness.py tests/test_hip_prefix_reuse.py tests/test_hip_bf16.py tests/test_hip_f16x3.py -m gpu -x -q 2>&1 | tail -2 >> $OUT
for v in old new old new; do
  export GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_$v.so
  echo "== $v" >> $OUT
  python tools/bench_generate.py --batches 1 8 16 --steps 2 --schedules kv_cache 2>/dev/null | grep "^{'schedule" >> $OUT
  python tools/bench_generate.py --batches 16 --steps 2 --schedules kv_cache --precision bf16 2>/dev/null | grep "^{'schedule" | sed 's/^/bf16 /' >> $OUT
  python tools/bench_e2e.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('e2e 8 clips: sample', round(d['generate_frames_per_sec'],1), 'end to end', round(d['end_to_end_generated_frames_per_sec'],1))" >> $OUT
done
cat $OUT
