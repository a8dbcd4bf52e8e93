#!/bin/bash
# same-box A/B of two prebuilt libraries (1xgpt_amd/lib_ab_old.so / lib_ab_new.so) on the MAGVIT2 conv stack: correctness with the
# new one, then per-shape conv rates and config 5 (encode -> sample -> decode) with each, interleaved
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/${1:-r03}_conv_ab.txt; : > $OUT
GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_new.so python -m pytest tests/test_hip_harness.py tests/test_hip_configs.py -m gpu -x -q -k "conv or decoder or encoder or config5 or tokenizer" 2>&1 | tail -2 >> $OUT
for rep in 1 2; do for v in old new; do
  echo "== $v (rep $rep)" >> $OUT
  GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_$v.so python tools/bench_conv.py --frames 16 2>/dev/null | tail -14 >> $OUT
  GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_$v.so python tools/bench_e2e.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('e2e', round(d['encode_frames_per_sec']), round(d['decode_frames_per_sec']), round(d['end_to_end_generated_frames_per_sec'],1), 'enc/dec TF', round(d['encode_tflops']), round(d['decode_tflops']))" >> $OUT
done; done
cat $OUT
