#!/bin/bash
# up to how many (sequence, head) pairs should the direct key-split attention kernel take the one-frame passes?  (study build:
# GENIE_ATTN_KEYSPLIT_MAX_PAIRS; above it the staged kernel's query-split modes run)  8 clips = 64 pairs, 16 = 128, 32 = 256
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/${1:-r03}_keysplit_pairs.txt; : > $OUT
export GENIE_HIP_LIBRARY=$R/1xgpt_amd/libgenie_hip_study.so
for mx in 32 64 128 256 32 64 128 256; do
  echo "== GENIE_ATTN_KEYSPLIT_MAX_PAIRS=$mx" >> $OUT
  GENIE_ATTN_KEYSPLIT_MAX_PAIRS=$mx python tools/bench_generate.py --batches 6 8 16 32 --steps 2 --schedules kv_cache 2>/dev/null | grep "^{'schedule" | cut -c1-140 >> $OUT
done
cat $OUT
