#!/bin/bash
# same-box A/B of prebuilt library variants (1xgpt_amd/build/ab/lib_<name>.so): short headline bench with each, twice, interleaved
cp 1xgpt_amd/libgenie_hip.so /tmp/lib_keep.so
for rep in 1 2; do
  for v in "$@"; do
    cp 1xgpt_amd/build/ab/lib_$v.so 1xgpt_amd/libgenie_hip.so
    python bench.py --no-cpu-baseline --no-train-leg --no-secondary --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['roofline']['achieved'],1))"
  done
done
cp /tmp/lib_keep.so 1xgpt_amd/libgenie_hip.so
