#!/bin/bash
# Upper bound on what overlapping gemm16_pp's epilogue with the next tile's K loop could buy: the headline's
# four GEMM shapes (128 clips x 15 frames = 491,520 rows, f16x3) with the epilogue's output stores REMOVED (study build,
# GENIE_PP_ABL=8: results wrong by construction) against the full kernel, on random operands, with board power and shader clock
# sampled under each.  Removing the stores is more than any overlap can do; if that gains x %, overlap gains less than x %.
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/${1:-r04}_pp_epilogue_bound.txt; : > $OUT
export GENIE_HIP_LIBRARY=$R/1xgpt_amd/libgenie_hip_study.so
sample() {
  for i in $(seq 1 8); do
    /opt/rocm/bin/rocm-smi --showpower --showclocks --json 2>/dev/null | python3 -c "
import json,sys,re
try:
    d=json.load(sys.stdin); best=None
    for k,c in d.items():
        p=[float(v) for kk,v in c.items() if 'ower' in kk and re.match(r'^[0-9.]+$', str(v))]
        s=[vv for kk,vv in c.items() if 'sclk' in kk.lower()]
        if p and (best is None or max(p)>best[0]): best=(max(p), s)
    print('$1', 'W', best[0], 'sclk', best[1])
except Exception as e:
    print('$1 parse error', e)
"
    sleep 0.4
  done
}
for shape in "491520 1536 512" "491520 512 512" "491520 2048 512" "491520 512 2048"; do
for abl in 0 8; do
  echo "== GENIE_PP_ABL=$abl  M N K = $shape" >> $OUT
  GENIE_PP_ABL=$abl python3 tools/bench_gemm_loop.py $shape 2>/dev/null >> $OUT &
  PID=$!
  sleep 7
  sample "abl$abl" | tail -3 >> $OUT
  wait $PID
done
done
cat $OUT
