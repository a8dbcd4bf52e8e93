#!/bin/bash
# Where the energy of the 256x256 phase-scheduled GEMM goes: the 4096^3 f16x3 loop on RANDOM operands with parts of the main loop
# removed (study build, GENIE_PP_ABL: 1 = no LDS-DMA inside the loop, 2 = no fragment reads inside the loop, 3 = both: matrix
# instructions on fixed random registers, 16 = no matrix instructions: data movement only, 256 = every K-tile re-reads the addresses of K-tile 0: the same LDS-DMA instructions, all L2 hits), board power and shader clock sampled by
# rocm-smi under each.  The results of the ablated runs are wrong by construction; only power / clock / time are read.
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/${1:-r03}_gemm_energy_split.txt; : > $OUT
export GENIE_HIP_LIBRARY=$R/1xgpt_amd/libgenie_hip_study.so
sample() {
  for i in $(seq 1 10); do
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
for shape in "4096 4096 4096" "196608 1536 512" "196608 512 2048"; do
for abl in 0 256 1 2 3 16; do
  echo "== GENIE_PP_ABL=$abl  M N K = $shape" >> $OUT
  GENIE_PP_ABL=$abl python3 tools/bench_gemm_loop.py $shape 2>/dev/null >> $OUT &
  PID=$!
  sleep 6
  sample "abl$abl" | tail -4 >> $OUT
  wait $PID
done
done
echo "== zero-filled operands, full kernel" >> $OUT
python3 tools/bench_gemm_loop.py --zero 2>/dev/null >> $OUT &
PID=$!; sleep 6; sample zero | tail -6 >> $OUT; wait $PID
cat $OUT
