#!/bin/bash
# Board power / shader clock sampled by rocm-smi while the 4096^3 GEMM microbench runs on zero-filled and on random operands
# (the same binary and launch): evidence for what bounds the matrix pipe on real data.  usage: tools/gpu_power_probe.sh
sample() {  # $1 = label: 12 samples, 0.25 s apart
  for i in $(seq 1 12); do
    /opt/rocm/bin/rocm-smi --showpower --showclocks --json 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.load(sys.stdin); c=d[sorted(d)[0]]
    keys=[k for k in c if 'ower' in k or 'sclk' in k.lower()]
    print('$1', {k:c[k] for k in keys})
except Exception as e:
    print('$1 parse error', e)
"
    sleep 0.25
  done
}
echo "== idle"; sample idle | tail -2
for mode in "--zero" ""; do
  echo "== bench_gemm $mode (f16x3 then bf16, 4096^3 only matters: long loop)"
  python3 tools/bench_gemm_loop.py $mode &
  PID=$!
  sleep 6
  sample "run$mode" | tail -8
  wait $PID
done
