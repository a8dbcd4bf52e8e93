#!/bin/bash
# N > 1 start-up on the ONE GPU of a gpurun box: the -m gpu multi-rank test, then <n> back-to-back 2-rank bench starts
# (ranks pinned to device 0, gloo collectives) with wall time and exit code of each -> gpurun_out/<tag>_multirank.txt
TAG=${1:-r03}; N=${2:-5}
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_multirank.txt
: > $OUT
python -m pytest tests/test_hip_multirank.py -m gpu -x -q >> $OUT 2>&1
echo "pytest rc=$?" >> $OUT
for i in $(seq 1 $N); do
  t0=$(date +%s)
  GENIE_FORCE_DEVICE=0 GENIE_DIST_BACKEND=gloo GENIE_HIP_INIT_TIMEOUT=90 GENIE_RDZV_TIMEOUT=180 timeout 400 \
    python bench.py --gpus 2 --steps 1 --warmup 0 --batch 16 --no-secondary --no-cpu-baseline --train-batch 4 \
    > gpurun_out/${TAG}_g2_$i.json 2> gpurun_out/${TAG}_g2_$i.err
  rc=$?
  t1=$(date +%s)
  echo "run $i rc=$rc wall=$((t1 - t0)) s  $(python -c "
import json,sys
try:
    d=json.load(open('gpurun_out/${TAG}_g2_$i.json')); print('value',round(d['value'],1),'n_gpus',d['n_gpus'],'ranks',d['config']['ranks_reported_by_backend'],'backend',d['config']['collective_backend'],'train',d.get('train_step',{}).get('value'))
except Exception as e: print('no line:',e)
")" >> $OUT
  grep -h "stalled\|giving up\|fresh" gpurun_out/${TAG}_g2_$i.err >> $OUT
done
cat $OUT
