#!/bin/bash
# BASELINE config 2 (C35 bf16 forward + CE, 64 clips): same-box pair fused sub-block kernels ON / OFF (GENIE_NO_FUSED=1 makes
# the weight table leave the fused streams out, i.e. the round-3 launch sequence), then the kernel trace + three PMC passes of
# the shipping path.   usage: tools/gpu_config2_final.sh <tag>
TAG=${1:-r04d}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/${TAG}_config2_fused_ab.txt; : > $OUT
for rep in 1 2 3; do
  for nf in 0 1; do
    GENIE_NO_FUSED=$nf python tools/bench_forward.py --precision bf16 --iters 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read())['results']; print('fused' if $nf == 0 else 'unfused', 'rep$rep', ' '.join('%s %.2f ms (%.0f model TFLOP/s)' % (r['path'].split(' ')[0], r['ms'], r['model_tflops']) for r in d), ' loss', d[0]['loss'])" >> $OUT
  done
done
cat $OUT
bash tools/gpu_profile_forward.sh ${TAG}_c35_bf16 | tail -24
