#!/bin/bash
# same-box A/B of prebuilt library variants 1xgpt_amd/lib_ab_<name>.so (python 1xgpt_amd/build.py --variant <name> -D...) on
# BASELINE config 2 (C35 bf16 forward + CE, 64 clips), interleaved repeats.   usage: tools/gpu_fwd_ab.sh <tag> <name> <name> ...
TAG=$1; shift
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/${TAG}_fwd_ab.txt; : > $OUT
for rep in 1 2 3; do
  for v in "$@"; do
    GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_$v.so python tools/bench_forward.py --precision bf16 --iters 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read())['results']; print('$v', 'rep$rep', ' '.join('%.2f ms' % r['ms'] for r in d), ' loss', d[0]['loss'])" >> $OUT
  done
done
cat $OUT
