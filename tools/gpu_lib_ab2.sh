#!/bin/bash
# same-box A/B of prebuilt library variants 1xgpt_amd/lib_ab_<name>.so (loaded through GENIE_HIP_LIBRARY): the headline bench and the
# bf16 leg, interleaved repeats.   usage: tools/gpu_lib_ab2.sh <tag> <name> <name> ...
TAG=$1; shift
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/${TAG}_lib_ab.txt; : > $OUT
for rep in 1 2 3; do
  for v in "$@"; do
    for prec in f16x3 bf16; do
    GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_$v.so python bench.py --precision $prec --no-cpu-baseline --no-train-leg --no-secondary --no-board-sampler --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$prec', 'rep$rep', round(d['value'],1), 'frames/s  GEMM', round(d['roofline']['achieved'],1), 'TF  ce', d['ce'])" >> $OUT
    done
  done
done
cat $OUT
