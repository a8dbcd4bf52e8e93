#!/bin/bash
# same-box A/B of two prebuilt libraries (1xgpt_amd/lib_ab_old.so / lib_ab_new.so) on the evaluate schedule: the shipped config and
# the GENIE_138M shape in f16x3, and the GENIE_138M shape in bf16 (all three write the spatial attention's operand planes from the
# qkv GEMM's epilogue).  usage: <tag>
TAG=$1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_qkv_epilogue_ab.txt
run() { GENIE_HIP_LIBRARY=$GRAFT_REPO_ROOT/1xgpt_amd/lib_ab_$1.so python bench.py --no-cpu-baseline --no-train-leg --no-secondary --steps 2 --warmup 1 --model $2 --precision $3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 $3', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms/step', 'ce', d.get('ce'))" >> $OUT; }
for rep in 1 2; do
  for lib in old new; do
    run $lib c35 f16x3
    run $lib c138 f16x3
    run $lib c138 bf16
  done
done
cat $OUT
