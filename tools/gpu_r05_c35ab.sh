#!/bin/bash
# same-box A/B of a variant library against the shipping one on `bench.py --model c35 --precision bf16` (the metric loop of
# evaluate.py:107-116 on the shipped config): usage: <tag> <variant name>
TAG=$1; VAR=$2
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_c35_bf16_ab.txt
for rep in 1 2; do
  echo "== shipping" >> $OUT
  python bench.py --no-cpu-baseline --no-train-leg --no-secondary --model c35 --precision bf16 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step')" >> $OUT
  echo "== variant $VAR" >> $OUT
  GENIE_HIP_LIBRARY=$GRAFT_REPO_ROOT/1xgpt_amd/lib_ab_$VAR.so python bench.py --no-cpu-baseline --no-train-leg --no-secondary --model c35 --precision bf16 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step')" >> $OUT
done
cat $OUT
