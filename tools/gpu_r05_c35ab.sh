#!/bin/bash
# same-box A/B of a variant library against the shipping one on `bench.py --model c35 --precision <prec>` (the metric loop of
# evaluate.py:107-116 on the shipped config): usage: <tag> <variant name> [precision: bf16 | f16x3]
TAG=$1; VAR=$2; PREC=${3:-bf16}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_c35_${PREC}_ab.txt
for rep in 1 2; do
  echo "== shipping" >> $OUT
  python bench.py --no-cpu-baseline --no-train-leg --no-secondary --model c35 --precision $PREC 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step', 'ce', d.get('ce'), 'selfcheck', d.get('parity_selfcheck',{}).get('ok'))" >> $OUT
  echo "== variant $VAR" >> $OUT
  GENIE_HIP_LIBRARY=$GRAFT_REPO_ROOT/1xgpt_amd/lib_ab_$VAR.so python bench.py --no-cpu-baseline --no-train-leg --no-secondary --model c35 --precision $PREC 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step', 'ce', d.get('ce'), 'selfcheck', d.get('parity_selfcheck',{}).get('ok'))" >> $OUT
done
cat $OUT
