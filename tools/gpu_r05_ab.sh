#!/bin/bash
# same-box A/B of a variant library against the shipping one on generate: usage: <tag> <variant name> [bench_generate args]
TAG=$1; VAR=$2; shift; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
  echo "== shipping" >> gpurun_out/${TAG}_ab.txt
  python tools/bench_generate.py "$@" 2>&1 | grep "^{'schedule" | cut -c1-140 >> gpurun_out/${TAG}_ab.txt
  echo "== variant $VAR" >> gpurun_out/${TAG}_ab.txt
  GENIE_HIP_LIBRARY=$GRAFT_REPO_ROOT/1xgpt_amd/lib_ab_$VAR.so python tools/bench_generate.py "$@" 2>&1 | grep "^{'schedule" | cut -c1-140 >> gpurun_out/${TAG}_ab.txt
done
cat gpurun_out/${TAG}_ab.txt
