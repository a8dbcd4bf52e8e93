#!/bin/bash
# same-box A/B of a variant library: frame tests on the variant first (a build flag may break it), then generate timings
TAG=$1; VAR=$2; shift; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
GENIE_HIP_LIBRARY=$GRAFT_REPO_ROOT/1xgpt_amd/lib_ab_$VAR.so timeout 600 python -m pytest tests/test_hip_frame.py tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -3
bash tools/gpu_r05_ab.sh $TAG $VAR "$@"
