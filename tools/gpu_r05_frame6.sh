#!/bin/bash
TAG=${1:-r05n}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gpu_tests.txt 2>&1; tail -4 gpurun_out/${TAG}_gpu_tests.txt
