#!/bin/bash
# round 5: the whole -m gpu suite, then bench.py as the driver runs it (wall-clock measured)
TAG=${1:-r05f}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gpu_tests.txt 2>&1; tail -4 gpurun_out/${TAG}_gpu_tests.txt
t0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"
tail -c 1500 gpurun_out/${TAG}_bench.json
tail -3 gpurun_out/${TAG}_bench.err
