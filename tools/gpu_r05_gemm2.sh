#!/bin/bash
TAG=${1:-r05s}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_frame.py -x -q -m gpu > gpurun_out/${TAG}_frame_tests.txt 2>&1; tail -3 gpurun_out/${TAG}_frame_tests.txt
python tools/bench_frame_gemm.py --rows 1024 2048 4096 > gpurun_out/${TAG}_frame_gemm.txt 2>&1; cat gpurun_out/${TAG}_frame_gemm.txt | cut -c1-200
python tools/bench_generate.py --batches 4 8 16 --steps 2 8 --schedules kv_cache > gpurun_out/${TAG}_generate.txt 2>&1
grep "^{'schedule" gpurun_out/${TAG}_generate.txt | cut -c1-150
python tools/bench_e2e.py > gpurun_out/${TAG}_e2e.json 2> gpurun_out/${TAG}_e2e.err; tail -1 gpurun_out/${TAG}_e2e.json | cut -c300-700
