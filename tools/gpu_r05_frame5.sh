#!/bin/bash
# round 5: generate as one library call: parity tests, timings
TAG=${1:-r05m}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hip_frame.py -x -q -m gpu > gpurun_out/${TAG}_frame_tests.txt 2>&1; tail -5 gpurun_out/${TAG}_frame_tests.txt
timeout 1200 python -m pytest tests/test_hip_prefix_reuse.py tests/test_hip_configs.py tests/test_hip_harness.py -x -q -m gpu -k "generate or single_frame or prompt_pass or config5" > gpurun_out/${TAG}_gen_tests.txt 2>&1; tail -3 gpurun_out/${TAG}_gen_tests.txt
python tools/bench_generate.py --batches 1 2 4 8 16 --steps 2 8 --schedules kv_cache > gpurun_out/${TAG}_generate.txt 2>&1
grep "^{'schedule" gpurun_out/${TAG}_generate.txt | cut -c1-150
python tools/bench_e2e.py > gpurun_out/${TAG}_e2e.json 2> gpurun_out/${TAG}_e2e.err; tail -1 gpurun_out/${TAG}_e2e.json | cut -c300-700
