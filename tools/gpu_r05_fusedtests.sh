#!/bin/bash
# round 5: fused-kernel unit tests (spatial entry point, launch proof), frame_linear, bf16 suites after the GELU unification
TAG=${1:-r05g}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hip_fused.py tests/test_hip_frame.py tests/test_hip_bf16.py -x -q -m gpu -s > gpurun_out/${TAG}_tests.txt 2>&1; tail -5 gpurun_out/${TAG}_tests.txt
grep -E "spatial fused unit|frame_linear M|fused launches|fused vs unfused|batch" gpurun_out/${TAG}_tests.txt | head -40
timeout 1200 python -m pytest tests -x -q -m gpu -k "bf16 or config2 or bench_config" > gpurun_out/${TAG}_tests2.txt 2>&1; tail -3 gpurun_out/${TAG}_tests2.txt
