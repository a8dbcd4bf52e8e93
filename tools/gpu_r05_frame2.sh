#!/bin/bash
# round 5: frame kernels re-check (parity tests, generate timings at 1-4 clips, batch-1 kernel trace)
TAG=${1:-r05c}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_frame.py -x -q -m gpu > gpurun_out/${TAG}_frame_tests.txt 2>&1; tail -5 gpurun_out/${TAG}_frame_tests.txt
timeout 900 python -m pytest tests/test_hip_prefix_reuse.py tests/test_hip_configs.py -x -q -m gpu -k "generate or single_frame or prompt_pass" > gpurun_out/${TAG}_gen_tests.txt 2>&1; tail -3 gpurun_out/${TAG}_gen_tests.txt
python tools/bench_generate.py --batches 1 2 4 --steps 2 8 --schedules kv_cache > gpurun_out/${TAG}_generate.txt 2>&1
grep "^{'schedule" gpurun_out/${TAG}_generate.txt | cut -c1-150
bash tools/gpu_profile_generate.sh ${TAG}_gen1 --batches 1 --steps 2 --schedules kv_cache > /dev/null 2>&1
head -14 gpurun_out/${TAG}_gen1_kernel_stats.txt | cut -c1-165; tail -3 gpurun_out/${TAG}_gen1_kernel_stats.txt
