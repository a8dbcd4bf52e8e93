#!/bin/bash
# round 5: frame kernels incl. the LDS-tiled mid-size Linear: parity, generate at 1-16 clips, e2e, traces at 8 and 16 clips
TAG=${1:-r05d}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_frame.py -x -q -m gpu > gpurun_out/${TAG}_frame_tests.txt 2>&1; tail -5 gpurun_out/${TAG}_frame_tests.txt
timeout 900 python -m pytest tests/test_hip_prefix_reuse.py tests/test_hip_configs.py -x -q -m gpu -k "generate or single_frame or prompt_pass or config5" > gpurun_out/${TAG}_gen_tests.txt 2>&1; tail -3 gpurun_out/${TAG}_gen_tests.txt
python tools/bench_generate.py --batches 1 2 4 8 16 --steps 2 8 --schedules kv_cache > gpurun_out/${TAG}_generate.txt 2>&1
grep "^{'schedule" gpurun_out/${TAG}_generate.txt | cut -c1-150
python tools/bench_e2e.py > gpurun_out/${TAG}_e2e.json 2> gpurun_out/${TAG}_e2e.err; tail -1 gpurun_out/${TAG}_e2e.json | cut -c1-700
bash tools/gpu_profile_generate.sh ${TAG}_gen16 --batches 16 --steps 2 --schedules kv_cache > /dev/null 2>&1
head -14 gpurun_out/${TAG}_gen16_kernel_stats.txt | cut -c1-165
bash tools/gpu_profile_generate.sh ${TAG}_gen8 --batches 8 --steps 2 --schedules kv_cache > /dev/null 2>&1
head -14 gpurun_out/${TAG}_gen8_kernel_stats.txt | cut -c1-165
