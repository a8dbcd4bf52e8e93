#!/bin/bash
# round 5: frame kernels with heads of 32: parity tests, generate-related suites, C35 / C138 generate timings
TAG=${1:-r05l}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hip_frame.py -x -q -m gpu > gpurun_out/${TAG}_frame_tests.txt 2>&1; tail -5 gpurun_out/${TAG}_frame_tests.txt
timeout 1200 python -m pytest tests/test_hip_prefix_reuse.py tests/test_hip_configs.py tests/test_hip_harness.py tests/test_hip_bench_config.py -x -q -m gpu -s -k "generate or single_frame or prompt_pass or config5 or fused_subblocks_effect" > gpurun_out/${TAG}_gen_tests.txt 2>&1; tail -3 gpurun_out/${TAG}_gen_tests.txt; grep "ids: fused" gpurun_out/${TAG}_gen_tests.txt
python tools/bench_generate.py --model c35 --batches 1 16 --steps 2 8 --schedules kv_cache > gpurun_out/${TAG}_generate_c35.txt 2>&1
grep "^{'schedule" gpurun_out/${TAG}_generate_c35.txt | cut -c1-150
GENIE_NO_FRAME_KERNELS=1 python tools/bench_generate.py --model c35 --batches 1 16 --steps 2 --schedules kv_cache > gpurun_out/${TAG}_generate_c35_rowmajor.txt 2>&1
grep "^{'schedule" gpurun_out/${TAG}_generate_c35_rowmajor.txt | cut -c1-150
python tools/bench_generate.py --batches 1 8 16 --steps 2 8 --schedules kv_cache > gpurun_out/${TAG}_generate.txt 2>&1
grep "^{'schedule" gpurun_out/${TAG}_generate.txt | cut -c1-150
bash tools/gpu_profile_generate.sh ${TAG}_gen1 --batches 1 --steps 2 --schedules kv_cache > /dev/null 2>&1
head -16 gpurun_out/${TAG}_gen1_kernel_stats.txt | cut -c1-165; tail -2 gpurun_out/${TAG}_gen1_kernel_stats.txt
bash tools/gpu_profile_generate.sh ${TAG}_gen16 --batches 16 --steps 2 --schedules kv_cache > /dev/null 2>&1
head -14 gpurun_out/${TAG}_gen16_kernel_stats.txt | cut -c1-165
