#!/bin/bash
# round-5 same-box baseline of the one-frame passes: generate timings at 1 / 8 / 16 clips + kernel traces at 8 and 16 clips
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/bench_generate.py --batches 1 8 16 --steps 2 8 --schedules kv_cache > gpurun_out/r05a_generate.txt 2>&1
python tools/bench_e2e.py > gpurun_out/r05a_e2e.json 2> gpurun_out/r05a_e2e.err
bash tools/gpu_profile_generate.sh r05a_gen16 --batches 16 --steps 2 --schedules kv_cache > /dev/null 2>&1
bash tools/gpu_profile_generate.sh r05a_gen8 --batches 8 --steps 2 --schedules kv_cache > /dev/null 2>&1
grep "^{'schedule" gpurun_out/r05a_generate.txt | cut -c1-150; tail -1 gpurun_out/r05a_e2e.json | cut -c1-600
head -14 gpurun_out/r05a_gen16_kernel_stats.txt | cut -c1-160
