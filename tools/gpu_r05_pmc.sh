#!/bin/bash
# round 5: counter passes of the default bench step on this round's code: GENIE_138M f16x3 + bf16, GENIE_35M f16x3 + bf16
cd $GRAFT_REPO_ROOT
for spec in "f16x3 c138" "bf16 c138" "f16x3 c35" "bf16 c35"; do
  set -- $spec
  out=gpurun_out/r05_pmc_bench_$2.json
  cp -f gpurun_out/r05p_$2_pmc_bench.json /dev/null 2>&1
  bash tools/gpu_pmc_bench.sh r05p_$2 $1 128 $2 > gpurun_out/r05p_$1_$2.log 2>&1
  tail -3 gpurun_out/r05p_$1_$2.log | cut -c1-200
done
ls -la gpurun_out/r05p_*_pmc_bench.json
