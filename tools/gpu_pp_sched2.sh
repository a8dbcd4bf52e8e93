#!/bin/bash
# needs the study build of the library (GENIE_STUDY=1 python 1xgpt_amd/build.py): the shipping library has no study knobs
export GENIE_HIP_LIBRARY=${GENIE_HIP_LIBRARY:-$(cd "$(dirname "$0")/.." && pwd)/1xgpt_amd/libgenie_hip_study.so}
# A/B: LDS-DMA issued in the LOAD part (GENIE_PP_SCHED=0) vs inside the matrix cluster (=2)
B=${1:-48}; TAG=${2:-pp}
mkdir -p gpurun_out
{
for e in 0 2 0 2; do
  echo "== GENIE_PP_SCHED=$e"; GENIE_PP_SCHED=$e python tools/bench_gemm.py --batch $B --prec f16x3 bf16 2>/dev/null
done
} > gpurun_out/${TAG}_sched2.log 2>&1
