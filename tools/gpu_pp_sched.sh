#!/bin/bash
# needs the study build of the library (GENIE_STUDY=1 python 1xgpt_amd/build.py): the shipping library has no study knobs
export GENIE_HIP_LIBRARY=${GENIE_HIP_LIBRARY:-$(cd "$(dirname "$0")/.." && pwd)/1xgpt_amd/libgenie_hip_study.so}
# A/B of the LDS-DMA placement (GENIE_PP_SCHED) + per-workgroup s_memtime stamps (GENIE_PP_ABL=32).  usage: <batch> <tag>
B=${1:-48}; TAG=${2:-pp}
mkdir -p gpurun_out
{
for r in 1 2; do
for s in 0 1; do
  echo "== GENIE_PP_SCHED=$s (round $r)"; GENIE_PP_SCHED=$s python tools/bench_gemm.py --batch $B --prec f16x3 bf16 2>/dev/null
done; done
echo "== timing stamps"; GENIE_PP_ABL=32 python tools/bench_gemm.py --batch $B --prec f16x3 bf16 2>&1 | grep -E "pp_timing" | sort | uniq -c | sort -rn | head -40
} > gpurun_out/${TAG}_sched.log 2>&1
