#!/bin/bash
# needs the study build of the library (GENIE_STUDY=1 python 1xgpt_amd/build.py): the shipping library has no study knobs
export GENIE_HIP_LIBRARY=${GENIE_HIP_LIBRARY:-$(cd "$(dirname "$0")/.." && pwd)/1xgpt_amd/libgenie_hip_study.so}
# A/B of the persistent gemm16_pp launch (GENIE_PP_PERSIST=1, default) against one workgroup per tile (=0), plus stamps
B=${1:-48}; TAG=${2:-pp}
mkdir -p gpurun_out
{
python -m pytest tests/test_hip_bf16.py tests/test_hip_configs.py tests/test_hip_f16x3.py -x -q -m gpu 2>&1 | tail -3
for e in 1 0 1 0; do
  echo "== GENIE_PP_PERSIST=$e"; GENIE_PP_PERSIST=$e python tools/bench_gemm.py --batch $B --prec f16x3 bf16 2>/dev/null
done
echo "== stamps, persistent, compile-time epilogue (ABL 33)"; GENIE_PP_ABL=33 python tools/bench_gemm.py --batch $B --prec f16x3 bf16 2>&1 | grep -E "pp_timing" | awk '{print $2,$3,$4,$5,$6,$8,$11,$13}' | sort | uniq -c | sort -k2 | awk 'NR%7==1'
} > gpurun_out/${TAG}_persist.log 2>&1
