#!/bin/bash
B=${1:-48}; TAG=${2:-pp}
mkdir -p gpurun_out
{
for g in 0 2 4 8 0 4; do
  echo "== GENIE_PP_STAGGER=$g"; GENIE_PP_STAGGER=$g GENIE_PP_STAGGER_MIN_TILES=1 python tools/bench_gemm.py --batch $B --prec f16x3 bf16 2>/dev/null
done
echo "== timing stamps, stagger 4"; GENIE_PP_STAGGER=4 GENIE_PP_STAGGER_MIN_TILES=1 GENIE_PP_ABL=32 python tools/bench_gemm.py --batch $B --prec f16x3 bf16 2>&1 | grep -E "pp_timing" | awk '{print $2,$3,$4,$5,$6,$8,$11,$13}' | sort | uniq -c | sort -k2 | awk 'NR%7==1' 
} > gpurun_out/${TAG}_stagger.log 2>&1
