#!/bin/bash
# Round-3 GEMM study (study build of the library): (1) per-tile stamps (prologue / main loop / epilogue, s_memtime ticks) of the
# f32-output epilogue over output widths N -- is a power-of-two output row stride slower? (2) non-temporal policy on the A / W
# LDS-DMA streams: time and FETCH_SIZE (one rocprofv3 --pmc pass each).   usage: tools/gpu_gemm_r03.sh <tag>
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
export GENIE_HIP_LIBRARY=$R/1xgpt_amd/libgenie_hip_study.so
OUT=$R/gpurun_out/${TAG}_gemm_study.txt
: > $OUT
ROWS=491520
echo "== stamps (GENIE_PP_ABL=32) over N, K = 512, M = $ROWS, f16x3 then bf16" >> $OUT
GENIE_PP_ABL=32 python tools/bench_gemm.py --rows $ROWS --prec f16x3 bf16 --shapes 512:512 768:512 1024:512 1280:512 1536:512 1792:512 2048:512 2304:512 512:2048 2>&1 | grep -E "pp_timing|TFLOP" >> $OUT
for abl in 32 96 160 224; do
  echo "== GENIE_PP_ABL=$abl (32 stamps, +64 nt on A, +128 nt on W)" >> $OUT
  GENIE_PP_ABL=$abl python tools/bench_gemm.py --rows $ROWS --prec f16x3 --shapes 512:512 1536:512 2048:512 512:2048 2>&1 | grep -E "TFLOP" >> $OUT
done
cd /tmp && export TMPDIR=/tmp
for abl in 32 96 160 224; do
  export GENIE_PP_ABL=$abl
  rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_pf$abl --output-format csv -- python3 $R/tools/bench_gemm.py --rows $ROWS --prec f16x3 --shapes 512:512 1536:512 2048:512 512:2048 > /dev/null 2>&1
  unset GENIE_PP_ABL
  f=$(find $R/gpurun_out/${TAG}_pf$abl -name "*counter_collection.csv" | head -1)
  echo "== FETCH_SIZE (KB as reported; x2 on gfx950 for bytes) GENIE_PP_ABL=$abl" >> $OUT
  python3 $R/tools/pmc_csv_summary.py "$f" $R/gpurun_out/${TAG}_pf$abl.json gemm16_pp >> $OUT 2>&1
  rm -rf $R/gpurun_out/${TAG}_pf$abl
done
cd $R
cat $OUT
