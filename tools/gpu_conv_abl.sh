#!/bin/bash
# needs the study build of the library (GENIE_STUDY=1 python 1xgpt_amd/build.py): the shipping library has no study knobs
export GENIE_HIP_LIBRARY=${GENIE_HIP_LIBRARY:-$(cd "$(dirname "$0")/.." && pwd)/1xgpt_amd/libgenie_hip_study.so}
# conv3x3 implicit-GEMM ablations (study knobs): full / no epilogue / no DMA in the loop / no MFMA / combinations
for a in 0 256 512 1024 768 1280 1536; do
  echo "ABL=$a"; GENIE_CONV_ABL=$a timeout 120 python tools/bench_conv.py --iters 5 2>&1 | grep "^{'H'" | head -4 | cut -c1-130
done
