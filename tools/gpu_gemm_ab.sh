#!/bin/bash
# needs the study build of the library (GENIE_STUDY=1 python 1xgpt_amd/build.py): the shipping library has no study knobs
export GENIE_HIP_LIBRARY=${GENIE_HIP_LIBRARY:-$(cd "$(dirname "$0")/.." && pwd)/1xgpt_amd/libgenie_hip_study.so}
# A/B of the 16-bit GEMM kernels on the model's shapes (run on the GPU box through gpurun).
# usage: tools/gpu_gemm_ab.sh <batch> <out-file>
B=${1:-48}; OUT=${2:-gpurun_out/gemm_ab.log}
mkdir -p gpurun_out
{
echo "== pp kernel (default)"; python tools/bench_gemm.py --batch $B --prec f16x3 bf16
echo "== pp kernel, fused GELU epilogue"; python tools/bench_gemm.py --batch $B --prec f16x3 bf16 --gelu 1
echo "== old kernels (GENIE_GEMM16_PP=0)"; GENIE_GEMM16_PP=0 python tools/bench_gemm.py --batch $B --prec f16x3 bf16
echo "== f16x3 TERMS=2"; GENIE_F16_TERMS=2 python tools/bench_gemm.py --batch $B --prec f16x3
echo "== f16x3 TERMS=1 (plain f16)"; GENIE_F16_TERMS=1 python tools/bench_gemm.py --batch $B --prec f16x3
} > $OUT 2>&1
