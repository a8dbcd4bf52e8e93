#!/bin/bash
# exact-precision GEMM (v_mfma_f32_32x32x2_f32; peak 157.3 TFLOP/s): the LDS-DMA kernel against the register-staged one (study build:
# GENIE_GEMM_F32_DMA=0), correctness tests, microbench at the reuse path's M, one PMC pass
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/${1:-r03}_exact_gemm.txt; : > $OUT
python -m pytest tests/test_hip_parity.py tests/test_hip_prefix_reuse.py -m gpu -x -q 2>&1 | tail -2 >> $OUT
for rep in 1 2; do
echo "== gemm_f32_dma_kernel<16> (shipping) (rep $rep)" >> $OUT
python tools/bench_gemm.py --rows 61440 --prec exact --shapes 1536:512 512:512 2048:512 512:2048 1024:512 2>/dev/null | grep TFLOP >> $OUT
echo "== gemm_f32_dma_kernel<32> (study build, GENIE_GEMM_F32_DMA=32) (rep $rep)" >> $OUT
GENIE_HIP_LIBRARY=$R/1xgpt_amd/libgenie_hip_study.so GENIE_GEMM_F32_DMA=32 python tools/bench_gemm.py --rows 61440 --prec exact --shapes 1536:512 512:512 2048:512 512:2048 1024:512 2>/dev/null | grep TFLOP >> $OUT
echo "== gemm_f32_nt_kernel (study build, GENIE_GEMM_F32_DMA=0) (rep $rep)" >> $OUT
GENIE_HIP_LIBRARY=$R/1xgpt_amd/libgenie_hip_study.so GENIE_GEMM_F32_DMA=0 python tools/bench_gemm.py --rows 61440 --prec exact --shapes 1536:512 512:512 2048:512 512:2048 1024:512 2>/dev/null | grep TFLOP >> $OUT
done
python tools/bench_gemm.py --rows 491520 --prec exact --shapes 1536:512 2048:512 2>/dev/null | grep TFLOP >> $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
   -d $R/gpurun_out/${1:-r03}_exact_pmc --output-format csv -- python3 $R/tools/bench_gemm.py --rows 61440 --prec exact --shapes 1536:512 512:2048 > /dev/null 2>&1
cd $R
f=$(find gpurun_out/${1:-r03}_exact_pmc -name "*counter_collection.csv" | head -1)
python tools/pmc_csv_summary.py "$f" gpurun_out/${1:-r03}_exact_pmc.json gemm_f32 >> $OUT 2>&1
rm -rf gpurun_out/${1:-r03}_exact_pmc
cat $OUT
