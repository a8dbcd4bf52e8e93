#!/bin/bash
# needs the study build of the library (GENIE_STUDY=1 python 1xgpt_amd/build.py): the shipping library has no study knobs
export GENIE_HIP_LIBRARY=${GENIE_HIP_LIBRARY:-$(cd "$(dirname "$0")/.." && pwd)/1xgpt_amd/libgenie_hip_study.so}
# Timing-only ablations (GENIE_PP_ABL) and one PMC pass of the phase-scheduled GEMM.  usage: tools/gpu_pp_study.sh <batch> <tag>
B=${1:-48}; TAG=${2:-pp}
mkdir -p gpurun_out
{
for abl in 0 1 2 3 4 8 16 19 11; do
  echo "== GENIE_PP_ABL=$abl"; GENIE_PP_ABL=$abl python tools/bench_gemm.py --batch $B --prec f16x3 bf16 2>/dev/null
done
} > gpurun_out/${TAG}_abl.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
   -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_gemm.py --batch $B --prec f16x3 bf16 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/${TAG}_pmc -name "*counter_collection.csv" | head -1)
python tools/pmc_csv_summary.py "$f" gpurun_out/${TAG}_pmc_summary.json gemm16 > gpurun_out/${TAG}_pmc_summary.txt 2>&1
rm -rf gpurun_out/${TAG}_pmc
