#!/bin/bash
# needs the study build of the library (GENIE_STUDY=1 python 1xgpt_amd/build.py): the shipping library has no study knobs
export GENIE_HIP_LIBRARY=${GENIE_HIP_LIBRARY:-$(cd "$(dirname "$0")/.." && pwd)/1xgpt_amd/libgenie_hip_study.so}
# SQ counter passes of the default bench step, reduced for the spatial attention kernel.  usage: tools/gpu_pmc_attn.sh <tag>
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-events --no-secondary --no-train-leg --no-cpu-baseline --batch 128"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
   -d $R/gpurun_out/${TAG}_pa1 --output-format csv -- python3 $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS \
   -d $R/gpurun_out/${TAG}_pa2 --output-format csv -- python3 $R/bench.py $ARGS > /dev/null 2>&1
cd $R
for i in 1 2; do
  f=$(find gpurun_out/${TAG}_pa$i -name "*counter_collection.csv" | head -1)
  python tools/pmc_csv_summary.py "$f" gpurun_out/${TAG}_pmc_attn_$i.json attn_spatial attn_temporal layer_norm > gpurun_out/${TAG}_pmc_attn_$i.txt 2>&1
  rm -rf gpurun_out/${TAG}_pa$i
done
cat gpurun_out/${TAG}_pmc_attn_1.txt gpurun_out/${TAG}_pmc_attn_2.txt
