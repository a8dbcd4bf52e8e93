#!/bin/bash
# SQ counter passes of the fused sub-block kernels (tools/bench_fused.py, shipping library).  usage: tools/gpu_pmc_fused.sh <tag>
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
   -d $R/gpurun_out/${TAG}_pf1 --output-format csv -- python3 $R/tools/bench_fused.py --iters 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS \
   -d $R/gpurun_out/${TAG}_pf2 --output-format csv -- python3 $R/tools/bench_fused.py --iters 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVES SQ_INSTS_VALU_TRANS_F32 \
   -d $R/gpurun_out/${TAG}_pf3 --output-format csv -- python3 $R/tools/bench_fused.py --iters 3 > /dev/null 2>&1
cd $R
for i in 1 2 3; do
  f=$(find gpurun_out/${TAG}_pf$i -name "*counter_collection.csv" | head -1)
  python tools/pmc_csv_summary.py "$f" gpurun_out/${TAG}_pmc_fused_$i.json fused_bf16 > gpurun_out/${TAG}_pmc_fused_$i.txt 2>&1
  rm -rf gpurun_out/${TAG}_pf$i
done
cat gpurun_out/${TAG}_pmc_fused_1.txt gpurun_out/${TAG}_pmc_fused_2.txt gpurun_out/${TAG}_pmc_fused_3.txt
