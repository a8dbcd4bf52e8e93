#!/bin/bash
# Ablations of the fused sub-block kernels (study build; results wrong by construction): where does the time go?
export GENIE_HIP_LIBRARY=$GRAFT_REPO_ROOT/1xgpt_amd/libgenie_hip_study.so
for abl in 0 1 2 3 4 8 15; do
  echo "== GENIE_FUSED_ABL=$abl (1 no LDS-DMA, 2 no residual rd/wr, 4 no GELU, 8 no LN loads)"
  GENIE_FUSED_ABL=$abl python tools/bench_fused.py 2>&1 | grep -v "amdgpu.ids\|STUDY"
done
