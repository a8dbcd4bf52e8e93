#!/bin/bash
# per-kernel time of the headline's GEMM flavours under timing variants of the QKV operand-plane epilogue (results of the variants are
# wrong by construction): rocprofv3 kernel trace of one bench step per library.  usage: <tag> <variant> ...
TAG=$1; shift
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/${TAG}_qkv_epilogue_abl.txt; : > $OUT
for v in "$@"; do
  cd /tmp && export TMPDIR=/tmp
  GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_$v.so rocprofv3 --kernel-trace -d $R/gpurun_out/${TAG}_tr_$v -o trace -- python3 $R/bench.py --no-cpu-baseline --no-train-leg --no-secondary --no-board-sampler --steps 1 --warmup 1 > /dev/null 2>&1
  cd $R
  db=$(find gpurun_out/${TAG}_tr_$v -name "*.db" | head -1)
  echo "== $v" >> $OUT
  python tools/rocprof_summary.py "$db" /tmp/ks_$v.txt > /dev/null 2>&1
  grep "gemm16_pp_kernel" /tmp/ks_$v.txt | cut -c1-60,92-150 >> $OUT
  rm -rf gpurun_out/${TAG}_tr_$v
done
cat $OUT
