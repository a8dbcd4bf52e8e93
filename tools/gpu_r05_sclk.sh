#!/bin/bash
# shader clock and matrix-pipe busy fraction per kernel NAME of one bench step (one --pmc pass, no trace domains).  usage: <tag> [bench args]
TAG=$1; shift
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_ps --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-events --no-board-sampler --no-secondary --no-train-leg --no-cpu-baseline "$@" > /dev/null 2>&1
cd $R
f=$(find gpurun_out/${TAG}_ps -name "*counter_collection.csv" | head -1)
echo "# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 1 --warmup 0 --no-events --no-board-sampler --no-secondary --no-train-leg --no-cpu-baseline $@" > gpurun_out/${TAG}_per_kernel_clock.txt
python tools/pmc_per_kernel.py "$f" /tmp/pk.txt > /dev/null; cat /tmp/pk.txt >> gpurun_out/${TAG}_per_kernel_clock.txt
rm -rf gpurun_out/${TAG}_ps
cat gpurun_out/${TAG}_per_kernel_clock.txt | cut -c1-60,108-160
