#!/bin/bash
# Three rocprofv3 --pmc passes of the default bench step (no trace domains) -> gpurun_out/<tag>_pmc_bench.json
# usage: tools/gpu_pmc_bench.sh <tag> <precision> <clips> [model: c138 | c35]
TAG=${1:-r02}; PREC=${2:-f16x3}; CLIPS=${3:-128}; MODEL=${4:-c138}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-events --no-board-sampler --no-secondary --no-train-leg --no-cpu-baseline --precision $PREC --batch $CLIPS --model $MODEL"
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_pf --output-format csv -- python3 $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_pw --output-format csv -- python3 $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_ps --output-format csv -- python3 $R/bench.py $ARGS > /dev/null 2>&1
cd $R
f1=$(find gpurun_out/${TAG}_pf -name "*counter_collection.csv" | head -1)
f2=$(find gpurun_out/${TAG}_pw -name "*counter_collection.csv" | head -1)
f3=$(find gpurun_out/${TAG}_ps -name "*counter_collection.csv" | head -1)
python tools/pmc_bench_summary.py $PREC $CLIPS "$f1" "$f2" "$f3" gpurun_out/${TAG}_pmc_bench.json "rocprofv3 --pmc {FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE} (three passes, csv) -- python3 bench.py $ARGS"
rm -rf gpurun_out/${TAG}_pf gpurun_out/${TAG}_pw gpurun_out/${TAG}_ps
