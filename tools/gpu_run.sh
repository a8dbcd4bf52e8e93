#!/bin/bash
# The GPU-box recipes of this repository in ONE parameterised script (run through gpurun: `gpurun -- bash tools/gpu_run.sh <task> <tag> ...`).
# Everything is written under gpurun_out/<tag>_*; what is worth keeping is copied into profiles/ by hand.
#
#   suite   <tag> [pytest args]        the -m gpu suite (measured bf16 deltas -> <tag>_measured_deltas.txt), then nothing else
#   bench   <tag> [bench.py args]      bench.py as the driver runs it (--gpus 1 --steps 20 --warmup 5 unless args are given), wall-clock measured
#   full    <tag>                      suite + bench
#   trace   <tag> [bench.py args]      rocprofv3 --kernel-trace of one bench step -> <tag>_kernel_stats.txt (per-kernel table)
#   tracecmd <tag> <python file> [args]  the same for any python entry (tools/bench_generate.py ...)
#   pmc     <tag> <precision> <clips> [model] [extra bench args]   three rocprofv3 --pmc passes -> <tag>_pmc_bench.json
#   ab      <tag> <variant> <variant> ...   same-box A/B of 1xgpt_amd/lib_ab_<variant>.so (python 1xgpt_amd/build.py --variant ...):
#                                      interleaved repeats of the command in $AB_CMD (default: the headline bench, 6 steps)
set -u
TASK=${1:?task}; TAG=${2:?tag}; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out
QUIET="--no-cpu-baseline --no-train-leg --no-secondary --no-board-sampler"

suite() {
  rm -f gpurun_out/${TAG}_measured_deltas.txt
  GENIE_TEST_RECORD=$R/gpurun_out/${TAG}_measured_deltas.txt timeout 2400 python -m pytest tests -x -q -m gpu "$@" > gpurun_out/${TAG}_gpu_tests.txt 2>&1
  echo "suite rc=$?"; tail -4 gpurun_out/${TAG}_gpu_tests.txt
}
bench() {
  local t0=$(date +%s)
  if [ $# -eq 0 ]; then set -- --gpus 1 --steps 20 --warmup 5; fi
  timeout 1200 python bench.py "$@" > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
  echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"
  tail -c 2500 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err
}
trace_py() {   # <python file> [args]: kernel trace -> per-kernel table
  local script=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $R/gpurun_out/${TAG}_trace -o trace -- python3 $R/$script "$@" \
      > $R/gpurun_out/${TAG}_under_rocprof.out 2> $R/gpurun_out/${TAG}_rocprof.err )
  local db=$(find gpurun_out/${TAG}_trace -name "*.db" | head -1)
  echo "# cd /tmp && rocprofv3 --kernel-trace -- python3 $script $*" > gpurun_out/${TAG}_kernel_stats.txt
  python tools/rocprof_summary.py "$db" gpurun_out/${TAG}_kernel_stats.txt | head -40
  if [ -f tools/rocprof_gaps.py ]; then python tools/rocprof_gaps.py "$db" >> gpurun_out/${TAG}_kernel_stats.txt 2>/dev/null; fi
  rm -rf gpurun_out/${TAG}_trace
  tail -c 1200 gpurun_out/${TAG}_under_rocprof.out
}
pmc() {
  local PREC=${1:-f16x3} CLIPS=${2:-128} MODEL=${3:-c138}; shift 3 2>/dev/null || true
  local ARGS="--steps 1 --warmup 0 --no-events $QUIET --precision $PREC --batch $CLIPS --model $MODEL $*"
  ( cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_pf --output-format csv -- python3 $R/bench.py $ARGS > /dev/null 2>&1
    rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_pw --output-format csv -- python3 $R/bench.py $ARGS > /dev/null 2>&1
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_ps --output-format csv -- python3 $R/bench.py $ARGS > /dev/null 2>&1 )
  local f1=$(find gpurun_out/${TAG}_pf -name "*counter_collection.csv" | head -1)
  local f2=$(find gpurun_out/${TAG}_pw -name "*counter_collection.csv" | head -1)
  local f3=$(find gpurun_out/${TAG}_ps -name "*counter_collection.csv" | head -1)
  python tools/pmc_bench_summary.py $PREC $CLIPS "$f1" "$f2" "$f3" gpurun_out/${TAG}_pmc_bench.json \
    "rocprofv3 --pmc {FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE} (three passes, csv) -- python3 bench.py $ARGS"
  rm -rf gpurun_out/${TAG}_pf gpurun_out/${TAG}_pw gpurun_out/${TAG}_ps
}
ab() {
  local OUT=gpurun_out/${TAG}_lib_ab.txt; : > $OUT
  local CMD=${AB_CMD:-"python bench.py $QUIET --steps 6 --warmup 2"}
  echo "# $CMD   (GENIE_HIP_LIBRARY=1xgpt_amd/lib_ab_<variant>.so; 'shipping' = 1xgpt_amd/libgenie_hip.so)" >> $OUT
  for rep in 1 2 3; do
    for v in "$@"; do
      local lib=$R/1xgpt_amd/lib_ab_$v.so; [ "$v" = shipping ] && lib=$R/1xgpt_amd/libgenie_hip.so
      GENIE_HIP_LIBRARY=$lib $CMD 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
keys=[k for k in ('value','ms_per_step','ce','ms','fps','ms_per_frame') if k in d]
print('$v', 'rep$rep', ' '.join(f'{k}={d[k]:.6g}' if isinstance(d[k],(int,float)) else f'{k}={d[k]}' for k in keys))" >> $OUT
    done
  done
  cat $OUT
}
case $TASK in
  suite) suite "$@";;
  bench) bench "$@";;
  full) suite; bench;;
  trace) trace_py bench.py $QUIET "$@";;
  tracecmd) trace_py "$@";;
  pmc) pmc "$@";;
  ab) ab "$@";;
  *) echo "unknown task $TASK"; exit 2;;
esac
