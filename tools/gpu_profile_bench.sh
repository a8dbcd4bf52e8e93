#!/bin/bash
# rocprofv3 kernel trace of the default bench step (f16x3, prefix reuse) -> per-kernel table.  usage: <tag> [bench args]
TAG=${1:-r02}; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-train-leg --no-secondary --no-board-sampler "$@" > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_rocprof.err
cd $GRAFT_REPO_ROOT
db=$(find gpurun_out/${TAG}_trace -name "*.db" | head -1)
echo "# cd /tmp && rocprofv3 --kernel-trace -- python3 bench.py --no-cpu-baseline --no-train-leg --no-secondary $@" > gpurun_out/${TAG}_kernel_stats.txt
python tools/rocprof_summary.py "$db" gpurun_out/${TAG}_kernel_stats.txt | head -30
rm -rf gpurun_out/${TAG}_trace
tail -c 1500 gpurun_out/${TAG}_bench_under_rocprof.json
