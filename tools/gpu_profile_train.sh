#!/bin/bash
# rocprofv3 kernel trace of the training step (tools/bench_train.py) -> per-kernel table.  usage: <tag> [bench_train args]
TAG=${1:-r02train}; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace -o trace -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_train.txt 2> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_rocprof.err
cd $GRAFT_REPO_ROOT
db=$(find gpurun_out/${TAG}_trace -name "*.db" | head -1)
echo "# cd /tmp && rocprofv3 --kernel-trace -- python3 tools/bench_train.py $@" > gpurun_out/${TAG}_kernel_stats.txt
python tools/rocprof_summary.py "$db" gpurun_out/${TAG}_kernel_stats.txt | head -45
rm -rf gpurun_out/${TAG}_trace
tail -3 gpurun_out/${TAG}_train.txt
