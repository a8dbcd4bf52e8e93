#!/bin/bash
# rocprofv3 kernel trace of any tools/*.py benchmark -> per-kernel table.  usage: <tag> <tools/script.py> [args]
TAG=$1; SCRIPT=$2; shift 2
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace -o trace -- python3 $GRAFT_REPO_ROOT/$SCRIPT "$@" > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_out.txt 2> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_rocprof.err
cd $GRAFT_REPO_ROOT
db=$(find gpurun_out/${TAG}_trace -name "*.db" | head -1)
echo "# cd /tmp && rocprofv3 --kernel-trace -- python3 $SCRIPT $@" > gpurun_out/${TAG}_kernel_stats.txt
python tools/rocprof_summary.py "$db" gpurun_out/${TAG}_kernel_stats.txt | head -40
rm -rf gpurun_out/${TAG}_trace
tail -3 gpurun_out/${TAG}_out.txt | cut -c1-600
