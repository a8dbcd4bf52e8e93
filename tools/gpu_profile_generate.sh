#!/bin/bash
# rocprofv3 kernel trace of batch-1 generate (prompt 8 -> 8 frames) -> per-kernel table.  usage: <tag> [bench_generate args]
TAG=${1:-r02gen}; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace -o trace -- python3 $GRAFT_REPO_ROOT/tools/bench_generate.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_generate.txt 2> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_rocprof.err
cd $GRAFT_REPO_ROOT
db=$(find gpurun_out/${TAG}_trace -name "*.db" | head -1)
echo "# cd /tmp && rocprofv3 --kernel-trace -- python3 tools/bench_generate.py $@" > gpurun_out/${TAG}_kernel_stats.txt
python tools/rocprof_summary.py "$db" gpurun_out/${TAG}_kernel_stats.txt | head -40
python tools/rocprof_gaps.py "$db" gpurun_out/${TAG}_kernel_stats.txt
rm -rf gpurun_out/${TAG}_trace
cat gpurun_out/${TAG}_generate.txt | head -5
