#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for c in 16 32 64; do python tools/bench_e2e.py --chunk $c 2>/dev/null | tail -1 | cut -c1-120,260-700; done > gpurun_out/r05t_e2e_chunks.txt
cat gpurun_out/r05t_e2e_chunks.txt
