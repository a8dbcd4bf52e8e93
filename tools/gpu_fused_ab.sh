#!/bin/bash
# same-box A/B of library variants (1xgpt_amd/lib_ab_<name>.so) on the fused kernels' microbenchmark, interleaved repeats
for rep in 1 2 3; do for v in "$@"; do echo -n "$v rep$rep: "; GENIE_HIP_LIBRARY=$GRAFT_REPO_ROOT/1xgpt_amd/lib_ab_$v.so python tools/bench_fused.py 2>&1 | grep "temporal_fused\|mlp_fused  " | awk '{printf "%s %s us  ", $1, $2}'; echo; done; done
