#!/bin/bash
# same-box A/B of the QKV-flavoured gemm16_pp epilogue (prebuilt lib_ab_old.so / lib_ab_new.so): parity tests with the new library,
# the qkv GEMM alone through the spatial path is not reachable from bench_gemm, so: headline bench + bf16 leg, interleaved repeats
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/${1:-r03}_qkv_epi_ab.txt; : > $OUT
GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_new.so python -m pytest tests/test_hip_bench_config.py tests/test_hip_bf16.py tests/test_hip_f16x3.py tests/test_hip_configs.py tests/test_hip_prefix_reuse.py -m gpu -x -q 2>&1 | tail -3 >> $OUT
bash tools/gpu_lib_ab2.sh ${1:-r03}_qkv old new >> $OUT 2>&1
for v in old new old new; do
  GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_$v.so python bench.py --precision f16x3 --breakdown --no-cpu-baseline --no-train-leg --no-secondary --no-board-sampler --steps 4 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), json.dumps(d.get('breakdown', d.get('kernel_classes')))[:600])" >> $OUT
done
cat $OUT
