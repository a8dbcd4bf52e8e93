#!/bin/bash
# same-box A/B of the exact-precision spatial attention's softmax (lib_ab_old.so = library expf, lib_ab_new.so = v_exp_f32 on
# log2(e)-scaled scores): every exact-precision test with the new library, then the exact evaluate with its per-class times
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/${1:-r03}_exact_attn_ab.txt; : > $OUT
GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_new.so python -m pytest tests/test_hip_parity.py tests/test_hip_prefix_reuse.py tests/test_hip_bench_config.py tests/test_hip_configs.py tests/test_hip_train.py tests/test_hip_harness.py -m gpu -x -q 2>&1 | tail -2 >> $OUT
for v in old new old new; do
  GENIE_HIP_LIBRARY=$R/1xgpt_amd/lib_ab_$v.so python bench.py --precision exact --batch 64 --breakdown --no-cpu-baseline --no-train-leg --no-secondary --no-board-sampler --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); b=d.get('breakdown',{}); print('$v', round(d['value'],1), 'frames/s  ce', d['ce'], ' attn_spatial ms', b.get('attn_spatial',{}).get('ms'), 'gemm ms', b.get('gemm',{}).get('ms'), 'step ms', b.get('step_ms'))" >> $OUT
done
cat $OUT
