#!/bin/bash
# bench.py on the SHIPPED config (magvit_n32_h8_d256): the JSON line in f16x3 (parity mode) and bf16 (fused sub-block kernels),
# plus the rocprofv3 kernel trace of the bf16 run.   usage: tools/gpu_bench_c35.sh <tag>
TAG=${1:-r04_c35}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
python bench.py --model c35 --precision f16x3 --no-train-leg --no-cpu-baseline > gpurun_out/${TAG}_bench_f16x3.json 2> gpurun_out/${TAG}_bench_f16x3.err; echo "f16x3 rc=$?"
python bench.py --model c35 --precision bf16 --no-train-leg --no-cpu-baseline > gpurun_out/${TAG}_bench_bf16.json 2> gpurun_out/${TAG}_bench_bf16.err; echo "bf16 rc=$?"
bash tools/gpu_profile_bench.sh ${TAG}_bf16 --model c35 --precision bf16 > gpurun_out/${TAG}_profile_bf16.log 2>&1
python - <<PY
import json
for p in ("f16x3","bf16"):
    d=json.load(open("gpurun_out/${TAG}_bench_%s.json"%p))
    r=d["roofline"]
    print(p,"value",round(d["value"],1),"ms",round(d["ms_per_step"],1),"ce",d["ce"],"kernel",r["kernel"][:60],"TF",round(r["achieved"],1),"frac",round(r["frac"],3),"selfcheck",d.get("parity_selfcheck",{}).get("ok"), d.get("parity_selfcheck",{}).get("clip0_vs_reference",{}).get("ce_delta"))
    print({k:(round(v["avg_launch_ms"],4),round(v["share_of_step_time"],4)) for k,v in d["kernel_classes"].items()})
PY
head -14 gpurun_out/${TAG}_bf16_kernel_stats.txt | cut -c1-150
