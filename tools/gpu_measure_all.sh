#!/bin/bash
# The round's measurement set on one box: default bench (JSON line), rocprofv3 kernel trace and PMC passes of the same
# command, generate (config 3), forward+CE (config 2), end-to-end (config 5).  usage: tools/gpu_measure_all.sh <tag>
TAG=${1:-r03}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gpu_tests.txt 2>&1; tail -3 gpurun_out/${TAG}_gpu_tests.txt
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
bash tools/gpu_profile_bench.sh ${TAG}_f16x3 > gpurun_out/${TAG}_profile.log 2>&1
bash tools/gpu_profile_bench.sh ${TAG}_bf16 --precision bf16 > gpurun_out/${TAG}_profile_bf16.log 2>&1
bash tools/gpu_profile_bench.sh ${TAG}_exact --precision exact --batch 64 > gpurun_out/${TAG}_profile_exact.log 2>&1
bash tools/gpu_pmc_bench.sh ${TAG} f16x3 128 > gpurun_out/${TAG}_pmc.log 2>&1
bash tools/gpu_pmc_bench.sh ${TAG} bf16 128 > gpurun_out/${TAG}_pmc_bf16.log 2>&1
python tools/bench_generate.py --batches 1 8 16 --steps 2 8 > gpurun_out/${TAG}_generate.txt 2>&1
python tools/bench_forward.py > gpurun_out/${TAG}_forward.txt 2>&1
python tools/bench_e2e.py > gpurun_out/${TAG}_e2e.json 2> gpurun_out/${TAG}_e2e.err
tail -c 300 gpurun_out/${TAG}_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench.json"))
print("value",d["value"],"ms",d["ms_per_step"],"ce",d["ce"],"gemm TF",d["roofline"]["achieved"],"frac",d["roofline"]["frac"],"share",d["roofline"]["gemm_share_of_step_time"],"traffic",d["roofline"].get("traffic"))
print({k:(round(v["avg_launch_ms"],4),round(v["share_of_step_time"],4)) for k,v in d["kernel_classes"].items()})
print("full",d["full_forward_schedule"]["value"],"train",d["train_step"].get("value"),"cpu",d["cpu_baseline"]["value"],"bf16",d["bf16_evaluate"].get("value"))
print("selfcheck", json.dumps(d.get("parity_selfcheck"))[:700])
PY
