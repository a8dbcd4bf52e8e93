#!/bin/bash
# BASELINE config 2 under rocprofv3: kernel trace + three PMC passes of tools/bench_forward.py.
# usage: tools/gpu_profile_forward.sh <tag> [bench_forward args, default: --model c35 --precision bf16 --batch 64 --iters 3]
TAG=${1:-r04_c35}; shift
ARGS=${@:-"--model c35 --precision bf16 --batch 64 --iters 3"}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/${TAG}_trace -o trace -- python3 $R/tools/bench_forward.py $ARGS > $R/gpurun_out/${TAG}_out.txt 2> $R/gpurun_out/${TAG}_rocprof.err
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_pf --output-format csv -- python3 $R/tools/bench_forward.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_pw --output-format csv -- python3 $R/tools/bench_forward.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_ps --output-format csv -- python3 $R/tools/bench_forward.py $ARGS > /dev/null 2>&1
cd $R
db=$(find gpurun_out/${TAG}_trace -name "*.db" | head -1)
echo "# cd /tmp && rocprofv3 --kernel-trace -- python3 tools/bench_forward.py $ARGS" > gpurun_out/${TAG}_kernel_stats.txt
python tools/rocprof_summary.py "$db" gpurun_out/${TAG}_kernel_stats.txt | head -40
for p in pf pw ps; do
  f=$(find gpurun_out/${TAG}_$p -name "*counter_collection.csv" | head -1)
  python tools/pmc_csv_summary.py "$f" gpurun_out/${TAG}_pmc_$p.json > /dev/null
done
python - <<PY
import json
pf=json.load(open("gpurun_out/${TAG}_pmc_pf.json")); pw=json.load(open("gpurun_out/${TAG}_pmc_pw.json")); ps=json.load(open("gpurun_out/${TAG}_pmc_ps.json"))
out={"_source":"rocprofv3 --pmc {FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE} (three passes, csv) -- python3 tools/bench_forward.py $ARGS; FETCH_SIZE in KiB x2 (gfx950 correction), WRITE_SIZE in KiB","kernels":{}}
for k in pf:
    e={"dispatches":pf[k]["FETCH_SIZE"]["dispatches"],"read_MB":pf[k]["FETCH_SIZE"]["avg"]*2*1024/1e6}
    if k in pw: e["write_MB"]=pw[k]["WRITE_SIZE"]["avg"]*1024/1e6
    if k in ps:
        gui=ps[k]["GRBM_GUI_ACTIVE"]["avg"]/8.0; ns=ps[k]["GRBM_GUI_ACTIVE"]["avg_ns_under_pmc"]
        e["mfma_busy_frac"]=ps[k]["SQ_VALU_MFMA_BUSY_CYCLES"]["avg"]/1024.0/gui if gui else None
        e["sclk_ghz"]=gui/ns if ns else None; e["avg_us_under_pmc"]=ns/1e3
    out["kernels"][k]=e
json.dump(out,open("gpurun_out/${TAG}_pmc.json","w"),indent=1)
for k,e in sorted(out["kernels"].items(), key=lambda kv:-kv[1].get("avg_us_under_pmc",0)*kv[1]["dispatches"])[:16]:
    print(k[-80:], {a:(round(b,3) if isinstance(b,float) else b) for a,b in e.items()})
PY
rm -rf gpurun_out/${TAG}_trace gpurun_out/${TAG}_pf gpurun_out/${TAG}_pw gpurun_out/${TAG}_ps gpurun_out/${TAG}_pmc_p?.json
tail -2 gpurun_out/${TAG}_out.txt | cut -c1-900
