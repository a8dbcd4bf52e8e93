#!/usr/bin/env python3
"""Reduce the three rocprofv3 --pmc passes of bench.py (tools/gpu_pmc_bench.sh) to profiles/pmc_bench.json:
per kernel class (gemm / attention_spatial / attention_temporal) HBM-side bytes per launch, matrix-pipe busy fraction and
shader clock.  usage: tools/pmc_bench_summary.py <precision> <clips> <fetch.csv> <write.csv> <sq.csv> <out.json> "<source note>"

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies
128-byte requests as 64 bytes, so the read side is doubled.  SQ_BUSY_CYCLES sums the 32 shader engines, GRBM_GUI_ACTIVE the
8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES sums busy cycles over the 1024 SIMDs."""
import csv
import json
import os
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
CLASSES = {"gemm": ("gemm16_", "gemm_f32"), "attention_spatial": ("attn_spatial",), "attention_temporal": ("attn_temporal",),
           "fused_mlp": ("mlp_fused_bf16",), "fused_spatial": ("spatial_attn_proj_bf16",), "fused_temporal": ("temporal_fused_bf16", "temporal_prefix_fused_bf16", "temporal_qkv_attn_f16x3"),
           "layernorm": ("layer_norm_fast",)}


def load(path):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0, 0.0]))
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            cls = next((c for c, pats in CLASSES.items() if any(p in name for p in pats)), None)
            if cls is None:
                continue
            a = acc[cls][row["Counter_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
            a[2] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    return acc


def main():
    prec, clips, f_fetch, f_write, f_sq, out, note = sys.argv[1], int(sys.argv[2]), *sys.argv[3:8]
    fetch, write, sq = load(f_fetch), load(f_write), load(f_sq)
    res = {}
    for cls in CLASSES:
        if cls not in fetch:
            continue
        nf, vf, _ = fetch[cls]["FETCH_SIZE"]
        nw, vw, _ = write[cls]["WRITE_SIZE"]
        rd = vf / nf * 1024.0 * 2.0
        wr = vw / nw * 1024.0
        e = {"launches_in_pass": nf, "clips": clips, "fetch_bytes_per_launch_x2_corrected": rd, "write_bytes_per_launch": wr,
             "hbm_bytes_per_launch": rd + wr, "hbm_bytes_per_launch_per_clip": (rd + wr) / clips}
        if cls in sq and "SQ_VALU_MFMA_BUSY_CYCLES" in sq[cls]:
            n, mf, ns = sq[cls]["SQ_VALU_MFMA_BUSY_CYCLES"]
            gui = sq[cls]["GRBM_GUI_ACTIVE"][1] / 8.0            # cycles of one XCD, summed over the launches
            e["mfma_busy_frac"] = mf / 1024.0 / gui
            e["sclk_ghz"] = gui / ns                             # cycles per ns while these kernels ran (under PMC)
            e["avg_launch_us_under_pmc"] = ns / n / 1e3
            e["note"] = ("matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); sclk = "
                         "GRBM_GUI_ACTIVE / 8 / kernel time: the board runs these kernels well below the 2.4 GHz the "
                         "2.5 PFLOP/s peak is quoted at")
        res[cls] = e
    doc = {}
    if os.path.exists(out):
        with open(out) as f:
            doc = json.load(f)
    doc.setdefault("_sources", {})[prec] = note   # each precision section is its own set of three --pmc passes
    doc["_source"] = "per precision section: see _sources"
    doc[prec] = res
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
