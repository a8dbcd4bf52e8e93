#!/usr/bin/env python3
"""Per kernel NAME (not class): launches, average duration, shader clock (GRBM_GUI_ACTIVE / 8 XCDs / kernel time) and matrix-pipe busy
fraction (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles) from one rocprofv3 --pmc pass (csv).
usage: tools/pmc_per_kernel.py <counter_collection.csv> [out.txt]"""
import csv
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0, 0.0]))
with open(sys.argv[1], newline="") as f:
    for row in csv.DictReader(f):
        a = acc[row["Kernel_Name"]][row["Counter_Name"]]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
        a[2] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
rows = []
for name, c in acc.items():
    if "GRBM_GUI_ACTIVE" not in c:
        continue
    n, gui, ns = c["GRBM_GUI_ACTIVE"]
    gui /= 8.0
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0, 0.0, 0.0])[1]
    rows.append((ns, name[:110], n, ns / n / 1e3, gui / ns, mf / 1024.0 / gui if gui else 0.0))
rows.sort(reverse=True)
out = ["%-110s %6s %10s %8s %9s" % ("kernel", "calls", "avg_us", "sclk_GHz", "mfma_busy")]
for ns, name, n, us, clk, busy in rows[:24]:
    out.append("%-110s %6d %10.1f %8.3f %9.3f" % (name, n, us, clk, busy))
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
