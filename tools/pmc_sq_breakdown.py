#!/usr/bin/env python3
"""Per kernel name: where the wave cycles go, from ONE rocprofv3 --pmc pass with the SQ counters
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES
(8 SQ slots, MI355X_MICROARCH.md "rocprofv3 PMC slots": WAIT_ANY = parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stall (MFMA dependency /
pipe), ACTIVE_INST_* = issuing; the three are disjoint shares of WAVE_CYCLES, all in quad-cycles; MFMA_BUSY in cycles).
usage: tools/pmc_sq_breakdown.py <counter_collection.csv> [out.txt]"""
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
    if "SQ_WAVE_CYCLES" not in c:
        continue
    n, wc, ns = c["SQ_WAVE_CYCLES"]
    g = lambda k: c.get(k, [0, 0.0, 0.0])[1]
    rows.append((ns, name[:96], n, ns / n / 1e3, g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_WAIT_INST_LDS") / wc,
                 g("SQ_ACTIVE_INST_VALU") / wc, g("SQ_ACTIVE_INST_LDS") / wc, g("SQ_LDS_BANK_CONFLICT") / max(wc, 1.0),
                 g("SQ_VALU_MFMA_BUSY_CYCLES") / 4.0 / wc))
rows.sort(reverse=True)
out = ["# shares of SQ_WAVE_CYCLES (summed over all waves); mfma = SQ_VALU_MFMA_BUSY_CYCLES / 4 / WAVE_CYCLES (matrix-pipe cycles per wave quad-cycle)",
       "%-96s %5s %9s %8s %9s %8s %7s %7s %8s %6s" % ("kernel", "calls", "avg_us", "wait_any", "wait_inst", "w_i_lds", "valu", "lds", "lds_conf", "mfma")]
for r in rows[:16]:
    out.append("%-96s %5d %9.1f %8.3f %9.3f %8.3f %7.3f %7.3f %8.4f %6.3f" % r[1:])
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
