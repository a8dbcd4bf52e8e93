#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc CSV output (p_counter_collection.csv) to per-kernel averages.
usage: tools/pmc_csv_summary.py counter_collection.csv OUT.json [name-substring ...]"""
import csv
import json
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def main():
    path, out = sys.argv[1], sys.argv[2]
    filt = sys.argv[3:]
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0, 0.0]))
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if filt and not any(s in name for s in filt):
                continue
            short = name.split("(")[0][-70:] + "|grid=" + row["Grid_Size"]
            a = acc[short][row["Counter_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
            a[2] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    res = {k: {c: {"dispatches": v[0], "avg": v[1] / v[0], "avg_ns_under_pmc": v[2] / v[0]} for c, v in d.items()}
           for k, d in acc.items()}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    for k, d in res.items():
        print(k, {c: (v["dispatches"], round(v["avg"], 1)) for c, v in d.items()})


if __name__ == "__main__":
    main()
