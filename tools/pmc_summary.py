#!/usr/bin/env python3
"""Reduce a rocprofv3 --pmc rocpd database to per-kernel counter averages (JSON).
usage: tools/pmc_summary.py IN.db OUT.json [substring-of-kernel-name ...]"""
import json
import sqlite3
import sys


def summarize(db, filters=()):
    c = sqlite3.connect(db)
    rows = c.execute("select name, counter_name, count(*), sum(counter_value), avg(counter_value), avg(duration) "
                     "from pmc_events group by name, counter_name").fetchall()
    out = {}
    for name, counter, n, tot, avg, dur in rows:
        if filters and not any(f in name for f in filters):
            continue
        out.setdefault(name[:120], {})[counter] = {"dispatches": n, "sum": tot, "avg_per_dispatch": avg,
                                                    "avg_duration_ns_under_pmc": dur}
    return out


if __name__ == "__main__":
    res = summarize(sys.argv[1], sys.argv[3:])
    with open(sys.argv[2], "w") as f:
        json.dump(res, f, indent=1)
    for k, v in res.items():
        print(k[:80], {c: round(x["avg_per_dispatch"], 1) for c, x in v.items()})
