#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace rocpd database (or *_kernel_stats.csv) into a small text table:
per kernel: launches, total ms, average us, share.  Usage: tools/rocprof_summary.py IN.db [OUT.txt]"""
import sqlite3
import sys


def summarize(db):
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
        "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = [f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'share':>6s}"]
    for n, k, ms, avg, mn, mx in rows:
        lines.append(f"{n[:90]:90s} {k:7d} {ms:10.3f} {avg:9.1f} {mn:9.1f} {mx:9.1f} {100 * ms / tot:5.1f}%")
    lines.append(f"{'TOTAL':90s} {sum(r[1] for r in rows):7d} {tot:10.3f}")
    return "\n".join(lines)


if __name__ == "__main__":
    txt = summarize(sys.argv[1])
    if len(sys.argv) > 2:
        with open(sys.argv[2], "a") as f:
            f.write(txt + "\n")
    print(txt)
