#!/usr/bin/env python3
"""Idle time between kernels in a rocprofv3 --kernel-trace rocpd database: for the latency-bound one-frame passes of generate the
question is how much of a pass is kernels and how much is the gap between dependent launches.
Usage: tools/rocprof_gaps.py IN.db [OUT.txt]   (bursts = runs of launches separated by < 200 us)"""
import sqlite3
import sys


def analyse(db):
    c = sqlite3.connect(db)
    rows = c.execute("select start, end, name from kernels order by start").fetchall()
    if not rows:
        return "no kernels"
    bursts, cur = [], [rows[0]]
    for r in rows[1:]:
        if r[0] - cur[-1][1] > 200_000:
            bursts.append(cur)
            cur = [r]
        else:
            cur.append(r)
    bursts.append(cur)
    big = [b for b in bursts if len(b) >= 100]
    out = [f"{len(rows)} kernel launches, {len(bursts)} bursts, {len(big)} with >= 100 launches"]
    tot_span = tot_busy = tot_n = 0
    gaps = []
    for b in big:
        span = b[-1][1] - b[0][0]
        busy = sum(e - s for s, e, _ in b)
        tot_span += span; tot_busy += busy; tot_n += len(b)
        gaps += [max(0, b[i + 1][0] - b[i][1]) for i in range(len(b) - 1)]
    if big:
        gaps.sort()
        out.append(f"in those bursts: span {tot_span / 1e6:.2f} ms, kernels {tot_busy / 1e6:.2f} ms = {100 * tot_busy / tot_span:.1f} % busy, "
                   f"{tot_n} launches, mean kernel {tot_busy / tot_n / 1e3:.2f} us, mean gap {sum(gaps) / len(gaps) / 1e3:.2f} us "
                   f"(median {gaps[len(gaps) // 2] / 1e3:.2f}, p90 {gaps[int(len(gaps) * 0.9)] / 1e3:.2f})")
    return "\n".join(out)


if __name__ == "__main__":
    txt = analyse(sys.argv[1])
    if len(sys.argv) > 2:
        with open(sys.argv[2], "a") as f:
            f.write(txt + "\n")
    print(txt)
