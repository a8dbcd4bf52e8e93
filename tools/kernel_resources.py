#!/usr/bin/env python3
"""Registers / scratch / occupancy of every kernel of one .hip source, from hipcc's -Rpass-analysis=kernel-resource-usage.
    python tools/kernel_resources.py 1xgpt_amd/csrc/kernels_frame.hip [extra -D flags]      (no GPU needed)"""
import re, subprocess, sys
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-x", "hip", "-c", src, "-o", "/dev/null", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
       "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: +(?:Function )?Name: (\S+)", line) or re.search(r"Name: (\S+)", line)
    if m and "Name:" in line:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    m = re.search(r"(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).split(" [")[0]] = int(m.group(2))
print(f"{'kernel':100s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'scratch':>7s} {'vspill':>6s} {'occ':>4s}")
for r in rows:
    print(f"{r['name'][:100]:100s} {r.get('VGPRs', -1):5d} {r.get('AGPRs', -1):5d} {r.get('TotalSGPRs', -1):5d} {r.get('ScratchSize', -1):7d} "
          f"{r.get('VGPRs Spill', -1):6d} {r.get('Occupancy', -1):4d}")
