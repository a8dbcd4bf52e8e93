#!/usr/bin/env python3
"""Reference point for DESIGN.md section 5 (not part of the product path): the vendor GEMM (torch.matmul -> hipBLASLt) on the same
board, bf16 and f16, random vs zero-filled operands.  If the board's power-managed clock is what caps 16-bit MFMA throughput on
real data, the vendor kernel shows the same two levels as gemm16_pp does."""
import json
import sys

import torch


def run(n, dtype, zero, iters=30):
    g = torch.Generator(device="cuda").manual_seed(n)
    a = torch.randn(n, n, device="cuda", generator=g).to(dtype)
    b = torch.randn(n, n, device="cuda", generator=g).to(dtype)
    if zero:
        a.zero_(); b.zero_()
    for _ in range(3):
        (a @ b.T)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        (a @ b.T)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return 2.0 * n ** 3 / ms / 1e9


if __name__ == "__main__":
    out = []
    for n in (4096, 8192):
        for dtype, name in ((torch.bfloat16, "bf16"), (torch.float16, "f16")):
            r, z = run(n, dtype, False), run(n, dtype, True)
            out.append({"n": n, "dtype": name, "random_TFLOPs": round(r, 1), "zero_TFLOPs": round(z, 1)})
            print(out[-1], flush=True)
    json.dump(out, sys.stdout)
    print()
