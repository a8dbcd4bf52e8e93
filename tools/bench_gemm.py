#!/usr/bin/env python3
"""GEMM micro-benchmark through the C ABI (genie_linear / genie_linear_lowp): the layer shapes of the C138-shape
model at M = 4096*B tokens, every precision, HIP-event timed, random operands (never zero-filled)."""
import argparse
import importlib
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
_lib = importlib.import_module("1xgpt_amd._lib")


def run(prec, M, N, K, iters=20, check=True, gelu=0, zero=False, acc=0, pad_a=0, pad_c=0):
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    if zero:  # power study only: all-zero operands toggle no multiplier inputs (never quote these numbers as throughput)
        x.zero_(); W.zero_()
    y = torch.empty(M, N + pad_c, device="cuda")[:, :N]   # (study build) padded output rows: GENIE_STUDY_LDC_PAD
    st = torch.cuda.current_stream().cuda_stream
    if prec == "exact":
        def call():
            _lib.check(lib.genie_linear(x.data_ptr(), W.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, gelu, acc, st), "lin")
    else:
        code = _lib.PREC_BF16 if prec == "bf16" else _lib.PREC_F16X3
        npl = 1 if prec == "bf16" else 2
        x16 = torch.empty(npl, M, K, dtype=torch.float16, device="cuda")
        W16 = torch.empty(npl, N, K, dtype=torch.float16, device="cuda")
        pack = lib.genie_pack_bf16 if prec == "bf16" else lib.genie_pack_split_f16
        _lib.check(pack(x.data_ptr(), x16.data_ptr(), x.numel(), st), "pack")
        _lib.check(pack(W.data_ptr(), W16.data_ptr(), W.numel(), st), "pack")
        if pad_a:  # (study build) activation planes with padded rows: GENIE_STUDY_LDA_PAD
            xp = torch.zeros(npl, M, K + pad_a, dtype=torch.float16, device="cuda")
            xp[:, :, :K] = x16
            x16 = xp

        def call():
            _lib.check(lib.genie_linear_lowp(code, x16.data_ptr(), W16.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, gelu, acc,
                                             st), "lin16")
    y.zero_()
    call()
    torch.cuda.synchronize()
    err = None
    if check:
        rows = torch.randint(0, M, (64,), device="cuda")
        ref = x[rows].double() @ W.double().T + b.double()
        if gelu:
            ref = torch.nn.functional.gelu(ref)
        err = (y[rows].double() - ref).abs().max().item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / ms / 1e9, err


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--prec", nargs="+", default=["exact", "f16x3", "bf16"])
    ap.add_argument("--gelu", type=int, default=0, help="fused erf-GELU epilogue")
    ap.add_argument("--acc", type=int, default=0, help="residual accumulate (y += ...) epilogue")
    ap.add_argument("--zero", action="store_true", help="zero-filled operands (clock / power study, not a throughput figure)")
    ap.add_argument("--shapes", nargs="*", default=None, help="N:K pairs instead of the model's shapes, e.g. 1280:512 1792:512")
    ap.add_argument("--rows", type=int, default=None, help="M (default 4096 * batch)")
    ap.add_argument("--pad-a", type=int, default=0, help="study build: extra elements per activation row (sets GENIE_STUDY_LDA_PAD)")
    ap.add_argument("--pad-c", type=int, default=0, help="study build: extra floats per output row (sets GENIE_STUDY_LDC_PAD)")
    a = ap.parse_args()
    if a.pad_a:
        os.environ["GENIE_STUDY_LDA_PAD"] = str(a.pad_a)
    if a.pad_c:
        os.environ["GENIE_STUDY_LDC_PAD"] = str(a.pad_c)
    M = a.rows or 4096 * a.batch
    shapes = [("qkv", 1536, 512), ("proj", 512, 512), ("fc1", 2048, 512), ("fc2", 512, 2048), ("readout", 1024, 512),
              ("sq4096", 4096, 4096)]
    if a.shapes:
        shapes = [(f"n{s}", int(s.split(":")[0]), int(s.split(":")[1])) for s in a.shapes]
    for prec in a.prec:
        for name, N, K in shapes:
            m = 4096 if name == "sq4096" else M
            ms, tf, err = run(prec, m, N, K, gelu=a.gelu, zero=a.zero, acc=a.acc,
                              pad_a=a.pad_a if prec != "exact" else 0, pad_c=a.pad_c if prec != "exact" else 0)
            print(f"{prec:6s} {name:8s} M={m:6d} N={N:5d} K={K:5d}  {ms * 1e3:9.1f} us  {tf:8.1f} TFLOP/s  max|err| {err:.2e}",
                  flush=True)
