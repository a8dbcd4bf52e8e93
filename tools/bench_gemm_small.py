"""GEMM micro-benchmark at one-frame-pass sizes (M = 256 * clips rows): which kernel the dispatcher picks and how long it takes."""
import argparse, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_lib = importlib.import_module("1xgpt_amd._lib")
ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, nargs="+", default=[8, 16, 32])
ap.add_argument("--prec", default="f16x3")
ap.add_argument("--acc", type=int, default=0)
a = ap.parse_args()
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
code = _lib.PREC_BF16 if a.prec == "bf16" else _lib.PREC_F16X3
npl = 1 if a.prec == "bf16" else 2
pack = lib.genie_pack_bf16 if a.prec == "bf16" else lib.genie_pack_split_f16
for clips in a.clips:
    M = 256 * clips
    for name, N, K in (("qkv", 1536, 512), ("proj", 512, 512), ("fc1", 2048, 512), ("fc2", 512, 2048)):
        x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda"); y = torch.zeros(M, N, device="cuda")
        x16 = torch.empty(npl, M, K, dtype=torch.float16, device="cuda"); W16 = torch.empty(npl, N, K, dtype=torch.float16, device="cuda")
        _lib.check(pack(x.data_ptr(), x16.data_ptr(), x.numel(), st), "pack"); _lib.check(pack(W.data_ptr(), W16.data_ptr(), W.numel(), st), "pack")
        def call():
            _lib.check(lib.genie_linear_lowp(code, x16.data_ptr(), W16.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, 0, a.acc, st), "lin")
        for _ in range(5): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): call()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 200
        print(f"{a.prec} M={M:5d} {name:5s} N={N:4d} K={K:4d} {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TFLOP/s", flush=True)
