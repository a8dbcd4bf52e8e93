#!/usr/bin/env python3
"""Time the fused spatial attention + out-projection kernel on random operand planes (C-ABI unit entry).
   python tools/bench_spatial_fused.py [--clips 64]"""
import argparse, importlib, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--clips", type=int, default=64); ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    _lib = importlib.import_module("1xgpt_amd._lib"); cfgmod = importlib.import_module("1xgpt_amd.config")
    lib = _lib.load(); cfg = _lib.make_cfg(cfgmod.c35(), _lib.PREC_BF16)
    n_seq = a.clips * 16; rows = n_seq * 256
    g = torch.Generator(device="cuda").manual_seed(0)
    planes = (torch.randn(3, rows, 256, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    x = torch.randn(rows, 256, device="cuda", generator=g); x16 = torch.empty(rows, 256, dtype=torch.bfloat16, device="cuda")
    pw = torch.randn(256, 256, device="cuda", generator=g) * .05; pb = torch.randn(256, device="cuda", generator=g) * .01
    st = torch.cuda.current_stream().cuda_stream
    sf = torch.empty(_lib.SPATIAL_PROJ_FUSED_ELEMS, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.genie_pack_spatial_proj_fused_bf16(pw.data_ptr(), sf.data_ptr(), st), "pack")
    aw = _lib.AttnWeights(); aw.fused_w16 = sf.data_ptr(); aw.proj_b = pb.data_ptr()
    f = lambda: _lib.check(lib.genie_spatial_attn_proj_fused_bf16(cfg, aw, planes.data_ptr(), x.data_ptr(), x16.data_ptr(), n_seq, st), "s")
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.iters * 1e3
    fl = rows * (4.0 * 256 * 256 + 2.0 * 256 * 256); by = rows * (3 * 512.0 + 2048.0 + 512.0)
    print(f"spatial_attn_proj: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  {by / us / 1e6:6.2f} TB/s (algorithmic)")

if __name__ == "__main__":
    main()
