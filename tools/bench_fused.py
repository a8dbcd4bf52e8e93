#!/usr/bin/env python3
"""Time the fused sub-block kernels of the shipped geometry on random data through their C-ABI unit entry points.
   python tools/bench_fused.py [--clips 64] [--iters 20]      (GENIE_HIP_LIBRARY=<study build> GENIE_FUSED_ABL=<bits> for ablations)"""
import argparse
import importlib
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    _lib = importlib.import_module("1xgpt_amd._lib")
    cfgmod = importlib.import_module("1xgpt_amd.config")
    lib = _lib.load()
    c = cfgmod.c35()
    cfg = _lib.make_cfg(c, _lib.PREC_BF16)
    B, rows = a.clips, a.clips * 4096
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    x = r(rows, 256)
    x16 = x.to(torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    qkv_w, proj_w, fc1_w, fc2_w = r(768, 256) * 0.05, r(256, 256) * 0.05, r(1024, 256) * 0.05, r(256, 1024) * 0.03
    tf = torch.empty(_lib.TEMPORAL_FUSED_ELEMS, dtype=torch.bfloat16, device="cuda")
    mf = torch.empty(_lib.MLP_FUSED_ELEMS, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.genie_pack_temporal_fused_bf16(qkv_w.data_ptr(), proj_w.data_ptr(), tf.data_ptr(), st), "pack_t")
    _lib.check(lib.genie_pack_mlp_fused_bf16(fc1_w.data_ptr(), fc2_w.data_ptr(), mf.data_ptr(), st), "pack_m")
    aw = _lib.AttnWeights()
    aw.fused_w16 = tf.data_ptr()
    pb, b1, b2, lg, lb = r(256) * 0.01, r(1024) * 0.01, r(256) * 0.01, torch.ones(256, device="cuda"), torch.zeros(256, device="cuda")
    aw.proj_b = pb.data_ptr()
    lw = _lib.LayerWeights()
    lw.mlp_fused_w16 = mf.data_ptr()
    lw.norm2_w, lw.norm2_b, lw.fc1_b, lw.fc2_b = lg.data_ptr(), lb.data_ptr(), b1.data_ptr(), b2.data_ptr()

    def timed(name, fn, flops, bytes_):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.iters * 1e3
        print(f"{name}: {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  {bytes_ / us / 1e6:6.2f} TB/s (algorithmic)", flush=True)

    timed("temporal_fused", lambda: _lib.check(lib.genie_temporal_fused_bf16(cfg, aw, x16.data_ptr(), x.data_ptr(), B, st), "t"),
          rows * (2.0 * 256 * 1024 + 4.0 * 16 * 256), rows * 2560.0)
    x.copy_(r(rows, 256))
    timed("mlp_fused     ", lambda: _lib.check(lib.genie_mlp_fused_bf16(cfg, lw, x.data_ptr(), 0, rows, 0, 0, st), "m"),
          rows * 4.0 * 256 * 1024, rows * 2048.0)
    x.copy_(r(rows, 256))
    timed("mlp_fused +x16", lambda: _lib.check(lib.genie_mlp_fused_bf16(cfg, lw, x.data_ptr(), x16.data_ptr(), rows, 0, 0, st), "m"),
          rows * 4.0 * 256 * 1024, rows * 2560.0)
    x.copy_(r(rows, 256))
    timed("mlp_fused +LN ", lambda: _lib.check(lib.genie_mlp_fused_bf16(cfg, lw, x.data_ptr(), x16.data_ptr(), rows, lg.data_ptr(),
                                                                         lb.data_ptr(), st), "m"),
          rows * 4.0 * 256 * 1024, rows * 2560.0)
    # mode 2: also the next block's norm1 + spatial qkv planes
    wq = r(768, 256) * 0.05
    sf = torch.zeros(_lib.SPATIAL_PROJ_FUSED_ELEMS + _lib.SPATIAL_QKV_FUSED_ELEMS, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.genie_pack_spatial_qkv_fused_bf16(wq.data_ptr(), sf.data_ptr() + 2 * _lib.SPATIAL_PROJ_FUSED_ELEMS, st), "pack_q")
    nx = _lib.LayerWeights()
    nx.norm1_w, nx.norm1_b = lg.data_ptr(), lb.data_ptr()
    nx.spatial.fused_w16, nx.spatial.w16_wide = sf.data_ptr(), _lib.FUSED_QKV_STREAM
    planes = torch.empty(3, rows, 256, dtype=torch.bfloat16, device="cuda")
    x.copy_(r(rows, 256))
    timed("mlp_fused +QKV", lambda: _lib.check(lib.genie_mlp_fused_qkv_bf16(cfg, lw, nx, x.data_ptr(), planes.data_ptr(), rows, st), "mq"),
          rows * (4.0 * 256 * 1024 + 2.0 * 256 * 768), rows * (2048.0 + 1536.0))


if __name__ == "__main__":
    main()
