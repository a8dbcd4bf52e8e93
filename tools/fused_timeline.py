#!/usr/bin/env python3
"""Per-block phase timeline of a fused sub-block kernel (study build, GENIE_FUSED_STAMPS=1): runs the kernel once on random data and
prints, per block index, the mean duration of [operand load | main loop | residual update] and how the phases of the two workgroups
that share a CU line up.   GENIE_HIP_LIBRARY=.../libgenie_hip_study.so GENIE_FUSED_STAMPS=1 python tools/fused_timeline.py t|m"""
import ctypes
import importlib
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "t"
    _lib = importlib.import_module("1xgpt_amd._lib")
    cfgmod = importlib.import_module("1xgpt_amd.config")
    lib = _lib.load()
    cfg = _lib.make_cfg(cfgmod.c35(), _lib.PREC_BF16)
    B, rows = 64, 64 * 4096
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    x = r(rows, 256)
    x16 = x.to(torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    tf = torch.empty(_lib.TEMPORAL_FUSED_ELEMS, dtype=torch.bfloat16, device="cuda")
    mf = torch.empty(_lib.MLP_FUSED_ELEMS, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.genie_pack_temporal_fused_bf16((r(768, 256) * .05).data_ptr(), (r(256, 256) * .05).data_ptr(), tf.data_ptr(), st), "p")
    _lib.check(lib.genie_pack_mlp_fused_bf16((r(1024, 256) * .05).data_ptr(), (r(256, 1024) * .03).data_ptr(), mf.data_ptr(), st), "p")
    aw, lw = _lib.AttnWeights(), _lib.LayerWeights()
    aw.fused_w16 = tf.data_ptr()
    lw.mlp_fused_w16 = mf.data_ptr()
    lg, lb = torch.ones(256, device="cuda"), torch.zeros(256, device="cuda")
    lw.norm2_w, lw.norm2_b = lg.data_ptr(), lb.data_ptr()
    run = (lambda: lib.genie_temporal_fused_bf16(cfg, aw, x16.data_ptr(), x.data_ptr(), B, st)) if which == "t" else \
          (lambda: lib.genie_mlp_fused_bf16(cfg, lw, x.data_ptr(), 0, rows, 0, 0, st))
    for _ in range(3):
        _lib.check(run(), "run")
    torch.cuda.synchronize()
    raw = lib._handle if hasattr(lib, "_handle") else None
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    buf = (ctypes.c_ulonglong * (37 * 512))()
    n = cdll.genie_study_fused_stamps(buf, 37 * 512)
    assert n > 0, "no stamps (study build + GENIE_FUSED_STAMPS=1?)"
    full = np.frombuffer(buf, dtype=np.uint64)
    a = full[:33 * 512].reshape(512, 33)
    cyc = full[33 * 512:].reshape(512, 4).astype(np.float64)
    live = a[:, 1] > 0                       # workgroups that ran (the MLP kernel launches one per CU, the temporal one two)
    a, cyc = a[live], cyc[live]
    NW = int(live.sum())
    tot = cyc.sum(1, keepdims=True)
    print(" wave 0 main-loop cycles by category (mean over workgroups): wait %.0f (%.0f%%)  dma issue %.0f (%.0f%%)  matrix section %.0f (%.0f%%)  "
          "valu section %.0f (%.0f%%)  total %.0f" % (tuple(v for c in range(4) for v in (cyc[:, c].mean(), 100 * (cyc[:, c] / tot[:, 0]).mean())) + (tot.mean(),)))
    ids, t = a[:, 0], a[:, 1:].reshape(NW, 8, 4).astype(np.int64)
    nb = int((t[0, :, 0] > 0).sum())
    t0 = t[:, 0, 0].min()
    us = lambda ticks: ticks / 100.0
    print(f"kernel {which}: {nb} blocks per workgroup; start skew over workgroups {us(t[:, 0, 0].max() - t0):.1f} us; last stamp at "
          f"{us(t[:, nb - 1, 3].max() - t0):.1f} us")
    for i in range(nb):
        d = t[:, i]
        print(f" block {i}: load {us((d[:, 1] - d[:, 0]).mean()):6.1f}  main {us((d[:, 2] - d[:, 1]).mean()):6.1f}  residual "
              f"{us((d[:, 3] - d[:, 2]).mean()):6.1f} us   (start spread p5..p95 {us(np.percentile(d[:, 0], 5) - t0):.1f}..{us(np.percentile(d[:, 0], 95) - t0):.1f})")
    # co-residency: workgroups with the same (xcc, se, cu) id
    hw = ids & np.uint64(0xFFFFFFFF)
    key = (ids >> np.uint64(32)).astype(np.int64) * 4096 + ((hw >> np.uint64(8)) & np.uint64(0xF)).astype(np.int64) * 64 + \
          ((hw >> np.uint64(13)) & np.uint64(0x7)).astype(np.int64) * 16  # xcc, cu_id (bits 8-11), se_id (13-15) -- best effort
    pairs = {}
    for w in range(NW):
        pairs.setdefault(int(key[w]), []).append(w)
    sizes = np.bincount([len(v) for v in pairs.values()])
    print(" workgroups per distinct hardware id:", {k: v for k, v in enumerate(sizes.tolist()) if v})
    st0 = np.sort(t[:, 0, 0] - t0)
    print(" first-block start times (us) p0/p25/p50/p75/p100:", [round(us(np.percentile(st0, q)), 1) for q in (0, 25, 50, 75, 100)],
          " end of last block p50/p100:", round(us(np.percentile(t[:, nb - 1, 3] - t0, 50)), 1), round(us((t[:, nb - 1, 3] - t0).max()), 1))
    off = []
    for v in pairs.values():
        if len(v) == 2:
            off.append(us(abs(int(t[v[0], 1, 0]) - int(t[v[1], 1, 0]))))
    if off:
        print(f" |start of block 1| difference between the two workgroups of a CU: mean {np.mean(off):.1f} us, p90 {np.percentile(off, 90):.1f} us")


if __name__ == "__main__":
    main()
