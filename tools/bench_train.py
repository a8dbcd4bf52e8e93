#!/usr/bin/env python3
"""Time the HIP training step (forward + backward + clip + AdamW) on synthetic clips.
   python tools/bench_train.py --config c138 --batch 8 --steps 3"""
import argparse
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = lambda n: importlib.import_module("1xgpt_amd" + ("." + n if n else ""))  # noqa: E731


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c138", choices=["c35", "c138"])
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--breakdown", action="store_true")
    ap.add_argument("--precision", default="exact", choices=["exact", "f16x3", "bf16"])
    a = ap.parse_args()
    cfgm, syn, _lib = pkg("config"), pkg("synthetic"), pkg("_lib")
    cfg = cfgm.c138() if a.config == "c138" else cfgm.c35()
    cfg.qk_norm = False
    if a.layers:
        cfg.num_layers = a.layers
    sd = syn.make_state_dict(cfg, seed=0, law="init")
    model = pkg("st_mask_git").STMaskGIT(cfg, precision=a.precision).load_numpy_state_dict(sd).to("cuda")
    tr = pkg("train").GenieTrainer(model, lr=1e-4, max_grad_norm=1.0)
    ids = torch.from_numpy(syn.make_clips(a.batch, cfg, seed=1)).cuda()
    collate = pkg("data").maskgit_collate
    torch.manual_seed(0)
    import random
    random.seed(0)
    batch = collate(ids, cfg)
    lib = _lib.load()
    for _ in range(a.warmup):
        out = tr.train_step(batch)
    torch.cuda.synchronize()
    if a.breakdown:
        lib.genie_profile_enable(0x1F)
        lib.genie_profile_reset()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = tr.train_step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    tokens = a.batch * cfg.T * cfg.S
    n_params = sum(p.numel() for p in model.parameters())
    flops = 6.0 * n_params * tokens  # the reference's own estimate (train.py:543)
    print(f"{a.config} L={cfg.num_layers} B={a.batch}: {dt*1e3:.1f} ms/step, {tokens/dt:.0f} tokens/s, "
          f"{flops/dt/1e12:.1f} TFLOP/s (6ND), loss {float(out['loss']):.4f}, |g| {float(out['grad_norm']):.4f}, "
          f"acts {tr._acts.numel()/2**30:.1f} GiB, ws {tr._ws.numel()/2**30:.1f} GiB")
    if a.breakdown:
        import ctypes as C
        buf = (C.c_double * 4)()
        for cls, name in enumerate(["gemm", "attn_spatial", "attn_temporal", "layernorm", "other"]):
            lib.genie_profile_read(cls, buf)
            if buf[0]:
                print(f"  {name:14s} launches {int(buf[0]):6d}  {buf[1]/a.steps:9.2f} ms/step  "
                      f"{buf[2]/max(buf[1],1e-9)/1e9:8.1f} TFLOP/s  {buf[3]/max(buf[1],1e-9)/1e6:8.1f} GB/s")
        lib.genie_profile_enable(0)


if __name__ == "__main__":
    main()
