#!/usr/bin/env python3
"""BASELINE config 2: STMaskGIT.forward (+ masked factored CE) on a batch of synthetic clips, one GPU.
   python tools/bench_forward.py --model c35 --precision bf16 --batch 64"""
import argparse
import importlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="c35", choices=["c35", "c138"])
    ap.add_argument("--precision", nargs="+", default=["bf16", "f16x3", "exact"])
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    cfgmod = importlib.import_module("1xgpt_amd.config")
    synth = importlib.import_module("1xgpt_amd.synthetic")
    STMaskGIT = importlib.import_module("1xgpt_amd.st_mask_git").STMaskGIT
    cfg = cfgmod.c138() if a.model == "c138" else cfgmod.c35()
    d, L = cfg.d_model, cfg.num_layers
    flops_clip = 4096 * (L * (32 * d * d + 4 * cfg.S * d + 4 * cfg.T * d) + 2 * d * 1024)
    sd = synth.make_state_dict(cfg, seed=0)
    ids = torch.from_numpy(synth.make_clips(a.batch, cfg, seed=1)).cuda()
    x = ids.clone().view(a.batch, cfg.T, -1)
    x[:, 8:] = cfg.image_vocab_size  # frames >= 8 masked, so the masked CE is defined
    x = x.view(a.batch, -1)
    res = []
    for prec in a.precision:
        m = STMaskGIT(cfg, precision=prec).load_numpy_state_dict(sd).to("cuda")
        for name, fn in (("forward (logits materialised, as the reference)", lambda: m(x, ids)),
                         ("ce_sums (readout fused into the CE, logits never stored)",
                          lambda: m.ce_sums(x.view(a.batch, cfg.T, 16, 16), ids.view(a.batch, cfg.T, 16, 16)))):
            out = fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.iters):
                out = fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.iters
            loss = float(out.loss) if hasattr(out, "loss") else float(out[0] / out[2])
            res.append({"model": a.model, "precision": prec, "batch": a.batch, "path": name, "ms": dt * 1e3,
                        "clips_per_s": a.batch / dt, "tokens_per_s": a.batch * 4096 / dt,
                        "model_tflops": flops_clip * a.batch / dt / 1e12, "loss": loss})
            print(res[-1], flush=True)
        del m
        torch.cuda.empty_cache()
    print(json.dumps({"workload": "forward + CE (BASELINE config 2)", "results": res}))


if __name__ == "__main__":
    main()
