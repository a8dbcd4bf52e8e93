#!/usr/bin/env python3
"""generate.py workload (BASELINE config 3): prompt 8 frames -> sample 8 frames, MaskGIT steps 2 and 8, temperature 0,
GENIE_138M-shape, full-forward schedule vs temporal KV cache, at batch 1 (the reference's CLI) and batched."""
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
    ap.add_argument("--precision", default="f16x3")
    ap.add_argument("--model", default="c138")
    ap.add_argument("--batches", type=int, nargs="+", default=[1, 16])
    ap.add_argument("--steps", type=int, nargs="+", default=[2, 8])
    ap.add_argument("--schedules", nargs="+", default=["full_forward", "kv_cache"], choices=["full_forward", "kv_cache"])
    a = ap.parse_args()
    cfgmod = importlib.import_module("1xgpt_amd.config")
    synth = importlib.import_module("1xgpt_amd.synthetic")
    G = importlib.import_module("1xgpt_amd.generate")
    STMaskGIT = importlib.import_module("1xgpt_amd.st_mask_git").STMaskGIT
    cfg = cfgmod.c138() if a.model == "c138" else cfgmod.c35()
    m = STMaskGIT(cfg, precision=a.precision).load_numpy_state_dict(synth.make_state_dict(cfg, seed=0)).to("cuda")
    res = []
    for B in a.batches:
        ex = torch.from_numpy(synth.make_clips(B, cfg, seed=7)).cuda().view(B, 16, 16, 16)
        for steps in a.steps:
            noise = torch.rand(8, max(steps - 1, 1), B, cfg.S, device="cuda")
            for name, fn in (("full_forward", G.generate_frames), ("kv_cache", G.generate_frames_cached)):
                if name not in a.schedules or (name == "full_forward" and B * steps > 64):
                    continue
                for _ in range(2):   # (the cached schedule captures its HIP graphs on the second call)
                    fn(m, ex, 8, steps, 0.0, False, noise=noise)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                reps = 3
                for _ in range(reps):
                    out = fn(m, ex, 8, steps, 0.0, False, noise=noise)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / reps
                res.append({"schedule": name, "batch": B, "maskgit_steps": steps, "seconds": dt,
                            "frames_per_sec": 8 * B / dt, "s_per_frame": dt / (8 * B)})
                print(res[-1], flush=True)
    print(json.dumps({"workload": "generate 8->8 frames, " + a.model + " " + a.precision, "results": res}))


if __name__ == "__main__":
    main()
