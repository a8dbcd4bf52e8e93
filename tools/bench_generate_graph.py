#!/usr/bin/env python3
"""generate (prompt 8 -> 8 new frames, KV cache) as ONE hipGraph: `genie_generate_cached` enqueues its whole loop on the caller's stream
without a host step, so the call can be captured (torch.cuda.CUDAGraph) and replayed -- does the graph close the gaps between the
5-6 us kernels of the one-frame passes (profiles/r05l_gen1_kernel_stats.txt: 76 % busy at batch 1)?
usage: tools/bench_generate_graph.py [--batches 1 8 16] [--steps 2 8] [--model c138]"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, nargs="+", default=[1, 8, 16])
    ap.add_argument("--steps", type=int, nargs="+", default=[2, 8])
    ap.add_argument("--model", default="c138")
    ap.add_argument("--precision", default="f16x3")
    a = ap.parse_args()
    cfgmod = importlib.import_module("1xgpt_amd.config")
    synth = importlib.import_module("1xgpt_amd.synthetic")
    _lib = importlib.import_module("1xgpt_amd._lib")
    STMaskGIT = importlib.import_module("1xgpt_amd.st_mask_git").STMaskGIT
    cfg_m = cfgmod.c138() if a.model == "c138" else cfgmod.c35()
    m = STMaskGIT(cfg_m, precision=a.precision).load_numpy_state_dict(synth.make_state_dict(cfg_m, seed=0, law="conditioned")).to("cuda")
    lib = _lib.load()
    cfg, w = m._weights()[:2]
    T, S, P = cfg_m.T, cfg_m.S, 8
    for B in a.batches:
        ids = torch.from_numpy(synth.make_clips(B, cfg_m, seed=3)).to("cuda").view(B, T, S).contiguous()
        ws = m._workspace(B)
        nbytes = lib.genie_prefix_cache_bytes(cfg, B)
        cache = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        for steps in a.steps:
            nz = torch.rand(T - P, max(steps - 1, 1), B, S, device="cuda")
            gen = torch.empty(B, T - P, S, dtype=torch.int64, device="cuda")

            def call(stream):
                _lib.check(lib.genie_generate_cached(cfg, w, ids.data_ptr(), B, P, T - P, steps, 0.0, _lib.UNMASK_RANDOM, nz.data_ptr(), 0, 0, 1,
                                                     gen.data_ptr(), 0, cache.data_ptr(), nbytes, ws.data_ptr(), ws.numel(), stream),
                           "genie_generate_cached")

            def timed(fn, reps=5):
                fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / reps

            t_direct = timed(lambda: call(torch.cuda.current_stream().cuda_stream))
            ref = gen.clone()
            res = {"batch": B, "steps": steps, "direct_ms_per_frame": round(t_direct / (T - P) * 1e3, 3), "direct_frames_per_s": round(B * (T - P) / t_direct, 1)}
            try:
                g = torch.cuda.CUDAGraph()
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    call(s.cuda_stream)            # warm-up on the capture stream (one-time attribute calls happen here)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    with torch.cuda.graph(g, stream=s):
                        call(torch.cuda.current_stream().cuda_stream)
                    res["capture_s"] = round(time.perf_counter() - t0, 3)
                torch.cuda.current_stream().wait_stream(s)
                gen.zero_()
                t_graph = timed(g.replay)
                res.update({"graph_ms_per_frame": round(t_graph / (T - P) * 1e3, 3), "graph_frames_per_s": round(B * (T - P) / t_graph, 1),
                            "ids_equal": bool(torch.equal(gen, ref))})
            except Exception as e:  # noqa: BLE001
                res["graph_error"] = f"{type(e).__name__}: {e}"[:200]
            print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
