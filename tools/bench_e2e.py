#!/usr/bin/env python3
"""BASELINE config 5: end-to-end MAGVIT2 encode -> GENIE sample -> MAGVIT2 decode on 256x256 RGB frames, everything
resident in HBM (the reference round-trips tokens and frames through numpy / PIL, eval_utils.py:39-41).
Synthetic frames and synthetic weights (no checkpoints offline)."""
import argparse
import importlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) / reps


def run_e2e(m, clips=8, steps=2, reps=2, chunk=64):
    """encode -> sample -> decode on the model `m` (an STMaskGIT on cuda); returns the result dict of this tool.
    chunk: frames per tokenizer call (the reference's decode_latents_wrapper default is 16, visualize.py:95; 64 fills the chip on
    the 16^2 .. 64^2 layers: encode 37.0 -> 30.7 ms, decode 21.8 -> 18.5 ms for 8 clips, profiles/r05t_e2e_tokenizer_chunks.txt)."""
    mv = importlib.import_module("1xgpt_amd.magvit2")
    G = importlib.import_module("1xgpt_amd.generate")
    cfg = m.config
    vq = mv.VQModel(mv.VQConfig())
    vq.load_state_dict({k: torch.from_numpy(v) for k, v in mv.make_vq_state_dict(vq, seed=1).items()})
    vq = vq.to(device="cuda").eval()
    B = clips
    g = torch.Generator(device="cuda").manual_seed(0)
    frames = torch.randint(0, 256, (B * 16, 3, 256, 256), dtype=torch.uint8, device="cuda", generator=g)

    he = vq.hip_encoder()

    def encode_hip():
        return torch.cat([he.encode_tokens(frames[i:i + chunk]) for i in range(0, B * 16, chunk)])

    tokens, t_enc = timed(encode_hip, reps)
    ids = tokens.view(B, 16, 16, 16)
    noise = torch.rand(8, max(steps - 1, 1), B, cfg.S, device="cuda")
    out, t_gen = timed(lambda: G.generate_frames_cached(m, ids, 8, steps, 0.0, False, noise=noise), reps)
    gen = out[:, 8:16].reshape(B * 8, 16, 16)

    hd = vq.hip_decoder()

    def decode_hip():
        return torch.cat([hd.decode_tokens(gen[i:i + chunk]) for i in range(0, B * 8, chunk)])

    rgb, t_dec = timed(decode_hip, reps)
    assert rgb.shape == (B * 8, 3, 256, 256) and rgb.dtype == torch.uint8 and rgb.is_cuda
    total = t_enc + t_gen + t_dec
    return {"clips": B, "frames_per_tokenizer_call": chunk, "encode_frames_per_sec": B * 16 / t_enc, "encode_tflops": 135.8e-3 * B * 16 / t_enc,
            "generate_frames_per_sec": B * 8 / t_gen,
            "decode_frames_per_sec": B * 8 / t_dec, "decode_tflops": 186.7e-3 * B * 8 / t_dec,
            "end_to_end_generated_frames_per_sec": B * 8 / total,
            "seconds": {"encode_hip": t_enc, "generate": t_gen, "decode_hip": t_dec}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=8)
    ap.add_argument("--precision", default="f16x3")
    ap.add_argument("--model", default="c138")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--chunk", type=int, default=64, help="frames per MAGVIT2 encode / decode call")
    a = ap.parse_args()
    cfgmod = importlib.import_module("1xgpt_amd.config")
    synth = importlib.import_module("1xgpt_amd.synthetic")
    STMaskGIT = importlib.import_module("1xgpt_amd.st_mask_git").STMaskGIT
    cfg = cfgmod.c138() if a.model == "c138" else cfgmod.c35()
    m = STMaskGIT(cfg, precision=a.precision).load_numpy_state_dict(synth.make_state_dict(cfg, seed=0)).to("cuda")
    res = run_e2e(m, a.clips, a.steps, chunk=a.chunk)
    res = {"workload": f"encode {a.clips}x16 frames -> sample 8 frames/clip ({a.steps} MaskGIT steps, KV cache, {a.precision}) -> "
                       f"decode {a.clips}x8 frames; {a.model}; MAGVIT2 encode and decode on hand-written implicit-GEMM convs", **res}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
