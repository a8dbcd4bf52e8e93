#!/usr/bin/env python3
"""CLI counterpart of the reference's `python genie/generate.py` (generate.py:21-116): prompt frames -> generated frames,
written as [prompt | generated | ground truth] video.bin + metadata.json (readable by RawTokenDataset / visualize).

  python tools/generate.py --checkpoint_dir DIR --val_data_dir data/val_v1.1 --output_dir data/genie_generated
  python tools/generate.py --synthetic --model c35 --output_dir /tmp/gen"""
import argparse
import importlib
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--val_data_dir", type=str, default="data/val_v1.1")
    ap.add_argument("--checkpoint_dir", type=str)
    ap.add_argument("--output_dir", type=str, default="data/genie_generated")
    ap.add_argument("--num_prompt_frames", type=int, default=8)
    ap.add_argument("--window_size", type=int, default=16)
    ap.add_argument("--example_ind", type=int, default=0)
    ap.add_argument("--teacher_force_time", action="store_true")
    ap.add_argument("--maskgit_steps", type=int, default=2)
    ap.add_argument("--temperature", type=float, default=0)
    ap.add_argument("--precision", choices=["exact", "f16x3", "bf16"], default="f16x3")
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--model", choices=["c138", "c35"], default="c35")
    ap.add_argument("--schedule", choices=["kv_cache", "full_forward"], default="kv_cache",
                    help="kv_cache: one-frame passes against a temporal KV cache (same frames up to f32 accumulation order); "
                         "full_forward: the reference's schedule, a full 16-frame forward per MaskGIT step (generate.py:81-95)")
    args = ap.parse_args()
    G = importlib.import_module("1xgpt_amd.generate")
    STMaskGIT = importlib.import_module("1xgpt_amd.st_mask_git").STMaskGIT
    if args.synthetic:
        cfgmod = importlib.import_module("1xgpt_amd.config")
        synth = importlib.import_module("1xgpt_amd.synthetic")
        cfg = cfgmod.c138() if args.model == "c138" else cfgmod.c35()
        model = STMaskGIT(cfg, precision=args.precision).load_numpy_state_dict(synth.make_state_dict(cfg, seed=0))
        example = torch.from_numpy(synth.make_clips(1, cfg, seed=1234 + args.example_ind))
        meta = {"s": model.h, "vocab_size": cfg.image_vocab_size, "hz": 2, "token_dtype": "uint32"}
    else:
        model = STMaskGIT.from_pretrained(args.checkpoint_dir, precision=args.precision)
        ds = importlib.import_module("1xgpt_amd.data").RawTokenDataset(args.val_data_dir, window_size=args.window_size,
                                                                      stride=G.STRIDE)
        example = ds[args.example_ind]["input_ids"][None]
        meta = ds.metadata
    model = model.to("cuda")
    ex = example.to("cuda").view(1, args.window_size, model.h, model.w)
    fn = G.generate_frames_cached if args.schedule == "kv_cache" else G.generate_frames
    out = fn(model, ex, args.num_prompt_frames, args.maskgit_steps, args.temperature, args.teacher_force_time)
    print(G.write_outputs(out, args.output_dir, meta, vars(args)))


if __name__ == "__main__":
    main()
