#!/usr/bin/env python3
"""CLI counterpart of the reference's `python genie/evaluate.py` (evaluate.py:35-66, 145-204) on the MI355X path.

  python tools/evaluate.py --checkpoint_dir DIR --val_data_dir data/val_v1.1 [--maskgit_steps 2] [--batch_size 16]
  python tools/evaluate.py --synthetic 32 --model c138            # no checkpoint / dataset offline: synthetic weights + clips
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/evaluate.py ...   # data-parallel

Prints the running means like the reference ({gen_time, loss, acc}); LPIPS is out of scope (needs the `lpips` AlexNet)."""
import argparse
import importlib
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description="Evaluate GENIE-style models (MI355X path).")
    ap.add_argument("--val_data_dir", type=str, default="data/val_v1.1")
    ap.add_argument("--checkpoint_dir", type=str)
    ap.add_argument("--batch_size", type=int, default=16)
    ap.add_argument("--maskgit_steps", type=int, default=2)
    ap.add_argument("--temperature", type=float, default=0)
    ap.add_argument("--max_examples", type=int)
    ap.add_argument("--precision", choices=["exact", "f16x3", "bf16"], default="f16x3")
    ap.add_argument("--synthetic", type=int, default=0, help="evaluate N synthetic clips with synthetic weights")
    ap.add_argument("--model", choices=["c138", "c35"], default="c35", help="shape for --synthetic")
    ap.add_argument("--no_reuse", action="store_true", help="reference schedule (15 x steps full forwards)")
    args = ap.parse_args()

    ev_mod = importlib.import_module("1xgpt_amd.evaluate")
    dist_mod = importlib.import_module("1xgpt_amd.distributed")
    cfgmod = importlib.import_module("1xgpt_amd.config")
    synth = importlib.import_module("1xgpt_amd.synthetic")
    STMaskGIT = importlib.import_module("1xgpt_amd.st_mask_git").STMaskGIT
    rank, world, local_rank = dist_mod.init_distributed()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if args.synthetic:
        cfg = cfgmod.c138() if args.model == "c138" else cfgmod.c35()
        model = STMaskGIT(cfg, precision=args.precision).load_numpy_state_dict(synth.make_state_dict(cfg, seed=0))
        clips = torch.from_numpy(synth.make_clips(args.synthetic, cfg, seed=1234))
        side = model.h
    else:
        model = STMaskGIT.from_pretrained(args.checkpoint_dir, precision=args.precision)
        ds = importlib.import_module("1xgpt_amd.data").RawTokenDataset(
            args.val_data_dir, window_size=ev_mod.WINDOW_SIZE, stride=ev_mod.STRIDE, filter_overlaps=True)
        n = len(ds) if args.max_examples is None else min(len(ds), args.max_examples)
        clips = ds.batch(range(n))
        side = ds.metadata["s"]
    args.latent_h = args.latent_w = side
    lo, hi = dist_mod.shard_range(clips.shape[0], rank, world)
    ev = ev_mod.GenieEvaluator(args, None, dev, model=model)
    res = ev_mod.evaluate_clips(ev, clips[lo:hi], batch_size=args.batch_size, distributed=world > 1,
                                reuse=not args.no_reuse)
    if rank == 0:
        res["gen_time_s_per_frame"] = res["seconds"] / max(res["frames"], 1)
        print(json.dumps(res))


if __name__ == "__main__":
    main()
