"""Rank process of tests/test_hip_multirank.py (not collected by pytest).  Runs the data-parallel HIP paths -- evaluate_clips
(distributed=True) and two GenieTrainer steps -- on this rank's shard and writes what it saw to <out_dir>/w<world>_r<rank>.json.
Launched plain (world 1) or under torch.distributed.run with GENIE_FORCE_DEVICE=0 GENIE_DIST_BACKEND=gloo (all ranks on the
one GPU of the test box; the collective is gloo's, the compute is the HIP library's)."""
import importlib
import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def pkg(name):
    return importlib.import_module("1xgpt_amd." + name)


def main():
    out_dir = sys.argv[1]
    D = pkg("distributed")
    rank, world, local_rank = D.init_distributed()
    dev = torch.device("cuda", D.local_device_index(local_rank))
    torch.cuda.set_device(dev)
    cfgmod, synth = pkg("config"), pkg("synthetic")
    STMaskGIT = pkg("st_mask_git").STMaskGIT
    res = {"rank": rank, "world": world, "lib": os.path.realpath(pkg("_lib").LIB_PATH)}

    # ---- data-parallel evaluate: 6 clips, sharded; real token geometry
    cfg = cfgmod.GenieConfig(num_layers=2, num_heads=2, d_model=128, T=16, S=256, num_factored_vocabs=2, qk_norm=False,
                             use_mup=False)
    sd = synth.make_state_dict(cfg, seed=22, law="conditioned")
    clips = torch.from_numpy(synth.make_clips(6, cfg, seed=5))
    lo, hi = D.shard_range(6, rank, world)
    model = STMaskGIT(cfg, precision="f16x3").load_numpy_state_dict(sd).to(dev)
    ev = pkg("evaluate").GenieEvaluator(SimpleNamespace(maskgit_steps=2, temperature=0, latent_h=16, latent_w=16), None, dev,
                                        model=model)
    r = pkg("evaluate").evaluate_clips(ev, clips[lo:hi].to(dev), batch_size=1, noise_seed=100, distributed=world > 1,
                                       reuse=True, clip_offset=lo)
    res["evaluate"] = {k: r[k] for k in ("loss", "acc", "frames", "clips")}

    # ---- data-parallel training: 4 clips, frames >= 2 masked in every clip (equal masked-token counts per rank, so the
    # average of the ranks' gradients is the gradient of the whole batch's mean loss: DDP == one big batch)
    tcfg = cfgmod.GenieConfig(num_layers=2, num_heads=2, d_model=64, T=4, S=16, num_factored_vocabs=2, qk_norm=False,
                              use_mup=False, num_prompt_frames=2)
    tsd = synth.make_state_dict(tcfg, seed=9, law="conditioned")
    ids = synth.make_clips(4, tcfg, seed=10)
    x = ids.reshape(4, tcfg.T, tcfg.S).copy()
    x[:, 2:] = tcfg.image_vocab_size
    lo, hi = D.shard_range(4, rank, world)
    tm = STMaskGIT(tcfg, precision="exact").load_numpy_state_dict(tsd).to(dev)
    tr = pkg("train").GenieTrainer(tm, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0, bucket_mb=0.05)
    batch = {"input_ids": torch.from_numpy(x[lo:hi].reshape(hi - lo, -1)).to(dev),
             "labels": torch.from_numpy(ids[lo:hi]).to(dev)}
    losses, norms = [], []
    for _ in range(2):
        o = tr.train_step(batch)
        loss = o["loss"].clone()
        if world > 1:
            torch.distributed.all_reduce(loss)
            loss /= world
        losses.append(float(loss))
        norms.append(float(o["grad_norm"]))
    p = tr.params.double()
    res["train"] = {"losses": losses, "grad_norms": norms, "n_params": int(p.numel()), "buckets": len(tr.reducer.bounds),
                    "param_sum": float(p.sum()), "param_abs_sum": float(p.abs().sum()),
                    "param_head": tr.params[:2048].cpu().numpy().astype(np.float64).tolist()}
    torch.cuda.synchronize()
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"w{world}_r{rank}.json"), "w") as f:
        json.dump(res, f)
    D.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
