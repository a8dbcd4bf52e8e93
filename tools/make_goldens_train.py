#!/usr/bin/env python3
"""Generate tests/golden/train_*.npz by running the REFERENCE's training step (imported read-only from
/root/reference): its MaskGIT collator (data.py:109-169), ``STMaskGIT.forward`` + autograd backward
(genie/st_mask_git.py:231-279), ``clip_grad_norm_`` and ``torch.optim.AdamW`` with the parameter grouping of
train.py:426-441, and the scheduler factor of train.py:468-481.

Build-container only (see tools/make_goldens.py for the placeholder modules and why they contribute no arithmetic).
Fixtures hold data only: the collator's captured random draws with its outputs, the loss, every gradient tensor,
the global gradient norm, and the parameters after two optimizer steps.
"""
import math
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as mg  # noqa: E402  (installs the placeholders, puts /root/reference on sys.path)
import torch  # noqa: E402

import data as ref_data  # noqa: E402  reference data.py (collator)

synthetic = mg.synthetic
OUT = mg.OUT


class DrawRecorder:
    """Records, in call order, every random draw the reference collator makes."""

    def __init__(self):
        self.log = []  # (kind, array)
        self._orig = {}

    def __enter__(self):
        rec = self

        def wrap_t(name, fn):
            def f(*a, **k):
                r = fn(*a, **k)
                rec.log.append((name, r.detach().cpu().numpy().copy()))
                return r
            return f

        def wrap_p(name, fn):
            def f(*a, **k):
                r = fn(*a, **k)
                rec.log.append((name, np.asarray(r, dtype=np.float64)))
                return r
            return f

        for name in ("rand", "rand_like", "randint"):
            self._orig[name] = getattr(torch, name)
            setattr(torch, name, wrap_t("torch." + name, self._orig[name]))
        for name in ("random", "randint", "uniform"):
            self._orig["py." + name] = getattr(random, name)
            setattr(random, name, wrap_p("py." + name, self._orig["py." + name]))
        return self

    def __exit__(self, *exc):
        for name in ("rand", "rand_like", "randint"):
            setattr(torch, name, self._orig[name])
        for name in ("random", "randint", "uniform"):
            setattr(random, name, self._orig["py." + name])


def collate_with_capture(rcfg, ids, seed):
    torch.manual_seed(seed)
    random.seed(seed)
    fn = ref_data.get_maskgit_collator(rcfg)
    feats = [{"input_ids": torch.from_numpy(row.copy())} for row in ids]
    with DrawRecorder() as rec:
        batch = fn(feats)
    return batch, rec.log


def reference_param_groups(model, weight_decay):
    """train.py:426-437 (names containing "bias" or "layer_norm.weight" get no decay)."""
    no_decay = ["bias", "layer_norm.weight"]
    return [
        {"params": [p for n, p in model.named_parameters() if not any(nd in n for nd in no_decay)],
         "weight_decay": weight_decay},
        {"params": [p for n, p in model.named_parameters() if any(nd in n for nd in no_decay)],
         "weight_decay": 0.0},
    ]


def train_fixture(name, cfg_kwargs, wseed, B=2, lr=1e-3, weight_decay=0.1, max_grad_norm=1.0,
                  betas=(0.9, 0.999), eps=1e-8, full_grads=True):
    torch.set_grad_enabled(True)
    model, cfg = mg.build_ref_model(cfg_kwargs, wseed)
    model.train()
    rcfg = model.config
    out = {"cfg": repr(cfg_kwargs), "weight_seed": wseed, "lr": lr, "weight_decay": weight_decay, "max_grad_norm": max_grad_norm,
           "beta1": betas[0], "beta2": betas[1], "eps": eps}
    opt = torch.optim.AdamW(reference_param_groups(model, weight_decay), lr=lr, betas=betas, eps=eps)
    # the reference's "custom_cosine" factor (train.py:468-481) with warmup 1, max 4 steps
    def lr_factor(step, warmup=1, max_steps=4, end_ratio=0.1):
        if step < warmup:
            return (step + 1) / warmup
        rem = max_steps - warmup
        return ((1 + math.cos(math.pi * (step - warmup) / rem)) / 2) * (1 - end_ratio) + end_ratio
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_factor)
    found_both = set()
    for step in range(2):
        # pick a collator seed so that the two steps exercise both branches (MLM and non-MLM)
        for seed in range(500 + 50 * step, 550 + 50 * step):
            ids = synthetic.make_clips(B, cfg, seed=seed)
            batch, log = collate_with_capture(rcfg, ids, seed)
            branch = "nonmlm" if any(k == "py.randint" for k, _ in log) else "mlm"
            if branch not in found_both:
                found_both.add(branch)
                break
        out[f"s{step}_clip_seed"] = seed
        out[f"s{step}_branch"] = branch
        out[f"s{step}_ids"] = ids
        out[f"s{step}_input_ids"] = batch["input_ids"].numpy()
        out[f"s{step}_labels"] = batch["labels"].numpy()
        out[f"s{step}_draw_kinds"] = np.array([k for k, _ in log])
        for i, (_, a) in enumerate(log):
            out[f"s{step}_draw_{i}"] = a
        opt.zero_grad()
        res = model(batch["input_ids"], batch["labels"])
        res.loss.backward()
        out[f"s{step}_loss"] = np.float64(res.loss.item())
        out[f"s{step}_acc"] = np.float64(res.acc.item())
        if step == 0:
            for n, p in model.named_parameters():
                g = p.grad.detach().numpy()
                if full_grads:
                    out[f"s{step}_grad/{n}"] = g.copy()
                else:  # big models: norms and a strided sample per tensor
                    out[f"s{step}_gradnorm/{n}"] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
                    out[f"s{step}_gradsample/{n}"] = g.reshape(-1)[:: max(1, g.size // 64)][:64].copy()
        tn = torch.nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm)
        out[f"s{step}_grad_norm"] = np.float64(tn.item())
        out[f"s{step}_lr"] = np.float64(sched.get_last_lr()[0])
        opt.step()
        sched.step()
    for n, p in model.named_parameters():
        a = p.detach().numpy()
        if full_grads:
            out[f"final_param/{n}"] = a.copy()
        else:
            out[f"final_paramsample/{n}"] = a.reshape(-1)[:: max(1, a.size // 64)][:64].copy()
    torch.set_grad_enabled(False)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: loss {out['s0_loss']:.6f} -> {out['s1_loss']:.6f}, |g| {out['s0_grad_norm']:.4f}, "
          f"branches {out['s0_branch']},{out['s1_branch']}, {os.path.getsize(path) / 1e6:.2f} MB")


def main():
    os.makedirs(OUT, exist_ok=True)
    base = dict(num_layers=2, num_heads=2, d_model=64, T=4, S=16, num_factored_vocabs=2, num_prompt_frames=2)
    train_fixture("train_tiny_ln", dict(base, qk_norm=False, use_mup=False), 31)
    train_fixture("train_tiny_qknorm", dict(base, qk_norm=True, use_mup=False), 32)
    # real frame geometry (T=16, S=256) at small width: kernels run their production tile shapes
    real = dict(num_layers=2, num_heads=2, d_model=128, T=16, S=256, num_factored_vocabs=2)
    train_fixture("train_shape_dh64", dict(real, qk_norm=False, use_mup=False), 33, B=1, full_grads=False)


if __name__ == "__main__":
    main()
