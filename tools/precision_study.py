#!/usr/bin/env python3
"""Accuracy of the candidate fast Linear formulations on the full-size anchors.

    GENIE_F16_TERMS=3  f16x3: hi.hi + hi.lo + lo.hi (f32-class, the parity-grade mode)
    GENIE_F16_TERMS=2  A exact (hi + lo), W rounded to f16 (2 MFMAs per algorithmic MFMA)
    GENIE_F16_TERMS=1  plain f16 (hi planes only, 1 MFMA)
    precision=bf16     bf16 operands

For each: CE delta vs the f32 reference golden, max |dlogit| on the probe logits, temperature-0 MaskGIT ids that
differ from the reference's (of 256).  The GEMM variant is a process-wide static, so run once per setting:
    for t in 3 2 1; do GENIE_F16_TERMS=$t python tools/precision_study.py f16x3; done; python tools/precision_study.py bf16
"""
import importlib
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_golden  # noqa: E402

STMaskGIT = importlib.import_module("1xgpt_amd.st_mask_git").STMaskGIT


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
    tag = prec + (f" TERMS={os.environ.get('GENIE_F16_TERMS', '3')}" if prec == "f16x3" else "")
    for name in ("anchor_c35", "anchor_c138"):
        z, cfg, sd = load_golden(name)
        m = STMaskGIT(cfg, precision=prec).load_numpy_state_dict(sd).to("cuda")
        ids = torch.from_numpy(z["ids"]).cuda()
        x = ids.view(-1, 16, 16, 16).clone()
        x[:, 8:] = cfg.image_vocab_size
        out = m(x.view(1, -1), ids)
        lg = out.logits.cpu().numpy()
        probe = np.stack([lg[:, :, t, s // 16, s % 16] for t, s in zip(z["probe_t"], z["probe_s"])], 1)
        s, _ = m.maskgit_generate(x.clone(), 8, maskgit_steps=2, noise=torch.from_numpy(z["mg_s2_noise"]).cuda())
        mism = int((s.cpu().numpy() != z["mg_s2_samples"]).sum())
        print(f"{tag:16s} {name:12s} CE delta {out.loss.item() - float(z['fwd_loss']):+.3e}  max|dlogit| "
              f"{np.abs(probe - z['probe_logits']).max():.3e}  id mismatches {mism}/256  (fixture min top-2 gap "
              f"{float(z['min_gap']):.1e})", flush=True)
        del m


if __name__ == "__main__":
    main()
