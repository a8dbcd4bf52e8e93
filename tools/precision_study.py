#!/usr/bin/env python3
"""Accuracy of cheaper Linear formulations, per GEMM CLASS and per LAYER RANGE.  Needs the study build of
the library (GENIE_STUDY=1 python 1xgpt_amd/build.py) -- the shipping library has no reduced-precision knob:

    GENIE_HIP_LIBRARY=1xgpt_amd/libgenie_hip_study.so python tools/precision_study.py > profiles/r03_precision_study_classes.txt

f16x3 computes every Linear as hi.hi + hi.lo + lo.hi (3 MFMAs per algorithmic MFMA, f32-class).  The 2-term form drops the
weight's lo plane (weights rounded to f16, activations exact): 2 MFMAs.  For each setting -- a set of classes (qkv_s, qkv_t,
proj_s, proj_t, fc1, fc2, readout) running on 2 terms, optionally only in layers [lo, hi) -- and each fixture: CE delta against
the f32 reference golden, max |dlogit| on the probe logits, temperature-0 MaskGIT ids that differ from the reference's.
Adoption bar: max|dlogit| < 5e-5 AND every golden id exact with margin on every fixture."""
import importlib
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_golden  # noqa: E402

_lib = importlib.import_module("1xgpt_amd._lib")
STMaskGIT = importlib.import_module("1xgpt_amd.st_mask_git").STMaskGIT
CLASSES = ["qkv_s", "qkv_t", "proj_s", "proj_t", "fc1", "fc2", "readout"]
FLOP_SHARE = {"qkv_s": 0.187, "qkv_t": 0.187, "proj_s": 0.062, "proj_t": 0.062, "fc1": 0.249, "fc2": 0.249, "readout": 0.004}


def measure(models):
    rows = []
    for name, (z, cfg, m) in models.items():
        ids = torch.from_numpy(z["ids"]).cuda()
        x = ids.view(-1, 16, 16, 16).clone()
        x[:, 8:] = cfg.image_vocab_size
        out = m(x.view(1, -1), ids)
        lg = out.logits.cpu().numpy()
        probe = np.stack([lg[:, :, t, s // 16, s % 16] for t, s in zip(z["probe_t"], z["probe_s"])], 1)
        s, _ = m.maskgit_generate(x.clone(), 8, maskgit_steps=2, noise=torch.from_numpy(z["mg_s2_noise"]).cuda())
        rows.append((name, out.loss.item() - float(z["fwd_loss"]), float(np.abs(probe - z["probe_logits"]).max()),
                     int((s.cpu().numpy() != z["mg_s2_samples"]).sum())))
    return rows


def main():
    # one clip = 16 row tiles per GEMM: send them to the 256x256 kernel anyway (it is the only one with a 2-term form)
    os.environ.setdefault("GENIE_GEMM16_PP_MIN_TILES", "0")
    assert _lib.load().genie_study_build(), "needs the -DGENIE_STUDY library: GENIE_HIP_LIBRARY=1xgpt_amd/libgenie_hip_study.so"
    models = {}
    for name in ("shape_dh32", "shape_dh64", "anchor_c35", "anchor_c138"):
        z, cfg, sd = load_golden(name)
        models[name] = (z, cfg, STMaskGIT(cfg, precision="f16x3").load_numpy_state_dict(sd).to("cuda"))
    settings = [("3 terms everywhere (f16x3)", 0, None)]
    settings += [(f"2 terms: {c}", 1 << i, None) for i, c in enumerate(CLASSES)]
    settings += [("2 terms: fc1 + fc2 (MLP)", (1 << 4) | (1 << 5), None), ("2 terms: proj_s + proj_t", (1 << 2) | (1 << 3), None),
                 ("2 terms: all classes", 127, None)]
    half = [("2 terms: fc1 + fc2, first half of the layers", (1 << 4) | (1 << 5), "first"),
            ("2 terms: fc1 + fc2, second half of the layers", (1 << 4) | (1 << 5), "second"),
            ("2 terms: fc1 + fc2, last quarter of the layers", (1 << 4) | (1 << 5), "last_quarter"),
            ("2 terms: all classes, last layer only", 127, "last")]
    print("# GEMM-class / layer-range study of the 2-term split (weights rounded to f16): tools/precision_study.py, study build")
    print(f"# {'setting':52s} {'fixture':12s} {'CE delta':>11s} {'max|dlogit|':>12s} {'ids != ref /256':>16s}  2-term share of GEMM FLOPs")
    for label, mask, rng in settings + half:
        os.environ["GENIE_F16_TERMS2_CLASSES"] = str(mask)
        share = sum(FLOP_SHARE[c] for i, c in enumerate(CLASSES) if mask >> i & 1)
        for name, (z, cfg, m) in models.items():
            L = cfg.num_layers
            lo, hi = {None: (0, 1 << 30), "first": (0, L // 2), "second": (L // 2, L), "last_quarter": (L - max(L // 4, 1), L),
                      "last": (L - 1, L)}[rng]
            os.environ["GENIE_F16_TERMS2_LAYER_LO"], os.environ["GENIE_F16_TERMS2_LAYER_HI"] = str(lo), str(hi)
            frac = share * (min(hi, L) - lo) / L
            (n, dce, dl, mism), = measure({name: (z, cfg, m)})
            ok = dl < 5e-5 and mism == 0
            print(f"{label:54s} {n:12s} {dce:+11.3e} {dl:12.3e} {mism:10d}/256   {frac:5.3f}   {'ok' if ok else 'FAILS the bar'}",
                  flush=True)
    os.environ["GENIE_F16_TERMS2_CLASSES"] = "0"


if __name__ == "__main__":
    main()
