"""Pin the CPU oracle (oracle/genie_oracle.py) against the reference's own outputs (tests/golden/*.npz,
made by tools/make_goldens.py from the imported reference).  CPU only."""
import math

import importlib

import numpy as np
import pytest

from oracle import genie_oracle as O

TINY = ["tiny_ln", "tiny_qknorm", "tiny_mup", "tiny_qknorm_mup"]
LOGIT_TOL = 2e-5  # fp32 reference vs fp32/fp64 oracle, 2 layers, |logit| ~ 2..7


@pytest.mark.parametrize("name", TINY)
def test_compute_logits(golden, name):
    z, cfg, sd = golden(name)
    H = W = math.isqrt(cfg.S)
    ids = z["ids"].reshape(-1, cfg.T, H, W)
    scale = max(1.0, float(np.abs(z["logits"]).max()) / 8)
    for nm in (O.F32, O.F64):
        lg = O.compute_logits(ids, sd, cfg, nm)
        assert lg.shape == z["logits"].shape
        assert np.abs(lg - z["logits"]).max() < LOGIT_TOL * scale


@pytest.mark.parametrize("name", TINY)
def test_forward_loss_acc(golden, name):
    z, cfg, sd = golden(name)
    loss, acc, logits = O.forward_loss_acc(z["fwd_input"], z["ids"], sd, cfg)
    assert abs(loss - float(z["fwd_loss"])) < 1e-4
    assert abs(acc - float(z["fwd_acc"])) < 1e-7
    assert abs(logits.astype(np.float64).sum() - float(z["fwd_logits_sum"])) < 0.05
    # no masked token -> 0/0 -> nan, the reference has no guard (st_mask_git.py:248-250)
    loss2, _, _ = O.forward_loss_acc(z["ids"], z["ids"], sd, cfg)
    assert math.isnan(loss2) and bool(z["fwd_nomask_loss_isnan"])


@pytest.mark.parametrize("name", TINY)
@pytest.mark.parametrize("steps", [1, 2, 3, 8])
@pytest.mark.parametrize("mode", ["random", "greedy"])
def test_maskgit_generate(golden, name, steps, mode):
    z, cfg, sd = golden(name)
    H = W = math.isqrt(cfg.S)
    prompt = z["ids"].reshape(-1, cfg.T, H, W).copy()
    prompt[:, 2:] = cfg.image_vocab_size
    k = f"mg_s{steps}_{mode}"
    s, fl = O.maskgit_generate(prompt, 2, sd, cfg, steps, 0.0, mode, noise=z[k + "_noise"])
    assert np.array_equal(s, z[k + "_samples"])
    assert np.array_equal(prompt, z[k + "_prompt_after"])  # in-place write-back (st_mask_git.py:223)
    if steps == 2 and mode == "random":
        assert fl.shape == z["mg_step0_factored_logits"].shape
        assert np.abs(fl - z["mg_step0_factored_logits"]).max() < 1e-4


def test_maskgit_asserts(golden):
    z, cfg, sd = golden("tiny_ln")
    H = W = math.isqrt(cfg.S)
    prompt = z["ids"].reshape(-1, cfg.T, H, W).copy()
    with pytest.raises(AssertionError):
        O.maskgit_generate(prompt, 0, sd, cfg)
    with pytest.raises(AssertionError):  # frames >= out_t not masked
        O.maskgit_generate(prompt, 2, sd, cfg)
    prompt[:, 2:] = cfg.image_vocab_size
    with pytest.raises(NotImplementedError):
        O.maskgit_generate(prompt, 2, sd, cfg, 2, unmask_mode="bogus", noise=np.zeros((1, 2, 16), np.float32))


@pytest.mark.parametrize("name", TINY)
def test_generate(golden, name):
    z, cfg, sd = golden(name)
    out = O.generate(z["ids"][:, :2 * cfg.S], 2 * cfg.S, sd, cfg, maskgit_steps=2, noise=z["gen_noise"],
                     return_logits=(name == "tiny_ln"))
    if name == "tiny_ln":
        out, gl = out
        assert np.abs(gl - z["gen_logits"]).max() < 1e-4
    assert np.array_equal(out, z["gen_out"])


@pytest.mark.parametrize("name", TINY)
def test_evaluate_harness(golden, name):
    z, cfg, sd = golden(name)
    loss, acc, samples, fl = O.evaluate_metrics(z["ids"], sd, cfg, 2, noise=z["ev_noise"])
    assert np.array_equal(samples, z["ev_samples"])
    assert np.abs(fl - z["ev_logits"]).max() < 1e-4
    assert abs(loss - float(z["ev_loss"])) < 1e-4
    assert abs(acc - float(z["ev_acc"])) < 1e-7


@pytest.mark.parametrize("name", ["shape_dh32", "shape_dh64"])
def test_real_geometry(golden, name):
    """T=16, S=256 token geometry: probe logits, forward CE, MaskGIT ids, evaluate ids and CE."""
    z, cfg, sd = golden(name)
    H = W = 16
    ids = z["ids"]
    x = ids.reshape(-1, 16, H, W).copy()
    x[:, 8:] = cfg.image_vocab_size
    loss, acc, logits = O.forward_loss_acc(x.reshape(1, -1), ids, sd, cfg)
    assert abs(loss - float(z["fwd_loss"])) < 1e-4
    assert abs(acc - float(z["fwd_acc"])) < 1e-7
    probe = np.stack([logits[:, :, t, s // W, s % W] for t, s in zip(z["probe_t"], z["probe_s"])], 1)
    assert np.abs(probe - z["probe_logits"]).max() < 5e-5
    p = x.copy()
    s, _ = O.maskgit_generate(p, 8, sd, cfg, 2, noise=z["mg_s2_noise"])
    assert np.array_equal(s, z["mg_s2_samples"])
    loss, acc, samples, _ = O.evaluate_metrics(ids, sd, cfg, 2, noise=z["ev_noise"])
    assert np.array_equal(samples, z["ev_samples"])
    assert abs(loss - float(z["ev_loss"])) < 1e-4
    assert abs(acc - float(z["ev_acc"])) < 1e-7


def test_qknorm_real_geometry(golden):
    z, cfg, sd = golden("shape_dh64_qknorm")
    x = z["ids"].reshape(-1, 16, 16, 16).copy()
    x[:, 8:] = cfg.image_vocab_size
    for steps in (2, 8):
        p = x.copy()
        s, _ = O.maskgit_generate(p, 8, sd, cfg, steps, noise=z[f"mg_s{steps}_noise"])
        assert np.array_equal(s, z[f"mg_s{steps}_samples"])


def test_mask_schedule():
    # n re-masked per step: SURVEY Appendix A / st_mask_git.py:199
    assert O.mask_counts(2, 256) == [182]
    assert O.mask_counts(4, 256) == [237, 182, 98]
    assert O.mask_counts(8, 256) == [252, 237, 213, 182, 143, 98, 50]
    assert O.mask_counts(1, 256) == []


def test_factorization_roundtrip():
    ids = np.array([0, 1, 511, 512, 262143, 131072 + 5])
    f = O.factorize_token_ids(ids)
    assert np.array_equal(f[:, 0], ids % 512) and np.array_equal(f[:, 1], ids // 512)
    assert np.array_equal(O.unfactorize_token_ids(f), ids)


def test_bits_from_tokens():
    ids = np.array([[[0, 1], [2, 262143]]])
    b = O.bits_from_tokens(ids)
    assert b.shape == (1, 18, 2, 2)
    assert np.all(b[0, :, 0, 0] == -1) and b[0, 0, 0, 1] == 1 and b[0, 1, 1, 0] == 1 and np.all(b[0, :, 1, 1] == 1)


def test_round_bf16():
    a = np.array([1.0, 1.00390625, 1.01171875, -3.14159, 65504.0, 1e-40], np.float32)
    r = O.round_bf16(a)
    import torch
    ref = torch.from_numpy(a).to(torch.bfloat16).float().numpy()
    assert np.array_equal(r, ref)


@pytest.mark.parametrize("name", ["anchor_c35", "anchor_c138"])
def test_full_size_anchor(golden, name):
    """32-layer shapes (C35 = shipped config, C138-shape = inferred GENIE_138M): forward CE + 2-step MaskGIT ids."""
    z, cfg, sd = golden(name)
    W = 16
    ids = z["ids"]
    x = ids.reshape(-1, 16, 16, 16).copy()
    x[:, 8:] = cfg.image_vocab_size
    loss, acc, logits = O.forward_loss_acc(x.reshape(1, -1), ids, sd, cfg)
    assert abs(loss - float(z["fwd_loss"])) < 1e-4
    assert abs(acc - float(z["fwd_acc"])) < 1e-7
    probe = np.stack([logits[:, :, t, s // W, s % W] for t, s in zip(z["probe_t"], z["probe_s"])], 1)
    assert np.abs(probe - z["probe_logits"]).max() < 2e-4
    s, _ = O.maskgit_generate(x.copy(), 8, sd, cfg, 2, noise=z["mg_s2_noise"])
    assert np.array_equal(s, z["mg_s2_samples"])


def test_torch_port_matches_oracle(golden):
    """oracle/genie_torch_port.py (the torch-CPU restatement timed by bench.py's cpu_baseline) against the reference's
    golden logits and the NumPy oracle, on every tiny config (LN / qk-norm / muP)."""
    TP = importlib.import_module("oracle.genie_torch_port")
    for name in ("tiny_ln", "tiny_qknorm", "tiny_mup", "tiny_qknorm_mup"):
        z, cfg, sd = golden(name)
        H = W = int(round(cfg.S ** 0.5))
        x = z["ids"].reshape(-1, cfg.T, H, W)
        lg = TP.compute_logits(x, TP.to_torch(sd), cfg)
        assert np.abs(lg - z["logits"]).max() < 2e-5 * max(1.0, float(np.abs(z["logits"]).max()) / 8)
        assert np.abs(lg - O.compute_logits(x, sd, cfg)).max() < 2e-5 * max(1.0, float(np.abs(z["logits"]).max()) / 8)


@pytest.mark.parametrize("name", ["ev_c138", "ev_c138_h16", "ev_c138_qknorm", "ev_c138_default", "ev_c138_robust", "ev_c35_robust"])
def test_bench_workload_golden(golden, name):
    """The benchmarked workload at full size (bench.py's weights and clip 0 through the reference's teacher-forced evaluate,
    tools/make_goldens.py c138_ev): the oracle's MaskGIT loop over the torch-CPU port of the forward reproduces the reference's
    ids and per-timestep CE on the first and the last timestep (all 15 take ~2 minutes of CPU; the GPU suite checks all)."""
    TP = importlib.import_module("oracle.genie_torch_port")
    z, cfg, sd = golden(name)
    sdt = TP.to_torch(sd)
    x = z["ids"].reshape(1, 16, 16, 16)
    for t in (1, 15):
        p = x.copy()
        p[:, t:] = cfg.image_vocab_size
        s, fl = O.maskgit_generate(p, t, sd, cfg, 2, 0.0, "random", noise=z["ev_noise"][t - 1],
                                   logits_fn=lambda q: TP.compute_logits(q, sdt, cfg))
        same = s[0] == z["ev_samples"][0, t - 1]
        if z["ev_frame_gap"][t - 1] > 6e-5:
            assert same.all(), (t, int((~same).sum()))
        else:
            assert same.mean() > 0.97, (t, same.mean())
        lab = np.concatenate([z["ids"].reshape(1, 16, 256)[:, :1], z["ids"].reshape(1, 16, 256)[:, t:t + 1]], 1).reshape(1, -1)
        ce = O.compute_loss(lab, fl[:, :, :, None], cfg)
        assert abs(ce - float(z["ev_loss_per_t"][t - 1])) < 1e-4, (t, ce, float(z["ev_loss_per_t"][t - 1]))
