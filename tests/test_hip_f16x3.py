"""GENIE_PREC_F16X3 (every Linear on the f16 matrix cores with split operands, 3 MFMAs per K-step) held to the
SAME bar as the exact f32 path: CE within 1e-4 of the reference goldens, temperature-0 MaskGIT ids bit-exact,
logits within f32-accumulation noise.  Needs a GPU: -m gpu."""
import math
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import pkg
from oracle import genie_oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
TINY = ["tiny_ln", "tiny_qknorm", "tiny_mup", "tiny_qknorm_mup"]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


@pytest.fixture(scope="module")
def models(golden):
    cache = {}

    def get(name):
        if name not in cache:
            z, cfg, sd = golden(name)
            cache[name] = pkg("st_mask_git").STMaskGIT(cfg, precision="f16x3").load_numpy_state_dict(sd).to("cuda")
        return cache[name]

    return get


def test_split_gemm_is_f32_class():
    """The split-f16 GEMM alone, through the muP-free readout of a 1-layer model is covered below; here the
    operand split itself: hi + lo/2048 reproduces f32 inputs to ~2^-22 (oracle emulation == definition)."""
    g = np.random.default_rng(0)
    a = (g.standard_normal(100000) * np.exp(g.uniform(-8, 4, 100000))).astype(np.float32)
    r = O.round_f16_split(a)
    err = np.abs(r.astype(np.float64) - a)
    # 22 bits in the f16 normal range; below it hi is flushed and lo alone carries 11 bits of a tiny value
    assert np.all(err <= 2.0 ** -21 * np.abs(a) + 2.0 ** -11 * 6.2e-5)


@pytest.mark.parametrize("name", TINY)
def test_compute_logits_golden(golden, models, name):
    z, cfg, sd = golden(name)
    m = models(name)
    H = W = math.isqrt(cfg.S)
    lg = m.compute_logits(dev(z["ids"]).view(-1, cfg.T, H, W)).cpu().numpy()
    scale = max(1.0, float(np.abs(z["logits"]).max()) / 8)
    assert np.abs(lg - z["logits"]).max() < 5e-5 * scale


@pytest.mark.parametrize("name", TINY)
def test_forward_and_maskgit_golden(golden, models, name):
    z, cfg, sd = golden(name)
    m = models(name)
    out = m(dev(z["fwd_input"]), dev(z["ids"]))
    assert abs(out.loss.item() - float(z["fwd_loss"])) < 1e-4
    assert abs(out.acc.item() - float(z["fwd_acc"])) < 1e-7
    H = W = math.isqrt(cfg.S)
    for steps in (1, 2, 3, 8):
        for mode in ("random", "greedy"):
            prompt = dev(z["ids"]).view(-1, cfg.T, H, W).clone()
            prompt[:, 2:] = cfg.image_vocab_size
            k = f"mg_s{steps}_{mode}"
            noise = dev(z[k + "_noise"]) if z[k + "_noise"].size else None
            s, _ = m.maskgit_generate(prompt, 2, maskgit_steps=steps, unmask_mode=mode, noise=noise)
            assert np.array_equal(s.cpu().numpy(), z[k + "_samples"]), (steps, mode)


@pytest.mark.parametrize("name", TINY)
def test_evaluator_golden(golden, models, name):
    z, cfg, sd = golden(name)
    H = W = math.isqrt(cfg.S)
    args = SimpleNamespace(maskgit_steps=2, temperature=0, latent_h=H, latent_w=W)
    ev = pkg("evaluate").GenieEvaluator(args, None, "cuda", model=models(name))
    samples, fl = ev.predict_zframe_logits(dev(z["ids"]), noise=dev(z["ev_noise"]))
    assert np.array_equal(samples.cpu().numpy(), z["ev_samples"])
    assert abs(pkg("eval_utils").compute_loss(dev(z["ids"]), fl) - float(z["ev_loss"])) < 1e-4


@pytest.mark.parametrize("name", ["shape_dh32", "shape_dh64"])
def test_real_geometry_golden(golden, models, name):
    z, cfg, sd = golden(name)
    m = models(name)
    ids = dev(z["ids"])
    x = ids.view(-1, 16, 16, 16).clone()
    x[:, 8:] = cfg.image_vocab_size
    out = m(x.view(1, -1), ids)
    assert abs(out.loss.item() - float(z["fwd_loss"])) < 1e-4
    lg = out.logits.cpu().numpy()
    probe = np.stack([lg[:, :, t, s // 16, s % 16] for t, s in zip(z["probe_t"], z["probe_s"])], 1)
    assert np.abs(probe - z["probe_logits"]).max() < 5e-5
    args = SimpleNamespace(maskgit_steps=2, temperature=0, latent_h=16, latent_w=16)
    ev = pkg("evaluate").GenieEvaluator(args, None, "cuda", model=m)
    sums = ev.evaluate_metric_sums(ids, noise=dev(z["ev_noise"])).tolist()
    assert abs(sums[0] / sums[1] - float(z["ev_loss"])) < 1e-4
    assert abs(sums[2] / sums[3] - float(z["ev_acc"])) < 1e-7
    samples, _ = ev.predict_zframe_logits(ids, noise=dev(z["ev_noise"]), return_logits=False)
    assert np.array_equal(samples.cpu().numpy(), z["ev_samples"])


def test_qknorm_real_geometry_golden(golden, models):
    z, cfg, sd = golden("shape_dh64_qknorm")
    m = models("shape_dh64_qknorm")
    x = dev(z["ids"]).view(-1, 16, 16, 16).clone()
    x[:, 8:] = cfg.image_vocab_size
    for steps in (2, 8):
        s, _ = m.maskgit_generate(x.clone(), 8, maskgit_steps=steps, noise=dev(z[f"mg_s{steps}_noise"]))
        assert np.array_equal(s.cpu().numpy(), z[f"mg_s{steps}_samples"])


@pytest.mark.parametrize("name", ["anchor_c35", "anchor_c138"])
def test_full_size_anchor_golden(golden, models, name):
    z, cfg, sd = golden(name)
    m = models(name)
    ids = dev(z["ids"])
    x = ids.view(-1, 16, 16, 16).clone()
    x[:, 8:] = cfg.image_vocab_size
    out = m(x.view(1, -1), ids)
    assert abs(out.loss.item() - float(z["fwd_loss"])) < 1e-4
    assert abs(out.acc.item() - float(z["fwd_acc"])) < 1e-7
    lg = out.logits.cpu().numpy()
    probe = np.stack([lg[:, :, t, s // 16, s % 16] for t, s in zip(z["probe_t"], z["probe_s"])], 1)
    assert np.abs(probe - z["probe_logits"]).max() < 3e-4
    s, _ = m.maskgit_generate(x.clone(), 8, maskgit_steps=2, noise=dev(z["mg_s2_noise"]))
    assert np.array_equal(s.cpu().numpy(), z["mg_s2_samples"])
    print(name, "f16x3 CE delta", out.loss.item() - float(z["fwd_loss"]), "max|dlogit|",
          np.abs(probe - z["probe_logits"]).max())
