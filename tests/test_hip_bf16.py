"""The bf16 matrix-core ("fast") precision of the HIP path, pinned two ways (needs a GPU: -m gpu):
  1. against the oracle run under the same 16-bit operand contract (oracle.BF16_MFMA: identical rounding
     points, f32 accumulation) -- differences are accumulation order plus rare 1-ulp bf16 flips;
  2. against the f32 reference goldens, to MEASURE (and bound) what bf16 operands cost in CE / logits.
"""
import math

import numpy as np
import pytest

from conftest import pkg, record_measure
from oracle import genie_oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


@pytest.fixture(scope="module")
def models(golden):
    cache = {}

    def get(name):
        if name not in cache:
            z, cfg, sd = golden(name)
            cache[name] = pkg("st_mask_git").STMaskGIT(cfg, precision="bf16").load_numpy_state_dict(sd).to("cuda")
        return cache[name]

    return get


@pytest.mark.parametrize("prec", ["bf16", "f16x3"])
@pytest.mark.parametrize("M,N,K", [(128, 192, 64), (100, 70, 64), (513, 1024, 256), (4096, 1536, 512), (37, 64, 2048),
                                   (256, 1536, 512), (256, 512, 512), (256, 2048, 512), (256, 512, 2048), (250, 1024, 512),
                                   (2048, 512, 2048), (4096, 512, 512), (1000, 512, 512)])
def test_linear_lowp(prec, M, N, K):
    """genie_linear_lowp on pre-packed operands vs float64 on the SAME rounded operands (asymmetric, ragged M/N).  The
    256-row shapes are one frame of the batch-1 generate path: the split-K small-problem kernel (kernels_gemm_sm.hip)."""
    lib_mod = pkg("_lib")
    L = lib_mod.load()
    g = np.random.default_rng(M + N + K)
    x = g.standard_normal((M, K), dtype=np.float32)
    W = (g.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = g.standard_normal(N, dtype=np.float32)
    y0 = g.standard_normal((M, N), dtype=np.float32)
    st = torch.cuda.current_stream().cuda_stream
    npl, code, pack, rnd = (1, lib_mod.PREC_BF16, L.genie_pack_bf16, O.round_bf16) if prec == "bf16" else \
        (2, lib_mod.PREC_F16X3, L.genie_pack_split_f16, O.round_f16_split)
    xd, Wd, bd = dev(x), dev(W), dev(b)
    x16 = torch.empty(npl, M, K, dtype=torch.float16, device="cuda")
    W16 = torch.empty(npl, N, K, dtype=torch.float16, device="cuda")
    lib_mod.check(pack(xd.data_ptr(), x16.data_ptr(), x.size, st), "pack")
    lib_mod.check(pack(Wd.data_ptr(), W16.data_ptr(), W.size, st), "pack")
    ref = rnd(x).astype(np.float64) @ rnd(W).astype(np.float64).T + b
    y = torch.empty(M, N, device="cuda")
    lib_mod.check(L.genie_linear_lowp(code, x16.data_ptr(), W16.data_ptr(), bd.data_ptr(), y.data_ptr(), M, N, K, 0, 0,
                                      st), "lowp")
    tol = 3e-5 * max(1.0, np.abs(ref).max())
    assert np.abs(y.cpu().numpy() - ref).max() < tol
    if prec == "f16x3":  # and it is f32-class against the UNROUNDED operands
        exact = x.astype(np.float64) @ W.astype(np.float64).T + b
        assert np.abs(y.cpu().numpy() - exact).max() < tol
    yd = dev(y0.copy())
    lib_mod.check(L.genie_linear_lowp(code, x16.data_ptr(), W16.data_ptr(), 0, yd.data_ptr(), M, N, K, 1, 1, st), "lowp")
    assert np.abs(yd.cpu().numpy() - (y0 + O.gelu_erf(ref - b))).max() < 2 * tol
    # residual accumulate WITH the bias (proj / fc2 of the model: the epilogue reads both ahead of its stores)
    yd = dev(y0.copy())
    lib_mod.check(L.genie_linear_lowp(code, x16.data_ptr(), W16.data_ptr(), bd.data_ptr(), yd.data_ptr(), M, N, K, 0, 1, st), "lowp")
    assert np.abs(yd.cpu().numpy() - (y0 + ref)).max() < 2 * tol


@pytest.mark.parametrize("name", ["tiny_ln", "tiny_qknorm", "tiny_mup", "tiny_qknorm_mup"])
def test_logits_vs_bf16_oracle(golden, models, name):
    z, cfg, sd = golden(name)
    m = models(name)
    H = W = math.isqrt(cfg.S)
    ids = z["ids"].reshape(-1, cfg.T, H, W)
    lg = m.compute_logits(dev(ids)).cpu().numpy()
    ref16 = O.compute_logits(ids, sd, cfg, O.BF16_MFMA)
    scale = max(1.0, float(np.abs(ref16).max()) / 8)
    err16 = np.abs(lg - ref16)
    # same rounding points: the bulk agrees to f32 accumulation noise, a few entries see a 1-ulp bf16 flip upstream
    assert np.median(err16) < 2e-4 * scale
    assert err16.max() < 3e-2 * scale
    err32 = np.abs(lg - z["logits"])
    assert err32.max() < 0.25 * scale  # what bf16 operands cost vs the f32 reference (measured)


@pytest.mark.parametrize("name", ["tiny_ln", "tiny_qknorm"])
def test_ce_and_sampling_vs_bf16_oracle(golden, models, name):
    z, cfg, sd = golden(name)
    m = models(name)
    out = m(dev(z["fwd_input"]), dev(z["ids"]))
    loss16, acc16, _ = O.forward_loss_acc(z["fwd_input"], z["ids"], sd, cfg, O.BF16_MFMA)
    assert abs(out.loss.item() - loss16) < 2e-3
    record_measure(f"bf16_tiny[{name}].loss_minus_reference", out.loss.item() - float(z["fwd_loss"]))
    assert abs(out.loss.item() - float(z["fwd_loss"])) < 1.5e-2   # vs f32 reference on 768 tokens: measured 7e-4 / -1.4e-3 (profiles/r06_bf16_deltas.txt)
    H = W = math.isqrt(cfg.S)
    prompt = dev(z["ids"]).view(-1, cfg.T, H, W).clone()
    prompt[:, 2:] = cfg.image_vocab_size
    s, fl = m.maskgit_generate(prompt, 2, maskgit_steps=1)
    p_host = z["ids"].reshape(-1, cfg.T, H, W).copy()
    p_host[:, 2:] = cfg.image_vocab_size
    so, _ = O.maskgit_generate(p_host, 2, sd, cfg, 1, nm=O.BF16_MFMA)
    assert (s.cpu().numpy() == so).mean() > 0.9  # argmax of near-identical logits


@pytest.mark.parametrize("name", ["shape_dh32", "shape_dh64", "shape_dh64_qknorm"])
def test_real_geometry_vs_bf16_oracle(golden, models, name):
    z, cfg, sd = golden(name)
    m = models(name)
    ids = z["ids"]
    x = ids.reshape(-1, 16, 16, 16).copy()
    x[:, 8:] = cfg.image_vocab_size
    out = m(dev(x.reshape(1, -1)), dev(ids))
    loss16, acc16, lg16 = O.forward_loss_acc(x.reshape(1, -1), ids, sd, cfg, O.BF16_MFMA)
    assert abs(out.loss.item() - loss16) < 2e-3
    record_measure(f"bf16_real_geometry[{name}].loss_minus_reference", out.loss.item() - float(z["fwd_loss"]))
    assert abs(out.loss.item() - float(z["fwd_loss"])) < 7e-3   # measured 2e-5 ... -6.4e-4 (profiles/r06_bf16_deltas.txt)
    lg = out.logits.cpu().numpy()
    err = np.abs(lg - lg16)
    # same rounding points, but with 4096 tokens x several bf16 rounding points a handful of activations sit on
    # a bf16 rounding boundary and flip by one ulp between two f32 accumulation orders; the flips propagate.
    # Bound: well below the bf16-vs-f32 difference itself (~3e-2 max).
    assert np.median(err) < 4e-3 and err.max() < 8e-2


@pytest.mark.parametrize("name", ["shape_dh32", "shape_dh64"])
def test_real_geometry_vs_erf_gelu_bf16_oracle(golden, models, name):
    """The bf16 contract oracle follows the kernels' polynomial GELU (oracle.gelu_poly); THIS test anchors the contract to the reference's
    own nn.GELU (erf form, st_transformer.py:18): the same bf16 rounding points with gelu=None.  Budget: the polynomial's |Phi error| <=
    1.3e-5 is 0.7 % of a bf16 half-ulp of the hidden, so ~1 hidden value in 100 lands on the neighbouring bf16 -- the logits may sit a
    little further from this oracle than from the polynomial one, never by more than the bf16-vs-f32 distance itself."""
    z, cfg, sd = golden(name)
    m = models(name)
    ids = z["ids"]
    x = ids.reshape(-1, 16, 16, 16).copy()
    x[:, 8:] = cfg.image_vocab_size
    out = m(dev(x.reshape(1, -1)), dev(ids))
    erf_contract = O.Numerics(np.float32, O.round_bf16, temporal_qkv=O.round_bf16, gelu=None)
    loss_e, _, lg_e = O.forward_loss_acc(x.reshape(1, -1), ids, sd, cfg, erf_contract)
    _, _, lg_p = O.forward_loss_acc(x.reshape(1, -1), ids, sd, cfg, O.BF16_MFMA)
    lg = out.logits.cpu().numpy()
    err_e, err_p = np.abs(lg - lg_e), np.abs(lg - lg_p)
    print(f"{name}: |logit - erf-GELU contract| median {np.median(err_e):.2e} max {err_e.max():.2e}; polynomial contract median "
          f"{np.median(err_p):.2e} max {err_p.max():.2e}; CE delta vs erf contract {out.loss.item() - loss_e:+.2e}")
    assert abs(out.loss.item() - loss_e) < 3e-3
    assert np.median(err_e) < 6e-3 and err_e.max() < 0.1
    assert np.median(err_e) < 2.5 * np.median(err_p) + 1e-3


def test_full_size_anchor_bf16(golden, models):
    """C138-shape, 32 layers: report-level check that bf16 CE stays within a few 1e-2 of the f32 reference."""
    z, cfg, sd = golden("anchor_c138")
    m = models("anchor_c138")
    ids = dev(z["ids"])
    x = ids.view(-1, 16, 16, 16).clone()
    x[:, 8:] = cfg.image_vocab_size
    out = m(x.view(1, -1), ids)
    record_measure("bf16_anchor_c138.loss_minus_reference", out.loss.item() - float(z["fwd_loss"]))
    assert abs(out.loss.item() - float(z["fwd_loss"])) < 3e-3   # measured 2.5e-4 (profiles/r06_bf16_deltas.txt)
    lg = out.logits.cpu().numpy()
    probe = np.stack([lg[:, :, t, s // 16, s % 16] for t, s in zip(z["probe_t"], z["probe_s"])], 1)
    print("bf16 vs f32 reference: CE delta", out.loss.item() - float(z["fwd_loss"]), "max |dlogit|",
          np.abs(probe - z["probe_logits"]).max())
