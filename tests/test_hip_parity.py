"""Parity of the HIP path (through the C ABI, via the drop-in modules) against the reference's golden
outputs and the CPU oracle.  Needs a real MI355X: run with ``-m gpu``.

Tolerances (exact = f32 MFMA path): |logit| differences <= 5e-5 on 2-layer fixtures (f32 accumulation
order only), CE within 1e-4 of the reference (the north-star bar), temperature-0 token ids bit-exact.
"""
import math
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import pkg
from oracle import genie_oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

TINY = ["tiny_ln", "tiny_qknorm", "tiny_mup", "tiny_qknorm_mup"]


@pytest.fixture(scope="module")
def models(golden):
    cache = {}

    def get(name, precision="exact"):
        key = (name, precision)
        if key not in cache:
            z, cfg, sd = golden(name)
            m = pkg("st_mask_git").STMaskGIT(cfg, precision=precision).load_numpy_state_dict(sd).to("cuda")
            cache[key] = m
        return cache[key]

    return get


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to("cuda")
    return t if dtype is None else t.to(dtype)


# ------------------------------------------------------------------------------------------ unit ops
@pytest.mark.parametrize("M,N,K", [(128, 192, 64), (100, 70, 32), (513, 1024, 256), (4096, 1536, 512), (37, 64, 2048)])
def test_linear(M, N, K):
    """genie_linear (f32 MFMA GEMM + fused epilogues) vs float64 numpy; asymmetric operands, ragged M/N."""
    att = pkg("attention")
    g = np.random.default_rng(M + N + K)
    x = g.standard_normal((M, K), dtype=np.float32)
    W = (g.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32)
    b = g.standard_normal(N, dtype=np.float32)
    y0 = g.standard_normal((M, N), dtype=np.float32)
    ref = x.astype(np.float64) @ W.astype(np.float64).T + b
    y = att.hip_linear(dev(x), dev(W), dev(b)).cpu().numpy()
    assert np.abs(y - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
    yg = att.hip_linear(dev(x), dev(W), dev(b), gelu=True).cpu().numpy()
    assert np.abs(yg - O.gelu_erf(ref)).max() < 3e-5 * max(1.0, np.abs(ref).max())
    out = dev(y0.copy())
    att.hip_linear(dev(x), dev(W), None, out=out, accumulate=True)
    assert np.abs(out.cpu().numpy() - (y0 + ref - b)).max() < 3e-5 * max(1.0, np.abs(ref).max())
    out = dev(y0.copy())   # accumulate with the bias (proj / fc2): residual and bias both read ahead of the tile's stores
    att.hip_linear(dev(x), dev(W), dev(b), out=out, accumulate=True)
    assert np.abs(out.cpu().numpy() - (y0 + ref)).max() < 3e-5 * max(1.0, np.abs(ref).max())


def test_layer_norm():
    lib = pkg("_lib")
    L = lib.load()
    g = np.random.default_rng(0)
    for rows, C in [(7, 64), (1000, 256), (333, 512), (5, 48)]:
        x = (g.standard_normal((rows, C)) * 3 + 1).astype(np.float32)
        gm, bt = g.standard_normal(C).astype(np.float32), g.standard_normal(C).astype(np.float32)
        y = torch.empty(rows, C, device="cuda")
        xx, gg, bb = dev(x), dev(gm), dev(bt)
        lib.check(L.genie_layer_norm(xx.data_ptr(), gg.data_ptr(), bb.data_ptr(), y.data_ptr(), rows, C, 1e-5,
                                     torch.cuda.current_stream().cuda_stream), "ln")
        ref = O.layer_norm(x.astype(np.float64), gm, bt)
        assert np.abs(y.cpu().numpy() - ref).max() < 1e-5


@pytest.mark.parametrize("d_model,qk_norm", [(32, False), (64, True), (64, False), (128, True), (128, False)])
@pytest.mark.parametrize("causal", [True, False])
def test_self_attention_module(d_model, qk_norm, causal):
    """The reference's own test shape (test_attention.py:5-18: num_heads=4, x=randn(1,16,d)) against the
    oracle restatement of BasicSelfAttention, plus a longer non-square case."""
    att = pkg("attention")
    g = np.random.default_rng(d_model)
    for Bn, N in [(1, 16), (3, 256)]:
        m = att.SelfAttention(num_heads=4, d_model=d_model, qk_norm=qk_norm, use_mup=False).to("cuda")
        sd = {"a.qkv.weight": g.standard_normal((3 * d_model, d_model), dtype=np.float32) / np.float32(np.sqrt(d_model)),
              "a.proj.weight": g.standard_normal((d_model, d_model), dtype=np.float32) / np.float32(np.sqrt(d_model)),
              "a.proj.bias": g.standard_normal(d_model, dtype=np.float32)}
        if qk_norm:
            sd["a.norm.weight"] = (1 + 0.1 * g.standard_normal(d_model // 4)).astype(np.float32)
            sd["a.norm.bias"] = (0.1 * g.standard_normal(d_model // 4)).astype(np.float32)
        m.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in sd.items()})
        x = g.standard_normal((Bn, N, d_model), dtype=np.float32)
        cfg = SimpleNamespace(num_heads=4, head_dim=d_model // 4, attn_scale=(d_model // 4) ** -0.5, qkv_bias=False,
                              qk_norm=qk_norm, proj_bias=True)
        ref = O.self_attention(x.astype(np.float64), sd, "a.", cfg, causal, O.F64)
        y = m(dev(x), causal=causal).cpu().numpy()
        assert np.abs(y - ref).max() < 1e-5  # the reference's own bar between its two back-ends is 1e-6 on N=16


@pytest.mark.parametrize("name", TINY)
def test_embed_and_block(golden, models, name):
    z, cfg, sd = golden(name)
    m = models(name)
    H = W = math.isqrt(cfg.S)
    ids = z["fwd_input"].reshape(-1, cfg.T, cfg.S)
    e = m.token_embed(dev(ids)).cpu().numpy() + sd["pos_embed_TSC"]
    ref = O.embed(ids, sd, cfg, O.F32)
    assert np.array_equal(e, ref) or np.abs(e - ref).max() < 1e-6
    blk = m.decoder.layers[0](dev(ref)).cpu().numpy()
    refb = O.st_block(ref.astype(np.float64), sd, 0, cfg, O.F64)
    assert np.abs(blk - refb).max() < 2e-5


# ------------------------------------------------------------------------------------------ golden parity
@pytest.mark.parametrize("name", TINY)
def test_compute_logits_golden(golden, models, name):
    z, cfg, sd = golden(name)
    m = models(name)
    H = W = math.isqrt(cfg.S)
    lg = m.compute_logits(dev(z["ids"]).view(-1, cfg.T, H, W)).cpu().numpy()
    assert lg.shape == z["logits"].shape
    scale = max(1.0, float(np.abs(z["logits"]).max()) / 8)
    assert np.abs(lg - z["logits"]).max() < 5e-5 * scale
    # frame-subset + token-major layout agree with the full BCTHW tensor
    sub = m.compute_logits_frames(dev(z["ids"]).view(-1, cfg.T, H, W), 1, 3, "token").cpu().numpy()
    full = lg.reshape(lg.shape[0], lg.shape[1], cfg.T, cfg.S)[:, :, 1:3].transpose(0, 2, 3, 1)
    assert np.array_equal(sub, full) or np.abs(sub - full).max() < 1e-5


@pytest.mark.parametrize("name", TINY)
def test_forward_loss_acc_golden(golden, models, name):
    z, cfg, sd = golden(name)
    m = models(name)
    out = m(dev(z["fwd_input"]), dev(z["ids"]))
    assert abs(out.loss.item() - float(z["fwd_loss"])) < 1e-4
    assert abs(out.acc.item() - float(z["fwd_acc"])) < 1e-7
    assert abs(out["logits"].double().sum().item() - float(z["fwd_logits_sum"])) < 0.05
    # compute_loss_and_acc on the returned logits (the reference's decomposition)
    H = W = math.isqrt(cfg.S)
    xin = dev(z["fwd_input"]).view(-1, cfg.T, H, W)
    loss2, acc2 = m.compute_loss_and_acc(out.logits, dev(z["ids"]).view(-1, cfg.T, H, W),
                                         xin[:, 1:] == m.mask_token_id)
    assert abs(loss2.item() - float(z["fwd_loss"])) < 1e-4 and abs(acc2.item() - float(z["fwd_acc"])) < 1e-7
    # fused path without logits
    s = m.ce_sums(xin, dev(z["ids"]).view(-1, cfg.T, H, W))
    assert abs((s[0] / s[2]).item() - float(z["fwd_loss"])) < 1e-4
    # no masked token -> nan (reference has no guard)
    assert math.isnan(m(dev(z["ids"]), dev(z["ids"])).loss.item())


@pytest.mark.parametrize("name", TINY)
@pytest.mark.parametrize("steps", [1, 2, 3, 8])
@pytest.mark.parametrize("mode", ["random", "greedy"])
def test_maskgit_golden(golden, models, name, steps, mode):
    z, cfg, sd = golden(name)
    m = models(name)
    H = W = math.isqrt(cfg.S)
    prompt = dev(z["ids"]).view(-1, cfg.T, H, W).clone()
    prompt[:, 2:] = cfg.image_vocab_size
    k = f"mg_s{steps}_{mode}"
    noise = dev(z[k + "_noise"]) if z[k + "_noise"].size else None
    s, fl = m.maskgit_generate(prompt, 2, maskgit_steps=steps, temperature=0.0, unmask_mode=mode, noise=noise)
    assert np.array_equal(s.cpu().numpy(), z[k + "_samples"])  # temperature-0 ids bit-exact
    assert np.array_equal(prompt.cpu().numpy(), z[k + "_prompt_after"])  # in-place write-back
    if steps == 2 and mode == "random":
        assert tuple(fl.shape) == z["mg_step0_factored_logits"].shape
        assert np.abs(fl.cpu().numpy() - z["mg_step0_factored_logits"]).max() < 1e-4


def test_maskgit_errors(golden, models):
    z, cfg, sd = golden("tiny_ln")
    m = models("tiny_ln")
    H = W = math.isqrt(cfg.S)
    prompt = dev(z["ids"]).view(-1, cfg.T, H, W).clone()
    with pytest.raises(AssertionError):
        m.maskgit_generate(prompt, 0)
    with pytest.raises(AssertionError):  # later frames not masked (st_mask_git.py:155)
        m.maskgit_generate(prompt.clone(), 2)
    prompt[:, 2:] = cfg.image_vocab_size
    with pytest.raises(NotImplementedError):
        m.maskgit_generate(prompt, 2, maskgit_steps=2, unmask_mode="bogus")
    # inputs that are not whole clips: the reference fails on them (its positional table does not broadcast, st_mask_git.py:261);
    # the library takes B from dim 0 and must not be handed less memory than B clips
    short = dev(z["ids"]).view(-1, cfg.T, H, W)[:, :cfg.T - 1]
    for call in (lambda: m.compute_logits(short), lambda: m.hidden_states(short), lambda: m.maskgit_generate(short.clone(), 1),
                 lambda: m(dev(z["ids"]), dev(z["ids"])[:1])):
        with pytest.raises(RuntimeError):
            call()
    with pytest.raises(RuntimeError):  # no CPU fallback
        m.maskgit_generate(prompt.cpu(), 2)
    with pytest.raises(AssertionError):
        m.generate(dev(z["ids"][:, :2 * cfg.S]), None, max_new_tokens=cfg.S + 1)


@pytest.mark.parametrize("name", TINY)
def test_generate_golden(golden, models, name):
    z, cfg, sd = golden(name)
    m = models(name)
    for kv in (True, False):   # one-frame passes against the temporal KV cache (the default) / the reference's full forwards
        out = m.generate(dev(z["ids"][:, :2 * cfg.S]), None, max_new_tokens=2 * cfg.S, return_logits=True,
                         maskgit_steps=2, temperature=0.0, noise=dev(z["gen_noise"]), kv_cache=kv)
        toks, gl = out
        assert np.array_equal(toks.cpu().numpy(), z["gen_out"]), kv
        if name == "tiny_ln":
            assert np.abs(gl.cpu().numpy() - z["gen_logits"]).max() < 1e-4, kv
    assert m.generate(dev(z["ids"][:, :2 * cfg.S]), None, max_new_tokens=cfg.S, maskgit_steps=1).shape == (z["ids"].shape[0], 3 * cfg.S)


@pytest.mark.parametrize("name", TINY)
def test_evaluator_golden(golden, models, name):
    z, cfg, sd = golden(name)
    ev_mod = pkg("evaluate")
    H = W = math.isqrt(cfg.S)
    args = SimpleNamespace(maskgit_steps=2, temperature=0, latent_h=H, latent_w=W)
    ev = ev_mod.GenieEvaluator(args, None, "cuda", model=models(name))
    samples, fl = ev.predict_zframe_logits(dev(z["ids"]), noise=dev(z["ev_noise"]))
    assert np.array_equal(samples.cpu().numpy(), z["ev_samples"])
    assert np.abs(fl.cpu().numpy() - z["ev_logits"]).max() < 1e-4
    loss = pkg("eval_utils").compute_loss(dev(z["ids"]), fl)
    assert abs(loss - float(z["ev_loss"])) < 1e-4
    sums = ev.evaluate_metric_sums(dev(z["ids"]), noise=dev(z["ev_noise"])).tolist()
    assert abs(sums[0] / sums[1] - float(z["ev_loss"])) < 1e-4
    assert abs(sums[2] / sums[3] - float(z["ev_acc"])) < 1e-7


@pytest.mark.parametrize("name", ["shape_dh32", "shape_dh64"])
def test_real_geometry_golden(golden, models, name):
    """T=16, S=256: forward CE, probe logits, 2-step MaskGIT ids and the full teacher-forced evaluate."""
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
    assert np.abs(probe - z["probe_logits"]).max() < 5e-5
    s, _ = m.maskgit_generate(x.clone(), 8, maskgit_steps=2, noise=dev(z["mg_s2_noise"]))
    assert np.array_equal(s.cpu().numpy(), z["mg_s2_samples"])
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
    """The shipped 35M config and the inferred GENIE_138M shape (L=32): CE within 1e-4 of the reference,
    2-step temperature-0 MaskGIT ids bit-exact."""
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


# ------------------------------------------------------------------------------------------ properties at size
def test_batch_independence_and_sampling_properties(golden, models):
    """Size-independent properties at a larger batch: clips are independent (row b of a batch == the
    single-clip run), 1-step MaskGIT == argmax of the logits, masked counts follow the cosine schedule."""
    z, cfg, sd = golden("shape_dh32")
    m = models("shape_dh32")
    ids = dev(pkg("synthetic").make_clips(6, cfg, seed=77))
    x = ids.view(-1, 16, 16, 16).clone()
    x[:, 5:] = cfg.image_vocab_size
    tok = m.compute_logits_frames(x, 5, 6, "token")  # (6,1,256,1024)
    one = m.compute_logits_frames(x[2:3].contiguous(), 5, 6, "token")
    assert torch.equal(tok[2:3], one)  # batch independence, bitwise
    s1, fl = m.maskgit_generate(x.clone(), 5, maskgit_steps=1)
    lo = tok[:, 0, :, :512].argmax(-1)
    hi = tok[:, 0, :, 512:].argmax(-1)
    assert torch.equal(s1.view(6, -1), hi * 512 + lo)
    # after a 3-step decode with the last step skipped nothing is left masked
    noise = torch.rand(2, 6, 256, device="cuda")
    p = x.clone()
    s3, _ = m.maskgit_generate(p, 5, maskgit_steps=3, noise=noise)
    assert int((s3 == cfg.image_vocab_size).sum()) == 0 and torch.equal(p[:, 5], s3)


def test_sample_temperature(golden, models):
    """temperature > 0 with caller-supplied uniforms == inverse-CDF sampling of the oracle."""
    z, cfg, sd = golden("tiny_ln")
    m = models("tiny_ln")
    H = W = math.isqrt(cfg.S)
    prompt = dev(z["ids"]).view(-1, cfg.T, H, W).clone()
    prompt[:, 2:] = cfg.image_vocab_size
    g = np.random.default_rng(3)
    u = g.random((1, 2, 2, cfg.S), dtype=np.float32)
    pn = prompt.cpu().numpy().copy()
    s, _ = m.maskgit_generate(prompt, 2, maskgit_steps=1, temperature=1.0, uniforms=dev(u))
    so, _ = O.maskgit_generate(pn, 2, sd, cfg, 1, 1.0, uniforms=u.reshape(1, 2, 2, H, W))
    agree = (s.cpu().numpy() == so).mean()
    assert agree >= 0.95  # cdf boundaries are f32 on the device, f64 in the oracle


def test_sample_temperature_exact_outside_cdf_margins():
    """temperature > 0, draw by draw: on the SAME logits and uniforms the kernel's inverse-CDF pick must equal the
    f64 inverse-CDF pick for every draw whose target u * sum lies further than 2e-6 (relative to the sum) from a CDF step -- the f32
    partial sums of the kernel cannot move a boundary by more than ~1e-7 -- and such ambiguous draws must be rare.  Confidence =
    product of the picked probabilities."""
    lib = pkg("_lib")
    L = lib.load()
    cfg = pkg("config").GenieConfig(num_layers=1, num_heads=2, d_model=64, T=4, S=256, num_factored_vocabs=2, qk_norm=False, use_mup=False)
    c = lib.make_cfg(cfg, lib.PREC_EXACT)
    g = np.random.default_rng(17)
    R, S, Vf = 24, cfg.S, 512
    logits = (g.standard_normal((R, S, 2 * Vf)) * 2.5).astype(np.float32)
    uni = g.random((2, R, S), dtype=np.float32)
    d_logits, d_uni = dev(logits), dev(uni)
    samples = torch.empty(R, S, dtype=torch.int64, device="cuda")
    conf = torch.empty(R, S, dtype=torch.float32, device="cuda")
    lib.check(L.genie_sample(c, d_logits.data_ptr(), lib.LAYOUT_TOKEN_MAJOR, R, 0.8, d_uni.data_ptr(), samples.data_ptr(),
                             conf.data_ptr(), torch.cuda.current_stream().cuda_stream), "genie_sample")
    got = samples.cpu().numpy()
    picks, ambiguous, want_conf = [], np.zeros((R, S), bool), np.ones((R, S))
    for k, f in enumerate((1, 0)):                       # the hi vocabulary is drawn first (st_mask_git.py:179)
        l = logits[:, :, f * Vf:(f + 1) * Vf].astype(np.float64)
        e = np.exp(l - l.max(-1, keepdims=True))
        tot = e.sum(-1)
        cdf = np.cumsum(e, -1)
        target = uni[k].astype(np.float64) * tot
        pick = np.minimum((cdf < target[..., None]).sum(-1), Vf - 1)
        ambiguous |= (np.abs(cdf - target[..., None]).min(-1) / tot) < 2e-6
        picks.append(pick)
        want_conf *= np.take_along_axis(e, pick[..., None], -1)[..., 0] / tot
    want = picks[0] * Vf + picks[1]
    assert ambiguous.mean() < 0.01, ambiguous.mean()
    clear = ~ambiguous
    assert np.array_equal(got[clear], want[clear]), int((got[clear] != want[clear]).sum())
    assert (got[ambiguous] == want[ambiguous]).mean() > 0.3 if ambiguous.any() else True
    np.testing.assert_allclose(conf.cpu().numpy()[clear], want_conf[clear], rtol=2e-5)


def test_bits_from_tokens():
    lib = pkg("_lib")
    L = lib.load()
    g = np.random.default_rng(1)
    ids = g.integers(0, 2 ** 18, size=(5, 256))
    zt = torch.empty(5, 18, 256, device="cuda")
    t = dev(ids)
    lib.check(L.genie_bits_from_tokens(t.data_ptr(), zt.data_ptr(), 5, 256, 18,
                                       torch.cuda.current_stream().cuda_stream), "bits")
    ref = O.bits_from_tokens(ids.reshape(5, 16, 16)).reshape(5, 18, 256)
    assert np.array_equal(zt.cpu().numpy(), ref)


def test_sample_temperature_distribution(golden):
    """temperature > 0: the reference draws sample ~ Categorical(probs / T) per factored vocabulary
    (st_mask_git.py:184-187; Categorical renormalises, so T only switches argmax -> sampling).  The kernel inverts the CDF on
    caller uniforms; 65,536 draws from ONE logits row must follow softmax(logits) as closely as torch's own Categorical does
    -- per-class frequencies within 5 sigma, total-variation distance at the sampling-noise level -- for both vocabularies,
    and the confidence is p_hi[s_hi] * p_lo[s_lo]."""
    z, cfg, sd = golden("shape_dh64")
    lib = pkg("_lib")
    L = lib.load()
    c = lib.make_cfg(cfg, lib.PREC_EXACT)
    g = torch.Generator(device="cuda").manual_seed(11)
    V, Vf, S, R = 1024, 512, cfg.S, 256
    row = torch.randn(V, device="cuda", generator=g) * 2.0
    logits = row.expand(R, S, V).contiguous()
    N = R * S
    for temperature in (1.0, 0.7):
        uni = torch.rand(2, R, S, device="cuda", generator=g)
        samples = torch.empty(R, S, dtype=torch.int64, device="cuda")
        conf = torch.empty(R, S, dtype=torch.float32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        lib.check(L.genie_sample(c, logits.data_ptr(), lib.LAYOUT_TOKEN_MAJOR, R, temperature, uni.data_ptr(),
                                 samples.data_ptr(), conf.data_ptr(), st), "genie_sample")
        s = samples.view(-1)
        lo, hi = s % Vf, s // Vf
        p_lo = torch.softmax(row[:Vf].double(), 0)
        p_hi = torch.softmax(row[Vf:].double(), 0)
        for name, draws, p in (("lo", lo, p_lo), ("hi", hi, p_hi)):
            f = torch.bincount(draws, minlength=Vf).double() / N
            sigma = torch.sqrt(p * (1 - p) / N)
            assert ((f - p).abs() <= 5 * sigma + 1e-9).all(), (name, temperature, ((f - p).abs() / sigma).max().item())
            tv = 0.5 * (f - p).abs().sum().item()
            ref = torch.distributions.Categorical(probs=(p / temperature).float()).sample((N,))   # the reference's sampler
            f_ref = torch.bincount(ref, minlength=Vf).double() / N
            tv_ref = 0.5 * (f_ref - p).abs().sum().item()
            assert tv < 1.5 * tv_ref + 2e-3, (name, temperature, tv, tv_ref)
        # the two vocabularies are drawn independently (separate uniforms): joint frequency of the top pair ~ product
        top = (int(p_hi.argmax()), int(p_lo.argmax()))
        pj = (p_hi[top[0]] * p_lo[top[1]]).item()
        fj = ((hi == top[0]) & (lo == top[1])).double().mean().item()
        assert abs(fj - pj) < 6 * (pj * (1 - pj) / N) ** 0.5
        want = (p_hi[hi] * p_lo[lo]).float()
        assert (conf.view(-1) - want).abs().max().item() < 1e-6


@pytest.mark.parametrize("temperature", [0.0, 0.8])
def test_sample_row_kernel_equals_strided_kernel(temperature):
    """Token-major logits of the 512-entry factored vocabularies take sample_rows_kernel (16-byte loads, values kept in registers);
    the same logits in the reference's (B, V, S) layout take the strided kernel.  Same operations in the same order: samples and
    confidences bit-identical (st_mask_git.py:171-190), argmax and inverse-CDF draws alike; ties included (duplicated maxima)."""
    lib = pkg("_lib")
    L = lib.load()
    cfg = pkg("config").GenieConfig(num_layers=1, num_heads=2, d_model=64, T=4, S=256, num_factored_vocabs=2, qk_norm=False, use_mup=False)
    c = lib.make_cfg(cfg, lib.PREC_EXACT)
    g = np.random.default_rng(23)
    R, S, Vf = 9, cfg.S, 512
    logits = (g.standard_normal((R, S, 2 * Vf)) * 2.5).astype(np.float32)
    logits[0, :, 100] = logits[0, :, 300] = 20.0       # ties: the first maximum wins in both kernels
    logits[1, :, Vf + 7] = logits[1, :, Vf + 8] = 19.0
    uni = g.random((2, R, S), dtype=np.float32)
    d_tok, d_uni = dev(logits), dev(uni)
    d_bvs = d_tok.permute(0, 2, 1).contiguous()         # (B, V, S)
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    for name, buf, layout in (("rows", d_tok, lib.LAYOUT_TOKEN_MAJOR), ("strided", d_bvs, lib.LAYOUT_BCTHW)):
        samples = torch.empty(R, S, dtype=torch.int64, device="cuda")
        conf = torch.empty(R, S, dtype=torch.float32, device="cuda")
        lib.check(L.genie_sample(c, buf.data_ptr(), layout, R, temperature, d_uni.data_ptr(), samples.data_ptr(), conf.data_ptr(), st),
                  "genie_sample")
        out[name] = (samples.cpu().numpy(), conf.cpu().numpy())
    assert np.array_equal(out["rows"][0], out["strided"][0])
    assert np.array_equal(out["rows"][1].view(np.uint32), out["strided"][1].view(np.uint32))
    if temperature == 0.0:
        assert (out["rows"][0][0] % Vf == 100).all() and (out["rows"][0][1] // Vf == 7).all()
