"""GPU (-m gpu): the fused temporal qkv + attention kernel of the shipped geometry in f16x3 (csrc/kernels_fused_f16x3.hip; d 256, 8 heads of 32).

It replaces the temporal qkv GEMM and attn_temporal_(prefix_)f32_mfma_kernel with the same arithmetic (split-f16 Linear operands, f32
accumulation, f32 attention): the two paths differ by summation order only, so they must agree to f32 noise -- in the plain forward
(st_transformer.py:77-78, attention.py:36-58) and in the evaluator's prefix-cache passes (evaluate.py:107-116), where the cache holds the
kernel's k / v accumulators instead of qkv rows.  The reference-level checks of this path are the f16x3 goldens of
tests/test_hip_bench_config.py (ev_c35: the reference's own run of the shipped config, ids bit-exact) -- they run through this kernel."""
import ctypes

import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def _cfg_sd(qkv_bias, layers=3, seed=277):
    cfg = pkg("config").GenieConfig(num_layers=layers, num_heads=8, d_model=256, T=16, S=256, num_factored_vocabs=2, qk_norm=False,
                                    use_mup=False, qkv_bias=qkv_bias)
    sd = pkg("synthetic").make_state_dict(cfg, seed=seed, law="conditioned")
    if qkv_bias:
        g = np.random.default_rng(7)
        for k in sd:
            if k.endswith("qkv.bias"):
                sd[k] = (0.05 * g.standard_normal(sd[k].shape)).astype(np.float32)
    return cfg, sd


def _model(cfg, sd, fused, monkeypatch):
    monkeypatch.setenv("GENIE_NO_FUSED", "0" if fused else "1")
    m = pkg("st_mask_git").STMaskGIT(cfg, precision="f16x3").load_numpy_state_dict(sd).to("cuda")
    layers = m._weights()[2]
    assert all(bool(l.temporal.fused_w16) == fused for l in layers)
    return m


def _launched(fn):
    _lib = pkg("_lib")
    lib = _lib.load()
    _lib.check(lib.genie_profile_enable(1 << _lib.KC_FUSED), "profile_enable")
    lib.genie_profile_reset()
    out = fn()
    kbuf = ctypes.create_string_buffer(4096)
    _lib.check(lib.genie_profile_kernels(_lib.KC_FUSED, kbuf, len(kbuf)), "profile_kernels")
    lib.genie_profile_enable(0)
    return out, {ln.split("\t")[0].split(" ")[0]: int(float(ln.split("\t")[1])) for ln in kbuf.value.decode().splitlines()}


@pytest.mark.parametrize("qkv_bias,B", [(False, 19), (True, 3)])
def test_forward_fused_temporal_matches_unfused(monkeypatch, qkv_bias, B):
    """19 clips: 1,216 blocks for 512 persistent workgroups -- every ring carries over from block to block, ragged last round."""
    cfg, sd = _cfg_sd(qkv_bias)
    synth = pkg("synthetic")
    x = synth.make_clips(B, cfg, seed=278).reshape(B, 16, 16, 16).copy()
    x[:, 8:] = cfg.image_vocab_size
    mf = _model(cfg, sd, True, monkeypatch)
    hf, launched = _launched(lambda: mf.hidden_states(dev(x)).cpu().numpy().copy())
    print("fused launches:", launched)
    assert launched.get("temporal_qkv_attn_f16x3_kernel<0>", 0) == cfg.num_layers, launched
    mu = _model(cfg, sd, False, monkeypatch)
    hu = mu.hidden_states(dev(x)).cpu().numpy().copy()
    scale = np.abs(hu).max()
    d = np.abs(hf - hu)
    print("fused vs unfused hidden: max", d.max(), "median", np.median(d), "scale", scale)
    assert np.isfinite(hf).all()
    assert d.max() < 2e-5 * scale and np.median(d) < 1e-6 * scale
    # batch independence: a clip's result does not depend on what else is in the batch beyond f32 summation order (the GEMMs around this
    # kernel pick their tiling by the batch; the kernel itself walks a clip's positions the same way in any batch)
    hb = mf.hidden_states(dev(x[1:3])).cpu().numpy()
    assert np.abs(hb - hf[1:3]).max() < 2e-5 * scale


def _prefix_passes(m, ids, n, frame0, masked_ids):
    _lib = pkg("_lib")
    lib = _lib.load()
    cfg, w = m._weights()[:2]
    B, S = ids.shape[0], m.config.S
    V = m.config.factored_vocab_size * m.config.num_factored_vocabs
    ws = m._workspace(B)
    nbytes = lib.genie_prefix_cache_bytes(cfg, B)
    cache = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ctx = ids[:, :n].contiguous()
    _lib.check(lib.genie_clean_pass(cfg, w, ctx.data_ptr(), B, n, n, cache.data_ptr(), nbytes, ws.data_ptr(), ws.numel(), st), "clean")
    logits = torch.empty(B, n, S, V, dtype=torch.float32, device="cuda")
    _lib.check(lib.genie_masked_frames_logits(cfg, w, masked_ids.data_ptr(), B, frame0, n, cache.data_ptr(), nbytes, logits.data_ptr(),
                                              ws.data_ptr(), ws.numel(), st), "masked")
    torch.cuda.synchronize()
    return logits.cpu().numpy()


@pytest.mark.parametrize("qkv_bias,n,frame0,B", [(False, 15, 1, 19), (True, 15, 1, 3), (False, 12, 1, 5), (True, 11, 0, 2), (False, 15, 0, 4)])
def test_prefix_passes_fused_temporal_matches_unfused(monkeypatch, qkv_bias, n, frame0, B):
    """Clean pass (k / v accumulators into the cache) + masked pass (read back) against qkv GEMM -> f32 qkv rows in the cache ->
    attn_temporal(_prefix)_f32_mfma.  n < 15: phantom frame slots; frame0 = 0: slot i sees cached frames < i only."""
    cfg, sd = _cfg_sd(qkv_bias, seed=377)
    synth = pkg("synthetic")
    ids = dev(synth.make_clips(B, cfg, seed=378).reshape(B, 16, 256))
    g = torch.Generator(device="cpu").manual_seed(5)
    masked = ids[:, frame0:frame0 + n].clone()
    hide = (torch.rand(masked.shape, generator=g) < 0.6).to("cuda")
    masked[hide] = cfg.image_vocab_size
    masked = masked.contiguous()
    mf = _model(cfg, sd, True, monkeypatch)
    lf, launched = _launched(lambda: _prefix_passes(mf, ids, n, frame0, masked))
    print("fused launches:", launched)
    assert launched.get("temporal_qkv_attn_f16x3_kernel<1>", 0) == cfg.num_layers, launched
    assert launched.get("temporal_qkv_attn_f16x3_kernel<2>", 0) == cfg.num_layers, launched
    mu = _model(cfg, sd, False, monkeypatch)
    lu = _prefix_passes(mu, ids, n, frame0, masked)
    d = np.abs(lf - lu)
    scale = np.abs(lu).max()
    print("prefix passes fused vs unfused logits: max", d.max(), "median", np.median(d), "scale", scale)
    assert np.isfinite(lf).all()
    assert d.max() < 2e-5 * max(scale, 1.0) and np.median(d) < 1e-6 * max(scale, 1.0)
    if frame0 == 0:   # nothing masked: slot i of the masked pass IS frame i of the plain forward
        lc = _prefix_passes(mf, ids, n, 0, ids[:, :n].contiguous())
        full = mf.compute_logits_frames(ids.view(B, 16, 16, 16), 0, n, "token").cpu().numpy().reshape(B, n, 256, -1)
        dc = np.abs(lc - full)
        print("unmasked prefix pass vs full forward: max", dc.max(), "median", np.median(dc))
        assert dc.max() < 2e-5 * max(scale, 1.0)


@pytest.mark.parametrize("nf", [16, 13])
def test_temporal_qkv_attn_entry_point_vs_oracle(nf):
    """genie_temporal_qkv_attn_f16x3 on random operands, 6 clips: a = causal_attention_T(qkv_t(x)) (attention.py:36-58) as [hi | lo'] planes
    against the oracle's f16x3 contract (split Linear operands, f32 everything else) with an identity out-projection -- the oracle then
    returns the 22-bit rounding of the attention output, which is what the planes hold.  nf = 13: modes 1 and 2 (shift 0 on the same input:
    the cached keys ARE the own ones) must reproduce mode 0's result; phantom frame slots."""
    from oracle import genie_oracle as O
    _lib = pkg("_lib")
    lib = _lib.load()
    c = pkg("config").c35()
    cfg = _lib.make_cfg(c, _lib.PREC_F16X3)
    st = torch.cuda.current_stream().cuda_stream
    g = np.random.default_rng(13)
    B, S, D = 6, 256, 256
    x = (g.standard_normal((B, nf, S, D)) * 1.5).astype(np.float32)
    sd = {"p.qkv.weight": (g.standard_normal((768, 256)) * 0.06).astype(np.float32), "p.proj.weight": np.eye(256, dtype=np.float32)}
    qw = dev(sd["p.qkv.weight"])
    ts = torch.empty(_lib.TEMPORAL_QKV_F16X3_ELEMS, dtype=torch.float16, device="cuda")
    _lib.check(lib.genie_pack_temporal_qkv_f16x3(qw.data_ptr(), ts.data_ptr(), st), "pack")
    aw = _lib.AttnWeights()
    aw.fused_w16 = ts.data_ptr()
    M = B * nf * S
    xd = dev(x)
    kv = torch.zeros(B * S * 8 * 1024, dtype=torch.float32, device="cuda")
    x_tc = x.transpose(0, 2, 1, 3).reshape(B * S, nf, D)
    import dataclasses
    cc = dataclasses.replace(c, proj_bias=False)   # (the identity out-projection of this test has no bias)
    ref = O.self_attention(x_tc, sd, "p.", cc, True, O.F16X3, chunk=4096).reshape(B, S, nf, D).transpose(0, 2, 1, 3)
    scale = np.abs(ref).max()
    modes = [0] if nf == 16 else [0, 1, 2]
    for mode in modes:
        planes = torch.zeros(2, M, D, dtype=torch.float16, device="cuda")
        _lib.check(lib.genie_temporal_qkv_attn_f16x3(cfg, aw, xd.data_ptr(), planes.data_ptr(), M * D, kv.data_ptr(), B, nf, mode, 0, st), f"mode {mode}")
        got = (planes[0].float() + planes[1].float() / 2048.0).cpu().numpy().reshape(B, nf, S, D)
        d = np.abs(got - ref)
        print("mode", mode, "max err", d.max(), "median", np.median(d), "scale", scale)
        assert np.isfinite(got).all() and d.max() < 3e-5 * scale and np.median(d) < 2e-6 * scale
    # a cache pass needs 11 <= nframes < T: refused otherwise, nothing computed
    planes = torch.zeros(2, M, D, dtype=torch.float16, device="cuda")
    assert lib.genie_temporal_qkv_attn_f16x3(cfg, aw, xd.data_ptr(), planes.data_ptr(), M * D, kv.data_ptr(), B, nf, 1, 0, st) == (0 if nf == 13 else _lib.E_UNSUPPORTED)
