"""GPU (-m gpu): the fused sub-block kernels of the shipped geometry (csrc/kernels_fused.hip; d 256, 8 heads of 32, T 16, bf16).

Each fused kernel must (a) agree with the launches it replaces -- same rounding points, so the difference is accumulation order
and one-ulp bf16 flips -- and (b) hold the bf16 bar of tests/test_hip_configs.py against the bf16-contract oracle
(oracle BF16_MFMA: st_transformer.py:70-83 / attention.py:36-61 with bf16 Linear operands)."""
import numpy as np
import pytest

from conftest import pkg
from oracle import genie_oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def _model(cfg, sd, fused, monkeypatch):
    monkeypatch.setenv("GENIE_NO_FUSED", "0" if fused else "1")
    m = pkg("st_mask_git").STMaskGIT(cfg, precision="bf16").load_numpy_state_dict(sd).to("cuda")
    m._weights()  # the opt-out is read when the weight table is built
    return m


def _uses_fused(m):
    return any(l.temporal.fused_w16 or l.mlp_fused_w16 for l in m._weights()[2])


@pytest.mark.parametrize("qkv_bias", [False, True])
def test_fused_blocks_match_unfused_and_oracle(monkeypatch, qkv_bias):
    cfg = pkg("config").GenieConfig(num_layers=2, num_heads=8, d_model=256, T=16, S=256, num_factored_vocabs=2, qk_norm=False,
                                    use_mup=False, qkv_bias=qkv_bias)
    synth = pkg("synthetic")
    sd = synth.make_state_dict(cfg, seed=77, law="conditioned")
    if qkv_bias:  # the synthetic law leaves biases at zero: make them count
        g = np.random.default_rng(5)
        for k in sd:
            if k.endswith("qkv.bias"):
                sd[k] = (0.05 * g.standard_normal(sd[k].shape)).astype(np.float32)
    B = 19  # 19 clips = 608 temporal blocks / 608 MLP blocks for 512 persistent workgroups: every ring carries over from one
            # block to the next, and the last round is ragged
    ids = synth.make_clips(B, cfg, seed=78)
    x = ids.reshape(B, 16, 16, 16).copy()
    x[:, 8:] = cfg.image_vocab_size
    mf = _model(cfg, sd, True, monkeypatch)
    assert _uses_fused(mf)
    _lib = pkg("_lib")
    lib = _lib.load()
    _lib.check(lib.genie_profile_enable(1 << _lib.KC_FUSED), "profile_enable")
    lib.genie_profile_reset()
    hf = mf.hidden_states(dev(x)).cpu().numpy().copy()
    import ctypes
    kbuf = ctypes.create_string_buffer(4096)
    _lib.check(lib.genie_profile_kernels(_lib.KC_FUSED, kbuf, len(kbuf)), "profile_kernels")
    lib.genie_profile_enable(0)
    launched = {ln.split("\t")[0].split(" ")[0]: int(float(ln.split("\t")[1])) for ln in kbuf.value.decode().splitlines()}
    print("fused launches:", launched)
    # proof of launch: every block ran its three sub-blocks as the fused kernels (19 clips = 304 sequences: above every size
    # threshold) -- nothing fell back to the unfused launches silently
    L = cfg.num_layers
    assert launched.get("spatial_attn_proj_bf16_kernel", 0) == L, launched
    assert launched.get("temporal_fused_bf16_kernel", 0) == L, launched
    assert sum(v for k, v in launched.items() if k.startswith("mlp_fused_bf16_kernel")) == L, launched
    mu = _model(cfg, sd, False, monkeypatch)
    assert not _uses_fused(mu)
    hu = mu.hidden_states(dev(x)).cpu().numpy().copy()
    scale = np.abs(hu).max()
    d = np.abs(hf - hu)
    print("fused vs unfused hidden: max", d.max(), "median", np.median(d), "scale", scale)
    assert np.isfinite(hf).all()
    assert np.median(d) < 2e-3 * scale and d.max() < 5e-2 * scale
    # against the bf16-contract oracle, first and last clip (the last clip's blocks sit in the ragged round)
    for b in (0, B - 1):
        ref = O.hidden_states(x[b:b + 1], sd, cfg, O.BF16_MFMA)[0]
        ef, eu = np.abs(hf[b] - ref), np.abs(hu[b] - ref)
        print("clip", b, "fused vs oracle max/median", ef.max(), np.median(ef), " unfused vs oracle", eu.max(), np.median(eu))
        assert np.median(ef) < 2e-3 * scale and ef.max() < 5e-2 * scale
        assert np.median(ef) < 2.0 * np.median(eu) + 1e-6     # no worse than the launches it replaces (same contract)
    # batch independence: the kernel a sub-block runs on depends on the batch (1 clip: unfused launches; 2-7: fused temporal / MLP, unfused
    # spatial; >= 8: all fused) -- the SAME rounding points and the same GELU everywhere (common.hpp gelu16_2), so a clip's hidden state
    # moves by summation order and one-ulp bf16 flips only: measured 2.5e-3 of the scale at the maximum, 2.5e-4 at the median
    for nb in (1, 2, 8):
        hb = mf.hidden_states(dev(x[4:4 + nb])).cpu().numpy()
        db = np.abs(hb - hf[4:4 + nb])
        print(f"clip 4 in a batch of {nb} vs in the batch of {B}: max {db.max():.3e} median {np.median(db):.3e} (scale {scale:.2f})")
        assert db.max() < 1e-2 * scale and np.median(db) < 6e-4 * scale
    # the readout takes the bf16 shadow of x that the last layer's fused MLP writes: logits of the last frames, fused vs unfused
    lf = mf.compute_logits_frames(dev(x), 14, 16, "token").cpu().numpy()
    lu = mu.compute_logits_frames(dev(x), 14, 16, "token").cpu().numpy()
    dl = np.abs(lf - lu)
    print("fused vs unfused logits: max", dl.max(), "median", np.median(dl))
    assert np.median(dl) < 4e-3 and dl.max() < 8e-2


# ---- the unit entry points of the C ABI (include/genie_hip.h: genie_temporal_fused_bf16, genie_mlp_fused_bf16) against the oracle's
# sub-block functions on the same random operands
def _bf16_bits_to_f32(t):
    return (t.view(torch.int16).cpu().numpy().astype(np.uint16).astype(np.uint32) << 16).view(np.float32)


def _unit_setup():
    _lib = pkg("_lib")
    cfgmod = pkg("config")
    c = cfgmod.c35()
    return _lib, _lib.load(), c, _lib.make_cfg(c, _lib.PREC_BF16), torch.cuda.current_stream().cuda_stream


def test_temporal_fused_entry_point_vs_oracle():
    """x += proj(causal_attention_T(qkv(x)))  (st_transformer.py:77-78, attention.py:36-61), 8 clips."""
    _lib, lib, c, cfg, st = _unit_setup()
    g = np.random.default_rng(11)
    B, T, S, D = 8, 16, 256, 256
    x = (g.standard_normal((B, T, S, D)) * 1.5).astype(np.float32)
    sd = {"p.qkv.weight": (g.standard_normal((768, 256)) * 0.06).astype(np.float32),
          "p.proj.weight": (g.standard_normal((256, 256)) * 0.06).astype(np.float32),
          "p.proj.bias": (g.standard_normal(256) * 0.05).astype(np.float32)}
    xd = dev(x)
    x16 = xd.to(torch.bfloat16)
    qw, pw, pb = dev(sd["p.qkv.weight"]), dev(sd["p.proj.weight"]), dev(sd["p.proj.bias"])
    tf = torch.empty(_lib.TEMPORAL_FUSED_ELEMS, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.genie_pack_temporal_fused_bf16(qw.data_ptr(), pw.data_ptr(), tf.data_ptr(), st), "pack")
    aw = _lib.AttnWeights()
    aw.fused_w16, aw.proj_b = tf.data_ptr(), pb.data_ptr()
    _lib.check(lib.genie_temporal_fused_bf16(cfg, aw, x16.data_ptr(), xd.data_ptr(), B, st), "temporal")
    got = xd.cpu().numpy()
    x_tc = x.transpose(0, 2, 1, 3).reshape(B * S, T, D)
    ref = x_tc + O.self_attention(x_tc, sd, "p.", c, True, O.BF16_MFMA, chunk=4096)
    ref = ref.reshape(B, S, T, D).transpose(0, 2, 1, 3)
    upd = np.abs(ref - x).max()                       # size of the update itself
    d = np.abs(got - ref)
    print("temporal unit: update", upd, "max err", d.max(), "median", np.median(d))
    assert np.isfinite(got).all() and d.max() < 2e-2 * upd and np.median(d) < 1e-3 * upd
    # fewer clips than the kernel takes: the entry point must say so, not compute garbage
    assert lib.genie_temporal_fused_bf16(cfg, aw, x16.data_ptr(), xd.data_ptr(), 1, st) != 0


@pytest.mark.parametrize("lnout", [False, True])
def test_mlp_fused_entry_point_vs_oracle(lnout):
    """x += fc2(gelu(fc1(LayerNorm(x))))  (st_transformer.py:81, 16-25) and its bf16 output: the copy of x, or the next block's norm1(x)."""
    _lib, lib, c, cfg, st = _unit_setup()
    g = np.random.default_rng(12)
    rows = 32768
    x = (g.standard_normal((rows, 256)) * 2.0 + g.standard_normal((rows, 1))).astype(np.float32)
    sd = {"p.fc1.weight": (g.standard_normal((1024, 256)) * 0.06).astype(np.float32), "p.fc1.bias": (g.standard_normal(1024) * 0.1).astype(np.float32),
          "p.fc2.weight": (g.standard_normal((256, 1024)) * 0.04).astype(np.float32), "p.fc2.bias": (g.standard_normal(256) * 0.1).astype(np.float32)}
    ln_g, ln_b = (1 + 0.2 * g.standard_normal(256)).astype(np.float32), (0.1 * g.standard_normal(256)).astype(np.float32)
    nx_g, nx_b = (1 + 0.2 * g.standard_normal(256)).astype(np.float32), (0.1 * g.standard_normal(256)).astype(np.float32)
    t = {k: dev(v) for k, v in sd.items()}
    lg, lb, ng, nb = dev(ln_g), dev(ln_b), dev(nx_g), dev(nx_b)
    mf = torch.empty(_lib.MLP_FUSED_ELEMS, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.genie_pack_mlp_fused_bf16(t["p.fc1.weight"].data_ptr(), t["p.fc2.weight"].data_ptr(), mf.data_ptr(), st), "pack")
    lw = _lib.LayerWeights()
    lw.mlp_fused_w16 = mf.data_ptr()
    lw.norm2_w, lw.norm2_b, lw.fc1_b, lw.fc2_b = lg.data_ptr(), lb.data_ptr(), t["p.fc1.bias"].data_ptr(), t["p.fc2.bias"].data_ptr()
    xd = dev(x)
    x16 = torch.zeros(rows, 256, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.genie_mlp_fused_bf16(cfg, lw, xd.data_ptr(), x16.data_ptr(), rows, ng.data_ptr() if lnout else 0,
                                        nb.data_ptr() if lnout else 0, st), "mlp")
    got, got16 = xd.cpu().numpy(), _bf16_bits_to_f32(x16)
    ref = x + O.mlp(O.layer_norm(x, ln_g, ln_b), sd, "p.", c, O.BF16_MFMA)
    upd = np.abs(ref - x).max()
    d = np.abs(got - ref)
    print("mlp unit: update", upd, "max err", d.max(), "median", np.median(d))
    assert np.isfinite(got).all() and d.max() < 2e-2 * upd and np.median(d) < 1e-3 * upd
    # the 16-bit output is a function of the kernel's OWN f32 result: exact bf16 rounding of it (copy), or of its LayerNorm
    want16 = O.round_bf16(O.layer_norm(got, nx_g, nx_b) if lnout else got)
    d16 = np.abs(got16 - want16)
    ulp = np.maximum(np.abs(want16), 2.0 ** -126) * 2.0 ** -7
    if lnout:
        # statistics summed in another order: rare one-ulp differences (plus f32 cancellation noise where the normalised value is ~0)
        assert (d16 <= ulp + 4e-6).all() and np.mean(d16 > 0) < 0.02
    else:
        assert (d16 == 0).all()
    assert lib.genie_mlp_fused_bf16(cfg, lw, xd.data_ptr(), 0, 100, 0, 0, st) != 0     # rows % 128 != 0: refused


def test_qkv_planes_from_the_mlp_kernel_match_the_qkv_gemm(monkeypatch):
    """Mode 2 of the fused MLP kernel (the next block's norm1 + spatial qkv Linear, st_transformer.py:74 / attention.py:37, written as the
    attention kernel's operand planes) against the same model with that hand-off switched off (LayerNorm'd row out, qkv GEMM launch):
    the same bf16 operands and rounding points, another summation order."""
    cfg = pkg("config").GenieConfig(num_layers=3, num_heads=8, d_model=256, T=16, S=256, num_factored_vocabs=2, qk_norm=False,
                                    use_mup=False)
    synth = pkg("synthetic")
    sd = synth.make_state_dict(cfg, seed=91, law="conditioned")
    B = 9
    ids = synth.make_clips(B, cfg, seed=92)
    x = ids.reshape(B, 16, 16, 16).copy()
    x[:, 10:] = cfg.image_vocab_size
    monkeypatch.setenv("GENIE_NO_FUSED_QKV", "0")
    ma = _model(cfg, sd, True, monkeypatch)
    assert all(l.spatial.w16_wide & pkg("_lib").FUSED_QKV_STREAM for l in ma._weights()[2])
    ha = ma.hidden_states(dev(x)).cpu().numpy().copy()
    monkeypatch.setenv("GENIE_NO_FUSED_QKV", "1")
    mb = _model(cfg, sd, True, monkeypatch)
    assert not any(l.spatial.w16_wide & pkg("_lib").FUSED_QKV_STREAM for l in mb._weights()[2])
    hb = mb.hidden_states(dev(x)).cpu().numpy().copy()
    scale = np.abs(hb).max()
    d = np.abs(ha - hb)
    print("planes from the MLP kernel vs qkv GEMM: max", d.max(), "median", np.median(d), "scale", scale)
    assert np.isfinite(ha).all() and d.max() < 5e-2 * scale and np.median(d) < 1e-3 * scale
    ref = O.hidden_states(x[:1], sd, cfg, O.BF16_MFMA)[0]
    ea, eb = np.abs(ha[0] - ref), np.abs(hb[0] - ref)
    print("vs oracle: with", ea.max(), np.median(ea), " without", eb.max(), np.median(eb))
    assert np.median(ea) < 2.0 * np.median(eb) + 1e-6


def test_mlp_fused_qkv_entry_point_planes_vs_oracle():
    """genie_mlp_fused_qkv_bf16: x += mlp(norm2(x)) and, from the result, the NEXT block's norm1 + spatial qkv Linear as the attention
    kernel's operand planes (st_transformer.py:81, then :74 / attention.py:37 of the following block).  Every plane element is
    compared with the oracle's value at the index the header documents (Q x scale x log2 e, K, V^T with its key order)."""
    _lib, lib, c, cfg, st = _unit_setup()
    g = np.random.default_rng(13)
    rows = 2 * 4096                      # 32 sequences of 256 rows
    x = (g.standard_normal((rows, 256)) * 2.0 + g.standard_normal((rows, 1))).astype(np.float32)
    sd = {"p.fc1.weight": (g.standard_normal((1024, 256)) * 0.06).astype(np.float32), "p.fc1.bias": (g.standard_normal(1024) * 0.1).astype(np.float32),
          "p.fc2.weight": (g.standard_normal((256, 1024)) * 0.04).astype(np.float32), "p.fc2.bias": (g.standard_normal(256) * 0.1).astype(np.float32)}
    ln_g, ln_b = (1 + 0.2 * g.standard_normal(256)).astype(np.float32), (0.1 * g.standard_normal(256)).astype(np.float32)
    nx_g, nx_b = (1 + 0.2 * g.standard_normal(256)).astype(np.float32), (0.1 * g.standard_normal(256)).astype(np.float32)
    wqkv = (g.standard_normal((768, 256)) * 0.08).astype(np.float32)
    t = {k: dev(v) for k, v in sd.items()}
    lg, lb, ng, nb, wq = dev(ln_g), dev(ln_b), dev(nx_g), dev(nx_b), dev(wqkv)
    mf = torch.empty(_lib.MLP_FUSED_ELEMS, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.genie_pack_mlp_fused_bf16(t["p.fc1.weight"].data_ptr(), t["p.fc2.weight"].data_ptr(), mf.data_ptr(), st), "pack")
    sf = torch.zeros(_lib.SPATIAL_PROJ_FUSED_ELEMS + _lib.SPATIAL_QKV_FUSED_ELEMS, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.genie_pack_spatial_qkv_fused_bf16(wq.data_ptr(), sf.data_ptr() + 2 * _lib.SPATIAL_PROJ_FUSED_ELEMS, st), "pack qkv")
    lw = _lib.LayerWeights()
    lw.mlp_fused_w16 = mf.data_ptr()
    lw.norm2_w, lw.norm2_b, lw.fc1_b, lw.fc2_b = lg.data_ptr(), lb.data_ptr(), t["p.fc1.bias"].data_ptr(), t["p.fc2.bias"].data_ptr()
    nx = _lib.LayerWeights()
    nx.norm1_w, nx.norm1_b = ng.data_ptr(), nb.data_ptr()
    nx.spatial.fused_w16, nx.spatial.w16_wide = sf.data_ptr(), _lib.FUSED_QKV_STREAM
    xd = dev(x)
    planes = torch.zeros(3, rows, 256, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.genie_mlp_fused_qkv_bf16(cfg, lw, nx, xd.data_ptr(), planes.data_ptr(), rows, st), "mlp_fused_qkv")
    got = xd.cpu().numpy()
    ref = x + O.mlp(O.layer_norm(x, ln_g, ln_b), sd, "p.", c, O.BF16_MFMA)
    upd = np.abs(ref - x).max()
    assert np.abs(got - ref).max() < 2e-2 * upd
    # the planes are a function of the kernel's own f32 rows: norm1 -> bf16 operands -> qkv -> bf16
    y = O.round_bf16(O.layer_norm(got, nx_g, nx_b))
    qkv = y @ O.round_bf16(wqkv).T                                    # (rows, 768): q | k | v, each 8 heads x 32
    pl = _bf16_bits_to_f32(planes).reshape(3, rows // 256, 8, -1)     # [part][seq][head][...]
    n_seq = rows // 256
    qk_scale = c.attn_scale * 1.4426950408889634
    want_q = (qkv[:, :256] * qk_scale).reshape(n_seq, 256, 8, 32).transpose(0, 2, 1, 3)      # [seq][head][pos][f]
    want_k = qkv[:, 256:512].reshape(n_seq, 256, 8, 32).transpose(0, 2, 1, 3)
    v = qkv[:, 512:].reshape(n_seq, 256, 8, 32).transpose(0, 2, 3, 1)                         # [seq][head][f][pos]
    perm = np.array([0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15])                   # stored position p' holds key perm[p'] of its 16-group
    want_v = v.reshape(n_seq, 8, 32, 16, 16)[..., perm].reshape(n_seq, 8, 32, 256)
    for name, gotp, want in (("q", pl[0].reshape(n_seq, 8, 256, 32), want_q), ("k", pl[1].reshape(n_seq, 8, 256, 32), want_k),
                             ("v^T", pl[2].reshape(n_seq, 8, 32, 256), want_v)):
        d = np.abs(gotp - want)
        tol = np.abs(want) * 2.0 ** -7 + 1e-2 * np.abs(want).max() * 2.0 ** -7     # one bf16 ulp of the value + a sliver of the scale
        bad = np.mean(d > tol)
        print(name, "plane: max err", d.max(), "beyond one ulp:", bad)
        assert bad < 2e-3 and d.max() < 0.05 * np.abs(want).max(), name
    # with a qkv bias the kernel must refuse (the driver then runs LayerNorm'd row + qkv GEMM)
    cb = _lib.make_cfg(pkg("config").GenieConfig(num_layers=32, num_heads=8, d_model=256, T=16, S=256, num_factored_vocabs=2,
                                                 qk_norm=False, use_mup=False, qkv_bias=True), _lib.PREC_BF16)
    assert lib.genie_mlp_fused_qkv_bf16(cb, lw, nx, xd.data_ptr(), planes.data_ptr(), rows, st) != 0


@pytest.mark.parametrize("n_seq", [128, 304, 1920])
@pytest.mark.parametrize("with_x16", [False, True])
def test_spatial_attn_proj_fused_entry_point_vs_reference_math(n_seq, with_x16):
    """genie_spatial_attn_proj_fused_bf16 on operand planes built HERE from random q, k, v in the formats of include/genie_hip.h
    (Q scale log2e, K head-major; V^T with its key order): x += proj(softmax(q k^T scale) v) over the 256 positions of each sequence,
    8 heads of 32 (attention.py:48-60 + the out-projection and residual of st_transformer.py:73-74), against the same math in f64
    on the bf16-rounded operands (the BF16_MFMA contract: bf16 q / k / v and out-projection operands, f32 softmax).  128 = the
    kernel's smallest problem (one workgroup per sequence, half the CUs idle), 304 = a ragged second round, 1,920 = the benchmark's
    128-clip pass."""
    _lib, lib, c, cfg, st = _unit_setup()
    g = torch.Generator(device="cuda").manual_seed(100 + n_seq)
    rows = n_seq * 256
    q = torch.randn(n_seq, 8, 256, 32, device="cuda", generator=g) * 1.2
    k = torch.randn(n_seq, 8, 256, 32, device="cuda", generator=g) * 1.2
    v = torch.randn(n_seq, 8, 256, 32, device="cuda", generator=g)
    wp = torch.randn(256, 256, device="cuda", generator=g) * 0.06
    pb = torch.randn(256, device="cuda", generator=g) * 0.05
    x = torch.randn(rows, 256, device="cuda", generator=g) * 1.5
    scale = float(c.attn_scale)
    q16 = (q * (scale * 1.4426950408889634)).to(torch.bfloat16)     # what the qkv GEMM's epilogue stores
    k16, v16 = k.to(torch.bfloat16), v.to(torch.bfloat16)
    perm = torch.tensor([0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15], device="cuda")
    vt = v16.transpose(2, 3).reshape(n_seq, 8, 32, 16, 16)[..., perm].reshape(n_seq, 8, 32, 256)   # stored position p' holds key perm[p']
    planes = torch.cat([q16.reshape(-1), k16.reshape(-1), vt.reshape(-1)]).contiguous()
    assert planes.numel() == 3 * rows * 256
    sf = torch.zeros(_lib.SPATIAL_PROJ_FUSED_ELEMS + _lib.SPATIAL_QKV_FUSED_ELEMS, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.genie_pack_spatial_proj_fused_bf16(wp.data_ptr(), sf.data_ptr(), st), "pack proj")
    aw = _lib.AttnWeights()
    aw.fused_w16, aw.proj_b = sf.data_ptr(), pb.data_ptr()
    xd = x.clone()
    x16 = torch.zeros(rows, 256, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.genie_spatial_attn_proj_fused_bf16(cfg, aw, planes.data_ptr(), xd.data_ptr(), x16.data_ptr() if with_x16 else 0, n_seq,
                                                      st), "spatial fused")
    # reference math in f64 on the same bf16 operands, chunked over sequences
    upd = torch.empty_like(x)
    wp64 = wp.to(torch.bfloat16).double()
    for s0 in range(0, n_seq, 64):
        sl = slice(s0, min(s0 + 64, n_seq))
        sc = (q16[sl].double() / 1.4426950408889634) @ k16[sl].double().transpose(2, 3)       # = q k^T scale
        o = torch.softmax(sc, dim=-1) @ v16[sl].double()                                       # (n, 8, 256, 32)
        o16 = o.transpose(1, 2).reshape(-1, 256).float().to(torch.bfloat16).double()          # attention output as proj operand
        upd[sl.start * 256:sl.stop * 256] = (o16 @ wp64.T + pb.double()).float()
    ref = x + upd
    u = upd.abs().max().item()
    d = (xd - ref).abs()
    print("spatial fused unit: n_seq", n_seq, "update", u, "max err", d.max().item(), "median", d.median().item())
    assert torch.isfinite(xd).all() and d.max().item() < 2e-2 * u and d.median().item() < 1e-3 * u
    if with_x16:
        assert torch.equal(x16, xd.to(torch.bfloat16))       # the 16-bit copy is the bf16 rounding of the kernel's own f32 result
    else:
        assert (x16 == 0).all()
    # below the kernel's size threshold the entry point must refuse, not compute garbage
    assert lib.genie_spatial_attn_proj_fused_bf16(cfg, aw, planes.data_ptr(), xd.data_ptr(), 0, 64, st) == _lib.E_UNSUPPORTED


# ---- the prefix-cache passes of the evaluator (evaluate.py:107-116) on the fused temporal kernel (csrc/kernels_fused_prefix.hip)
def _prefix_passes(m, ids, n, frame0, masked_ids):
    """genie_clean_pass over frames 0..n-1, then one genie_masked_frames_logits on `masked_ids` (B, n, S): logits (B, n, S, V)."""
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


@pytest.mark.parametrize("qkv_bias,n,frame0,B", [(False, 15, 1, 19), (True, 15, 1, 3), (False, 12, 1, 5), (True, 9, 0, 2), (False, 15, 0, 4)])
def test_prefix_passes_fused_temporal_matches_unfused(monkeypatch, qkv_bias, n, frame0, B):
    """Clean pass + masked pass with the temporal sub-block as ONE kernel each (K / V fragment images in the cache) against the
    launches they replace (qkv GEMM -> bf16 qkv rows in the cache, attn_temporal(_prefix)_f32_mfma, proj GEMM): same rounding
    points, so the logits differ by accumulation order and one-ulp bf16 flips.  n < 15: phantom frame slots; frame0 = 0: slot i sees
    cached frames < i only; 19 clips: persistent workgroups carry their ring from block to block, ragged last round."""
    cfg = pkg("config").GenieConfig(num_layers=3, num_heads=8, d_model=256, T=16, S=256, num_factored_vocabs=2, qk_norm=False,
                                    use_mup=False, qkv_bias=qkv_bias)
    synth = pkg("synthetic")
    sd = synth.make_state_dict(cfg, seed=177, law="conditioned")
    if qkv_bias:
        g = np.random.default_rng(6)
        for k in sd:
            if k.endswith("qkv.bias"):
                sd[k] = (0.05 * g.standard_normal(sd[k].shape)).astype(np.float32)
    ids = dev(synth.make_clips(B, cfg, seed=178).reshape(B, 16, 256))
    g = torch.Generator(device="cpu").manual_seed(5)
    masked = ids[:, frame0:frame0 + n].clone()
    hide = (torch.rand(masked.shape, generator=g) < 0.6).to("cuda")
    masked[hide] = cfg.image_vocab_size
    masked = masked.contiguous()
    _lib = pkg("_lib")
    lib = _lib.load()
    mf = _model(cfg, sd, True, monkeypatch)
    _lib.check(lib.genie_profile_enable(1 << _lib.KC_FUSED), "profile_enable")
    lib.genie_profile_reset()
    lf = _prefix_passes(mf, ids, n, frame0, masked)
    import ctypes
    kbuf = ctypes.create_string_buffer(4096)
    _lib.check(lib.genie_profile_kernels(_lib.KC_FUSED, kbuf, len(kbuf)), "profile_kernels")
    lib.genie_profile_enable(0)
    launched = {ln.split("\t")[0].split(" ")[0]: int(float(ln.split("\t")[1])) for ln in kbuf.value.decode().splitlines()}
    print("fused launches:", launched)
    L = cfg.num_layers
    assert launched.get("temporal_prefix_fused_bf16_kernel<1>", 0) == L, launched     # clean pass, every layer (the last one for its K / V only)
    assert launched.get("temporal_prefix_fused_bf16_kernel<2>", 0) == L, launched     # masked pass
    mu = _model(cfg, sd, False, monkeypatch)
    lu = _prefix_passes(mu, ids, n, frame0, masked)
    d = np.abs(lf - lu)
    print("prefix passes fused vs unfused logits: max", d.max(), "median", np.median(d), "scale", np.abs(lu).max())
    assert np.isfinite(lf).all()
    assert np.median(d) < 4e-3 and d.max() < 8e-2
    # and against the full forward on the clean clip where the two coincide: with nothing masked, slot i of the masked pass IS frame
    # frame0 + i of the plain forward (the cached keys are the clean ones, its own key is clean too)
    if frame0 == 0:
        lc = _prefix_passes(mf, ids, n, 0, ids[:, :n].contiguous())
        full = mf.compute_logits_frames(ids.view(B, 16, 16, 16), 0, n, "token").cpu().numpy().reshape(B, n, 256, -1)
        dc = np.abs(lc - full)
        print("unmasked prefix pass vs full forward: max", dc.max(), "median", np.median(dc))
        assert np.median(dc) < 4e-3 and dc.max() < 8e-2


def test_temporal_prefix_fused_entry_point_vs_oracle():
    """genie_temporal_prefix_fused_bf16 on random operands, 8 clips x 15 frames: mode 1 (clean pass) = x += proj(causal_attention(qkv(x))) over
    the 15 frames (st_transformer.py:77-78, attention.py:36-61) against the bf16-contract oracle; mode 2 with shift 0 on the SAME input reads the
    fragment images mode 1 left: slot i sees the cached keys j < i and its own -- which are the clean ones -- so it must reproduce the same update."""
    _lib, lib, c, cfg, st = _unit_setup()
    g = np.random.default_rng(12)
    B, nf, S, D = 8, 15, 256, 256
    x = (g.standard_normal((B, nf, S, D)) * 1.5).astype(np.float32)
    sd = {"p.qkv.weight": (g.standard_normal((768, 256)) * 0.06).astype(np.float32),
          "p.proj.weight": (g.standard_normal((256, 256)) * 0.06).astype(np.float32),
          "p.proj.bias": (g.standard_normal(256) * 0.05).astype(np.float32)}
    qw, pw, pb = dev(sd["p.qkv.weight"]), dev(sd["p.proj.weight"]), dev(sd["p.proj.bias"])
    tf = torch.empty(_lib.TEMPORAL_FUSED_ELEMS, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.genie_pack_temporal_fused_bf16(qw.data_ptr(), pw.data_ptr(), tf.data_ptr(), st), "pack")
    aw = _lib.AttnWeights()
    aw.fused_w16, aw.proj_b = tf.data_ptr(), pb.data_ptr()
    kv = torch.zeros(B * S * 16 * 1024, dtype=torch.uint8, device="cuda")
    x1 = dev(x)
    _lib.check(lib.genie_temporal_prefix_fused_bf16(cfg, aw, x1.data_ptr(), kv.data_ptr(), B, nf, 1, 0, st), "mode 1")
    x2 = dev(x)
    _lib.check(lib.genie_temporal_prefix_fused_bf16(cfg, aw, x2.data_ptr(), kv.data_ptr(), B, nf, 2, 0, st), "mode 2")
    x_tc = x.transpose(0, 2, 1, 3).reshape(B * S, nf, D)
    ref = x_tc + O.self_attention(x_tc, sd, "p.", c, True, O.BF16_MFMA, chunk=4096)
    ref = ref.reshape(B, S, nf, D).transpose(0, 2, 1, 3)
    upd = np.abs(ref - x).max()
    for name, got in (("mode 1", x1.cpu().numpy()), ("mode 2", x2.cpu().numpy())):
        d = np.abs(got - ref)
        print(name, "update", upd, "max err", d.max(), "median", np.median(d))
        assert np.isfinite(got).all() and d.max() < 2e-2 * upd and np.median(d) < 1e-3 * upd
    # the pass must have fewer frames than the model (a full-length cache is genie_frame_pass's): refused, nothing computed
    assert lib.genie_temporal_prefix_fused_bf16(cfg, aw, x1.data_ptr(), kv.data_ptr(), B, 16, 1, 0, st) != 0
