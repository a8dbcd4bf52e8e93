"""The one-frame passes of generate on the fragment-order kernels (csrc/kernels_frame.hip; generate.py:81-95,
st_mask_git.py:163-169 restricted to the frame being decoded): weight packing bit-exact, every frame pass equal to the same
frames of the full 16-frame forward (temporal attention is causal), two frames per pass equal to one by one, the merged
commit + first-step schedule of generate equal to the plain KV-cache schedule and to the reference goldens.  Needs a GPU: -m gpu."""
import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def _model(d, heads, layers=2, precision="f16x3", seed=11, **kw):
    kw = dict(dict(qk_norm=False, use_mup=False), **kw)
    cfg = pkg("config").GenieConfig(num_layers=layers, num_heads=heads, d_model=d, T=16, S=256, num_factored_vocabs=2, **kw)
    sd = pkg("synthetic").make_state_dict(cfg, seed=seed, law="conditioned")
    g = np.random.default_rng(seed + 1)
    for k in sd:       # the synthetic law leaves biases at zero: make every bias the kernels add count
        if k.endswith(".bias") and "norm" not in k:
            sd[k] = (0.05 * g.standard_normal(sd[k].shape)).astype(np.float32)
    return cfg, pkg("st_mask_git").STMaskGIT(cfg, precision=precision).load_numpy_state_dict(sd).to("cuda")


def test_pack_frame_w16_is_the_row_major_split_in_fragment_order():
    """genie_pack_frame_w16: block (32 rows x 64 k) -> [plane][step] fragments, lane 32 h + r = row r, k = 16 step + 8 h .. + 7;
    the same hi / lo' values as genie_pack_split_f16 (bit-exact)."""
    _lib = pkg("_lib")
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    N, K = 96, 192
    g = torch.Generator(device="cpu").manual_seed(5)
    W = (torch.randn(N, K, generator=g) * torch.logspace(-6, 1, K)[None, :]).cuda().contiguous()
    rm = torch.empty(2, N, K, dtype=torch.float16, device="cuda")
    fr = torch.empty(2 * N * K, dtype=torch.float16, device="cuda")
    _lib.check(lib.genie_pack_split_f16(W.data_ptr(), rm.data_ptr(), W.numel(), st), "pack_split")
    _lib.check(lib.genie_pack_frame_w16(W.data_ptr(), fr.data_ptr(), N, K, st), "pack_frame")
    rm = rm.cpu().numpy().view(np.uint16)
    got = fr.cpu().numpy().view(np.uint16).reshape(N // 32, K // 64, 2, 4, 2, 32, 8)   # rb, kb, plane, step, h, r, e
    want = rm.reshape(2, N // 32, 32, K // 64, 4, 2, 8).transpose(1, 3, 0, 4, 5, 2, 6)  # p, rb, r, kb, s, h, e -> rb, kb, p, s, h, r, e
    assert np.array_equal(got, want)
    assert lib.genie_pack_frame_w16(W.data_ptr(), fr.data_ptr(), 48, 64, st) == _lib.E_SHAPE


@pytest.mark.parametrize("d,heads,kw", [(512, 8, {}), (256, 4, {}), (128, 2, {}), (256, 4, dict(qkv_bias=True)),
                                        (128, 2, dict(qkv_bias=True, proj_bias=False, mlp_bias=False)),
                                        (256, 8, {}),                       # the shipped geometry magvit_n32_h8_d256: heads of 32
                                        (512, 16, dict(qkv_bias=True)), (128, 4, {}),
                                        # the reference's default attention variant (genie/config.py:33: per-head LayerNorm of q and k,
                                        # norm1 / norm2 = Identity), heads of 64 and of 32, and the dataclass defaults proper (+ muP scale)
                                        (512, 8, dict(qk_norm=True)), (256, 8, dict(qk_norm=True, qkv_bias=True)),
                                        (128, 2, dict(qk_norm=True, use_mup=True))])
def test_frame_passes_equal_full_forward_frames(d, heads, kw):
    """genie_frame_pass / genie_frames_pass on the fragment-order kernels against the full 16-frame forward of the same model
    (256x256-tile GEMMs, LDS-DMA attention kernels: validated against the oracle and the reference goldens elsewhere): logits of
    every decoded frame within f32 accumulation-order noise; one frame per pass, two frames per pass, 1 / 2 / 3 / 5 clips; with and
    without the Linear biases."""
    cfg, m = _model(d, heads, **kw)
    _lib = pkg("_lib")
    lib = _lib.load()
    c, w = m._weights()[:2]
    assert w.out_frame_w16 and w.layers_host[0].spatial.frame_w16 and w.layers_host[0].mlp_frame_w16   # the new path is the one that runs
    T, S = cfg.T, cfg.S
    V = cfg.factored_vocab_size * cfg.num_factored_vocabs
    for B in (1, 2, 3, 5):    # 256 .. 2,560 rows per pass: the register-direct kernels below 2,048 rows, the LDS-tiled ones from there
        ids = dev(pkg("synthetic").make_clips(B, cfg, seed=70 + B)).view(B, T, S)
        ids[:, 3, ::3] = cfg.image_vocab_size       # some mask tokens
        full = m.compute_logits(ids.view(B, T, 16, 16))          # (B, V, T, 16, 16)
        scale = full.abs().max().item()
        ws = m._workspace(B)
        nbytes = lib.genie_prefix_cache_bytes(c, B)
        st = torch.cuda.current_stream().cuda_stream

        def ref(t):
            return full[:, :, t].reshape(B, V, S).transpose(1, 2)   # (B, S, V)

        # one frame per pass
        cache = torch.zeros(nbytes // 4, dtype=torch.float32, device="cuda")
        for t in range(4):
            lg = torch.full((B, S, V), float("nan"), dtype=torch.float32, device="cuda")
            _lib.check(lib.genie_frame_pass(c, w, ids[:, t].contiguous().data_ptr(), B, t, cache.data_ptr(), nbytes, lg.data_ptr(),
                                            ws.data_ptr(), ws.numel(), st), "genie_frame_pass")
            err = (lg - ref(t)).abs()
            assert err.max().item() < 3e-5 * max(1.0, scale) and err.median().item() < 1e-5, (B, t, err.max().item(), err.median().item())
        # two frames per pass: frames (0, 1), then (2, 3) against the slots the first pass wrote; the cache equals the one-by-one cache
        cache2 = torch.zeros(nbytes // 4, dtype=torch.float32, device="cuda")
        for t0 in (0, 2):
            lg = torch.full((B, S, V), float("nan"), dtype=torch.float32, device="cuda")
            _lib.check(lib.genie_frames_pass(c, w, ids[:, t0:t0 + 2].contiguous().data_ptr(), B, t0, 2, cache2.data_ptr(), nbytes,
                                             lg.data_ptr(), ws.data_ptr(), ws.numel(), st), "genie_frames_pass")
            err = (lg - ref(t0 + 1)).abs()
            assert err.max().item() < 3e-5 * max(1.0, scale), (B, t0, err.max().item())
        L = cfg.num_layers
        a = cache.view(L, B, T, S, 3 * d)[:, :, :4]
        b = cache2.view(L, B, T, S, 3 * d)[:, :, :4]
        if (B * S >= 2048) == (2 * B * S >= 2048):
            assert torch.equal(a, b)   # same kernels: per-row arithmetic does not depend on how many frames share a pass
        else:                          # one frame on the register-direct kernels, two on the LDS-tiled ones: f32 summation order
            assert (a - b).abs().max().item() < 3e-5 * max(1.0, a.abs().max().item())
        assert (cache2.view(L, B, T, S, 3 * d)[:, :, 4:] == 0).all()


def test_prompt_in_one_frames_pass_fills_the_cache_like_single_frame_passes():
    """generate()'s prompt as ONE 8-frame genie_frames_pass (2,048 rows: the LDS-tiled kernels) leaves every layer's cache slots as 8
    one-frame passes (register-direct kernels) do, up to f32 summation order; later slots stay untouched."""
    cfg, m = _model(512, 8, layers=3)
    _lib = pkg("_lib")
    lib = _lib.load()
    c, w = m._weights()[:2]
    B, P, T, S, d, L = 1, 8, cfg.T, cfg.S, cfg.d_model, cfg.num_layers
    ids = dev(pkg("synthetic").make_clips(B, cfg, seed=55)).view(B, T, S)
    ws = m._workspace(B)
    nbytes = lib.genie_prefix_cache_bytes(c, B)
    st = torch.cuda.current_stream().cuda_stream
    ca = torch.zeros(nbytes // 4, dtype=torch.float32, device="cuda")
    cb = torch.zeros(nbytes // 4, dtype=torch.float32, device="cuda")
    _lib.check(lib.genie_frames_pass(c, w, ids[:, :P].contiguous().data_ptr(), B, 0, P, ca.data_ptr(), nbytes, 0, ws.data_ptr(), ws.numel(), st),
               "genie_frames_pass")
    for t in range(P):
        _lib.check(lib.genie_frame_pass(c, w, ids[:, t].contiguous().data_ptr(), B, t, cb.data_ptr(), nbytes, 0, ws.data_ptr(), ws.numel(), st),
                   "genie_frame_pass")
    a = ca.view(L, B, T, S, 3 * d)
    b = cb.view(L, B, T, S, 3 * d)
    assert torch.isfinite(a).all()
    assert (a[:, :, :P] - b[:, :, :P]).abs().max().item() < 3e-5 * max(1.0, b.abs().max().item())
    assert (a[:, :, P:] == 0).all()


def test_frames_pass_refuses_what_it_does_not_cover():
    _lib = pkg("_lib")
    lib = _lib.load()
    cfg, m = _model(128, 2, precision="exact")
    c, w = m._weights()[:2]
    B = 1
    ids = dev(pkg("synthetic").make_clips(B, cfg, seed=3)).view(B, cfg.T, cfg.S)
    ws = m._workspace(B)
    nbytes = lib.genie_prefix_cache_bytes(c, B)
    cache = torch.zeros(nbytes // 4, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.genie_frames_pass(c, w, ids[:, 0:2].contiguous().data_ptr(), B, 0, 2, cache.data_ptr(), nbytes, 0, ws.data_ptr(), ws.numel(), st)
    assert rc == _lib.E_UNSUPPORTED and b"frames_pass" in lib.genie_last_error()
    assert (cache == 0).all()
    assert lib.genie_frames_pass(c, w, ids.data_ptr(), B, 15, 2, cache.data_ptr(), nbytes, 0, ws.data_ptr(), ws.numel(), st) == _lib.E_ARG


@pytest.mark.parametrize("heads", [4, 8])
@pytest.mark.parametrize("steps", [2, 3])
def test_merged_commit_schedule_equals_plain_kv_cache_schedule(steps, heads):
    """generate_frames_cached with the commit pass of frame t carrying step 0 of frame t + 1 (genie_frames_pass, 2 frames) produces
    the frames of the schedule that runs them as two passes -- same per-row arithmetic, so exactly the same ids; and the loop as one
    library call (genie_generate_cached) produces what the Python-driven loop produces."""
    cfg, m = _model(256, heads, layers=3)
    G = pkg("generate")
    for B in (1, 2):
        ex = dev(pkg("synthetic").make_clips(B, cfg, seed=40 + B)).view(B, 16, 16, 16)
        noise = torch.rand(8, max(steps - 1, 1), B, cfg.S, device="cuda")
        for tf in (False, True):
            plain = G.generate_frames_cached(m, ex, 8, steps, 0.0, tf, noise=noise, merge_commit=False, host_loop=True)
            merged = G.generate_frames_cached(m, ex, 8, steps, 0.0, tf, noise=noise, merge_commit=True, host_loop=True)
            assert torch.equal(plain, merged)
            # ... and the whole loop as ONE library call (genie_generate_cached) gives the same frames again, merged or not
            for mc in (False, True):
                one = G.generate_frames_cached(m, ex, 8, steps, 0.0, tf, noise=noise, merge_commit=mc)
                assert torch.equal(one, plain)
    # greedy unmasking needs no draws; a temperature > 0 run only has to be well-formed (its draws are its own)
    ex = dev(pkg("synthetic").make_clips(1, cfg, seed=44)).view(1, 16, 16, 16)
    a = G.generate_frames_cached(m, ex, 8, steps, 0.0, False, unmask_mode="greedy")
    b = G.generate_frames_cached(m, ex, 8, steps, 0.0, False, unmask_mode="greedy", host_loop=True)
    assert torch.equal(a, b)
    hot = G.generate_frames_cached(m, ex, 8, steps, 0.7, False)
    assert hot.shape == (1, 24, 16, 16) and torch.equal(hot[:, :8], ex[:, :8]) and torch.equal(hot[:, 16:], ex[:, 8:])
    assert int(hot[:, 8:16].min()) >= 0 and int(hot[:, 8:16].max()) < cfg.image_vocab_size


@pytest.mark.parametrize("M,N,K,mode", [(256, 1536, 512, 1), (512, 512, 512, 1), (96, 192, 128, 1), (1024, 2048, 512, 1), (2048, 1536, 512, 2),
                                        (1280, 512, 2048, 2), (4096, 512, 512, 2), (384, 128, 256, 2), (4096, 1536, 512, 2),
                                        (4352, 1024, 256, 2)])
def test_frame_linear_vs_f64(M, N, K, mode):
    """genie_frame_linear (nn.Linear on fragment-ordered split operands; st_transformer.py:16-25, attention.py:27-29) against the
    f64 product of the SAME split operands (hi + lo / 2048 of genie_pack_split_f16): the kernels' only error is the dropped lo.lo term
    and f32 accumulation order.  mode 1 = register-direct kernel, 2 = LDS-tiled kernel; ragged chip fills, K = 128 .. 2048."""
    _lib = pkg("_lib")
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda", generator=g) * 2.0
    W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)

    def packs(t):
        rm = torch.empty((2,) + tuple(t.shape), dtype=torch.float16, device="cuda")
        fr = torch.empty(2 * t.numel(), dtype=torch.float16, device="cuda")
        _lib.check(lib.genie_pack_split_f16(t.data_ptr(), rm.data_ptr(), t.numel(), st), "pack_split")
        _lib.check(lib.genie_pack_frame_w16(t.data_ptr(), fr.data_ptr(), t.shape[0], t.shape[1], st), "pack_frame")
        return rm[0].double() + rm[1].double() / 2048.0, fr

    x64, x_fr = packs(x)
    w64, w_fr = packs(W)
    y = torch.full((M, N), float("nan"), device="cuda")
    _lib.check(lib.genie_frame_linear(x_fr.data_ptr(), w_fr.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, mode, st), "frame_linear")
    ref = x64 @ w64.T + b.double()
    err = (y.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    print(f"frame_linear M={M} N={N} K={K} mode={mode}: max err {err:.2e} of {scale:.2f}")
    assert err < 5e-6 * scale * max(1.0, (K / 512) ** 0.5)
    # without a bias
    _lib.check(lib.genie_frame_linear(x_fr.data_ptr(), w_fr.data_ptr(), 0, y.data_ptr(), M, N, K, mode, st), "frame_linear")
    assert (y.double() - (ref - b.double())).abs().max().item() < 5e-6 * scale * max(1.0, (K / 512) ** 0.5)


@pytest.mark.parametrize("B,n_prompt,n_new,steps", [(1, 1, 15, 1), (3, 2, 14, 3), (2, 5, 11, 2)])
def test_module_generate_on_kv_cache_equals_full_forward_schedule(B, n_prompt, n_new, steps):
    """STMaskGIT.generate (st_mask_git.py:65-113) on the temporal KV cache (genie_generate_cached: one library call) against the same call
    on the reference's schedule (a full forward over the canvas per MaskGIT step): same frames up to f32 accumulation order, the same
    step-0 logits; odd batch, one prompt frame, one MaskGIT step.  (The KV-cache form also takes a canvas shorter than T; the reference's
    own schedule cannot: its positional table does not broadcast, st_mask_git.py:261 -- here a RuntimeError instead of a fault.)"""
    cfg, m = _model(128, 2, layers=2)
    S = cfg.S
    ids = dev(pkg("synthetic").make_clips(B, cfg, seed=60 + B)).view(B, cfg.T, S)[:, :n_prompt].reshape(B, n_prompt * S)
    noise = torch.rand(n_new, max(steps - 1, 1), B, S, device="cuda")
    a, la = m.generate(ids, None, max_new_tokens=n_new * S, return_logits=True, maskgit_steps=steps, temperature=0.0, noise=noise, kv_cache=True)
    b, lb = m.generate(ids, None, max_new_tokens=n_new * S, return_logits=True, maskgit_steps=steps, temperature=0.0, noise=noise, kv_cache=False)
    assert a.shape == b.shape == (B, (n_prompt + n_new) * S) and la.shape == lb.shape == (B, 512, 2, n_new, 16, 16)
    assert torch.equal(a[:, :n_prompt * S], ids)
    first = a[:, n_prompt * S:(n_prompt + 1) * S] == b[:, n_prompt * S:(n_prompt + 1) * S]
    assert first.float().mean().item() > 0.995          # the first new frame sees identical context in both schedules
    assert (la[:, :, :, 0] - lb[:, :, :, 0]).abs().max().item() < 3e-5 * max(1.0, lb.abs().max().item())
    assert (a == b).float().mean().item() > 0.97         # later frames inherit any flipped argmax of the earlier ones
    short = m.generate(ids, None, max_new_tokens=S, maskgit_steps=steps, noise=noise[:1])      # a canvas shorter than T: KV cache only
    assert short.shape == (B, (n_prompt + 1) * S) and torch.equal(short, a[:, :(n_prompt + 1) * S])
    if n_prompt + 1 < cfg.T:
        with pytest.raises(RuntimeError):
            m.generate(ids, None, max_new_tokens=S, maskgit_steps=steps, kv_cache=False)


@pytest.mark.parametrize("precision", ["f16x3", "exact"])
def test_generate_when_the_prompt_fills_all_but_one_frame_of_a_short_model(precision):
    """T = 2, one prompt frame: max(P, 2) == T, so the loop scratch of genie_generate_cached does not fit behind the passes' workspace inside
    genie_workspace_bytes(cfg, B) -- the module sizes its buffer with genie_generate_workspace_bytes and generate() (KV cache = the default)
    works; a caller that passes the smaller buffer gets GENIE_E_ARG, nothing enqueued."""
    cfg = pkg("config").GenieConfig(num_layers=2, num_heads=2, d_model=128, T=2, S=256, num_factored_vocabs=2, qk_norm=False, use_mup=False)
    sd = pkg("synthetic").make_state_dict(cfg, seed=5, law="conditioned")
    m = pkg("st_mask_git").STMaskGIT(cfg, precision=precision).load_numpy_state_dict(sd).to("cuda")
    _lib = pkg("_lib")
    lib = _lib.load()
    c = m._weights()[0]
    B, S = 2, cfg.S
    assert lib.genie_generate_workspace_bytes(c, B, 1) > lib.genie_workspace_bytes(c, B)
    ids = dev(pkg("synthetic").make_clips(B, cfg, seed=6)).view(B, cfg.T, S)[:, :1].reshape(B, S)
    noise = torch.rand(1, 1, B, S, device="cuda")
    a = m.generate(ids, None, max_new_tokens=S, maskgit_steps=2, temperature=0.0, noise=noise, kv_cache=True)
    b = m.generate(ids, None, max_new_tokens=S, maskgit_steps=2, temperature=0.0, noise=noise, kv_cache=False)
    assert a.shape == b.shape == (B, 2 * S) and torch.equal(a[:, :S], ids)
    assert (a == b).float().mean().item() > 0.995
