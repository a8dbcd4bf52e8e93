"""GPU (-m gpu): generate.py harness, tokenizer ends (bits / u8 rescale / sign->ids kernels) and the on-device
MAGVIT2 decode/encode against the reference's golden outputs."""
import ast

import numpy as np
import pytest

from conftest import GOLDEN, pkg
from oracle import genie_oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


@pytest.fixture(scope="module")
def harness():
    return np.load(f"{GOLDEN}/harness.npz")


@pytest.fixture(scope="module")
def magvit():
    return np.load(f"{GOLDEN}/magvit_small.npz")


@pytest.mark.parametrize("precision", ["exact", "f16x3"])
def test_generate_harness_matches_reference(tmp_path, harness, precision):
    cfg = pkg("config").GenieConfig(**ast.literal_eval(str(harness["cfg"])))
    sd = pkg("synthetic").make_state_dict(cfg, seed=int(harness["weight_seed"]))
    m = pkg("st_mask_git").STMaskGIT(cfg, precision=precision).load_numpy_state_dict(sd).to("cuda")
    G = pkg("generate")
    ex = dev(harness["gen_example"])
    for tf, key in [(False, "gen_ar"), (True, "gen_tf")]:
        out = G.generate_frames(m, ex, num_prompt_frames=2, maskgit_steps=2, temperature=0.0, teacher_force_time=tf,
                                noise=dev(harness[key + "_noise"]))
        assert np.array_equal(out.cpu().numpy(), harness[key + "_outputs"])  # [prompt | generated | ground truth]
    # on-disk layout round-trips through the dataset reader (generate.py:105-116, visualize.py:153-165)
    D = pkg("data")
    meta = G.write_outputs(out, tmp_path, {"s": 4, "vocab_size": 262144, "hz": 30, "token_dtype": "uint32"},
                           {"window_size": 4, "num_prompt_frames": 2, "maskgit_steps": 2})
    assert meta["num_images"] == 6 and meta["h"] == 4 and meta["t"] == 4
    ds = D.RawTokenDataset(tmp_path, 1, filter_interrupts=False)
    assert np.array_equal(np.asarray(ds.data), out.cpu().numpy().reshape(6, 4, 4))


def test_tokenizer_byte_and_bit_kernels(magvit):
    mv = pkg("magvit2")
    z = mv.bits_from_tokens(dev(magvit["bits_ids"]))
    assert np.array_equal(z.cpu().numpy(), magvit["bits_z"])
    assert np.array_equal(mv.tokens_from_bits(dev(magvit["bits_z"])).cpu().numpy(), magvit["bits_ids"])
    r = dev(magvit["rescale_in_bf16_as_f32"]).to(torch.bfloat16)
    assert np.array_equal(mv.rescale_magvit_output(r).cpu().numpy(), magvit["rescale_out"])  # bit-exact bytes
    y16 = dev(magvit["dec_out_bf16_as_f32"]).to(torch.bfloat16)
    assert np.array_equal(mv.rescale_magvit_output(y16).cpu().numpy(), magvit["dec_u8_bf16"])
    assert np.array_equal(mv.rescale_magvit_output(dev(magvit["dec_out_f32"])).cpu().numpy(), magvit["dec_u8_f32"])
    # edge values: clamp both sides, truncation (not rounding)
    e = torch.tensor([-3.0, -1.0, -0.9961, 0.0, 0.999, 1.0, 5.0], device="cuda")
    assert mv.rescale_magvit_output(e).cpu().tolist() == [0, 0, 0, 127, 254, 255, 255]


def test_single_backend(magvit):
    """The product package executes MAGVIT2 only on the hand-written conv stack: the nn.Modules are parameter holders and an
    uncovered geometry (the 32/64-wide toy config) is an error, not a silent torch/MIOpen fallback."""
    mv = pkg("magvit2")
    small = ast.literal_eval(str(magvit["cfg"]))
    m = mv.VQModel(mv.VQConfig(**small)).to("cuda")
    with pytest.raises(RuntimeError, match="parameter holder"):
        m.decoder(torch.zeros(1, 18, 4, 4, device="cuda"))
    with pytest.raises(ValueError, match="multiples of 64"):
        m.decode_tokens(dev(magvit["dec_tokens"]))
    with pytest.raises(ValueError, match="multiples of 64"):
        mv.decode_latents_wrapper(batch_size=1, model=m)


def test_hip_decoder_conv_stack(magvit):
    """Hand-written conv path (implicit-GEMM bf16 MFMA convs, fused GroupNorm+swish, depth-to-space epilogue) against
    the reference decoder's f32 output on a mid-size config with the shipped width classes (256/128)."""
    mv = pkg("magvit2")
    z = np.load(f"{GOLDEN}/magvit_mid.npz")
    cfg = ast.literal_eval(str(z["cfg"]))
    m = mv.VQModel(mv.VQConfig(**cfg))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in mv.make_vq_state_dict(m, int(z["weight_seed"])).items()})
    m = m.to("cuda")
    hd = mv.HipDecoder(m.decoder)
    tok = dev(z["dec_tokens"])
    y = hd.decode_tokens(tok, return_float=True).cpu().numpy()
    ref = z["dec_out_f32"]
    assert y.shape == ref.shape
    err = np.abs(y - ref)
    scale = np.abs(ref).max()
    # bf16 activations + bf16 weights through ~14 conv layers: a few 1e-2 of the output range
    assert np.median(err) < 0.01 * scale and err.max() < 0.08 * scale, (np.median(err), err.max(), scale)
    u8 = hd.decode_tokens(tok).cpu().numpy().astype(np.int32)
    d = np.abs(u8 - z["dec_u8_f32"].astype(np.int32))
    assert np.median(d) <= 2 and d.mean() < 3
    # the torch formulation (oracle, MIOpen bf16 convs on this GPU) of the same parameters sits at the same distance
    from oracle import magvit2_oracle as MO
    import copy
    mb = copy.deepcopy(m).to(torch.bfloat16)
    ym = MO.decoder_forward(mb.decoder, mv.bits_from_tokens(tok).to(torch.bfloat16)).float().cpu().numpy()
    assert np.abs(ym - ref).max() < 0.08 * scale
    # VQModel.decode_tokens and the reference-shaped wrapper are this same stack, device-resident
    assert torch.equal(m.decode_tokens(tok), hd.decode_tokens(tok))
    frames = mv.decode_latents_wrapper(batch_size=2, model=m)(tok)
    assert frames.is_cuda and frames.dtype == torch.uint8 and torch.equal(frames, hd.decode_tokens(tok))


def test_hip_encoder_conv_stack(magvit):
    """Hand-written encoder (padded conv_in, strided implicit-GEMM downsample, 1x1 conv_out GEMM, sign-bit packing) against
    the reference encoder's f32 code: every bit whose pre-quantisation value is not within bf16 noise of zero agrees."""
    mv = pkg("magvit2")
    z = np.load(f"{GOLDEN}/magvit_mid.npz")
    cfg = ast.literal_eval(str(z["cfg"]))
    m = mv.VQModel(mv.VQConfig(**cfg))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in mv.make_vq_state_dict(m, int(z["weight_seed"])).items()})
    m = m.to("cuda")
    he = mv.HipEncoder(m.encoder)
    ids = he.encode_tokens(dev(z["enc_frames"])).cpu().numpy()
    h = z["enc_h"]  # (n, 18, h, w) f32 reference code
    assert ids.shape == (h.shape[0], h.shape[2], h.shape[3])
    bits = (ids[:, None] >> np.arange(18)[None, :, None, None]) & 1
    ref_bits = (h > 0).astype(np.int64)
    scale = np.abs(h).std()
    robust = np.abs(h) > 0.05 * scale
    assert np.array_equal(bits[robust], ref_bits[robust])
    assert (bits == ref_bits).mean() > 0.97
    # encode -> decode round trip stays on the device
    hd = mv.HipDecoder(m.decoder)
    rgb = hd.decode_tokens(torch.from_numpy(ids).cuda())
    assert rgb.is_cuda and rgb.dtype == torch.uint8 and tuple(rgb.shape) == (3, 3, 32, 32)


@pytest.mark.parametrize("n,H,W,cin,cout,d2s,stride", [(2, 16, 16, 64, 512, False, 1), (1, 32, 32, 128, 1024, True, 1),
                                                       (2, 16, 32, 128, 256, False, 2), (1, 64, 64, 64, 128, False, 1)])
def test_conv_fused_groupnorm_statistics(n, H, W, cin, cout, d2s, stride):
    """genie_conv3x3_gn_bf16 + genie_group_norm_swish_fused_bf16 (GroupNorm statistics from the conv epilogue) against
    (a) the plain conv + separate statistics pass (same conv bytes; normalised output within one bf16 ulp) and
    (b) torch GroupNorm in f64 on the stored conv output (improved_model.py:24-34: GroupNorm(32, eps 1e-6) + x*sigmoid(x))."""
    _lib = pkg("_lib")
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(n * 1000 + cout)
    Hi, Wi = H * stride, W * stride
    x = torch.randn(n, Hi, Wi, cin, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / (3 * cin ** 0.5)).contiguous()
    b = torch.randn(cout, device="cuda", generator=g)
    wp = torch.empty(cout, 9, cin, dtype=torch.bfloat16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.genie_pack_conv_weight(w.data_ptr(), wp.data_ptr(), cout, cin, 9, st), "pack")
    zero = torch.zeros(64, dtype=torch.bfloat16, device="cuda")
    C = cout // 4 if d2s else cout
    oshape = (n, 2 * H, 2 * W, C) if d2s else (n, H, W, C)
    res = None if (d2s or stride == 2) else torch.randn(oshape, device="cuda", generator=g).to(torch.bfloat16)
    rp = 0 if res is None else res.data_ptr()
    y0, y1 = torch.empty(oshape, dtype=torch.bfloat16, device="cuda"), torch.empty(oshape, dtype=torch.bfloat16, device="cuda")
    if stride == 2:
        _lib.check(lib.genie_conv3x3_s2_bf16(x.data_ptr(), wp.data_ptr(), b.data_ptr(), y0.data_ptr(), zero.data_ptr(), n, H, W, cin,
                                             cout, st), "conv_s2")
    else:
        _lib.check(lib.genie_conv3x3_bf16(x.data_ptr(), wp.data_ptr(), b.data_ptr(), rp, y0.data_ptr(), zero.data_ptr(), n, H, W, cin,
                                          cout, int(d2s), st), "conv")
    part = torch.empty(lib.genie_conv_gn_part_floats(n, H, W, cout), dtype=torch.float32, device="cuda")
    _lib.check(lib.genie_conv3x3_gn_bf16(x.data_ptr(), wp.data_ptr(), b.data_ptr(), rp, y1.data_ptr(), zero.data_ptr(), n, H, W, cin,
                                         cout, int(d2s), stride, part.data_ptr(), 32, st), "conv_gn")
    assert torch.equal(y0, y1)                                   # the convolution itself is unchanged
    gamma = (1 + 0.1 * torch.randn(C, device="cuda", generator=g)).contiguous()
    beta = (0.1 * torch.randn(C, device="cuda", generator=g)).contiguous()
    HWo = oshape[1] * oshape[2]
    z0, z1 = torch.empty_like(y0), torch.empty_like(y0)
    ws = torch.empty(lib.genie_group_norm_scratch_floats(n, HWo, 32), dtype=torch.float32, device="cuda")
    _lib.check(lib.genie_group_norm_swish_bf16(y0.data_ptr(), gamma.data_ptr(), beta.data_ptr(), z0.data_ptr(), ws.data_ptr(), n, HWo,
                                               C, 32, 1e-6, 1, st), "gn")
    ws1 = torch.empty(n * 64, dtype=torch.float32, device="cuda")
    _lib.check(lib.genie_group_norm_swish_fused_bf16(y1.data_ptr(), gamma.data_ptr(), beta.data_ptr(), z1.data_ptr(), part.data_ptr(),
                                                     ws1.data_ptr(), n, H, W, cout, int(d2s), 32, 1e-6, 1, st), "gn_fused")
    stats_sep, stats_fused = ws[:n * 64].view(n, 32, 2), ws1.view(n, 32, 2)
    assert (stats_sep - stats_fused).abs().max().item() < 2e-5 * max(1.0, stats_sep.abs().max().item())
    yf = y1.double().permute(0, 3, 1, 2)
    ref = torch.nn.functional.group_norm(yf, 32, gamma.double(), beta.double(), eps=1e-6)
    ref = (ref * torch.sigmoid(ref)).permute(0, 2, 3, 1)
    err = (z1.double() - ref).abs()
    assert (err <= 2.0 ** -7 * ref.abs() + 1e-3).all(), err.max().item()   # one bf16 rounding of the result
    assert ((z0.float() - z1.float()).abs() <= 2.0 ** -7 * z0.float().abs() + 1e-6).all()
    # order-fixed reductions: same bytes on a second call
    part2 = torch.empty_like(part)
    _lib.check(lib.genie_conv3x3_gn_bf16(x.data_ptr(), wp.data_ptr(), b.data_ptr(), rp, y1.data_ptr(), zero.data_ptr(), n, H, W, cin,
                                         cout, int(d2s), stride, part2.data_ptr(), 32, st), "conv_gn")
    used = part.view(-1, 64)[:, : 2 * (128 // (C // 32))]
    assert torch.equal(used, part2.view(-1, 64)[:, : 2 * (128 // (C // 32))])


@pytest.mark.parametrize("n,H,W,cin,cout,d2s,res", [(3, 24, 40, 64, 128, False, True),    # non power-of-two sides, ragged last tile
                                                    (2, 16, 16, 128, 8, False, False),    # narrow head (32-column form)
                                                    (1, 20, 12, 64, 256, True, False),    # depth-to-space, non power-of-two
                                                    (2, 8, 512, 64, 128, False, True),    # rows wider than a 256-pixel tile (halo pixels used)
                                                    (5, 16, 16, 192, 384, False, False)])  # 3 channel chunks, 3 column tiles
def test_conv3x3_slab_against_f64(n, H, W, cin, cout, d2s, res):
    """genie_conv3x3_bf16 (stride 1: one LDS slab per vertical tap serving the three horizontal taps, persistent tiles) against
    torch conv2d in f64 on the same bf16 operands (improved_model.py: Conv2d(k=3, padding=1); Upsampler's depth-to-space
    :185-237): borders, image seams inside a tile, ragged tiles, narrow heads."""
    _lib = pkg("_lib")
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(n * 100 + cout + W)
    x = torch.randn(n, H, W, cin, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / (3 * cin ** 0.5)).contiguous()
    b = torch.randn(cout, device="cuda", generator=g)
    wp = torch.empty(cout, 9, cin, dtype=torch.bfloat16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.genie_pack_conv_weight(w.data_ptr(), wp.data_ptr(), cout, cin, 9, st), "pack")
    zero = torch.zeros(64, dtype=torch.bfloat16, device="cuda")
    C = cout // 4 if d2s else cout
    oshape = (n, 2 * H, 2 * W, C) if d2s else (n, H, W, C)
    r = torch.randn(oshape, device="cuda", generator=g).to(torch.bfloat16) if res else None
    y = torch.full(oshape, float("nan"), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.genie_conv3x3_bf16(x.data_ptr(), wp.data_ptr(), b.data_ptr(), 0 if r is None else r.data_ptr(), y.data_ptr(),
                                      zero.data_ptr(), n, H, W, cin, cout, int(d2s), st), "conv")
    wq = wp.view(cout, 3, 3, cin).permute(0, 3, 1, 2).double()            # the packed (bf16-rounded) weights
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), wq, b.double(), padding=1)
    if d2s:  # DCR depth-to-space: channel (i*2 + j)*C + c -> pixel (2y+i, 2x+j)
        ref = ref.view(n, 2, 2, C, H, W).permute(0, 3, 4, 1, 5, 2).reshape(n, C, 2 * H, 2 * W)
    ref = ref.permute(0, 2, 3, 1)
    if r is not None:
        ref = ref + r.double()
    err = (y.double() - ref).abs()
    assert torch.isfinite(y.float()).all()
    assert (err <= 2.0 ** -7 * ref.abs() + 2e-3).all(), err.max().item()
