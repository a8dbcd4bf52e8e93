"""CPU oracle for the GENIE forward / MaskGIT sampling path.  TEST INFRASTRUCTURE ONLY.

This file is a clean-room NumPy restatement of what the reference computes on the hot path
(SURVEY.md section 8a / Appendix A).  It is the checker the HIP path is compared against; it is
never imported by the product package ``1xgpt_amd`` (only by tests/, __graft_entry__.smoke() and the
``cpu_baseline`` leg of bench.py).

Pinning: the reference holds no golden vectors for this path (its only test is test_attention.py,
which needs CUDA+xformers).  The oracle is therefore pinned against outputs of the reference itself,
imported in the build container by tools/make_goldens.py and committed under tests/golden/
(tests/test_oracle_golden.py checks every one of them).  Exception: the muP readout
(``use_mup=True``) depends on the un-vendored ``mup`` package (janEbert fork @fsdp-fix,
requirements.txt:11); its formula ``Linear(output_mult * x / width_mult)`` is restated from the
comment at genie/st_mask_git.py:317-323 -- **parity unpinned** for that branch.

All ``file:line`` citations are relative to the reference tree.
"""
import math

import numpy as np
from scipy.special import erf as _erf

MASK_NEG = None  # set per dtype: -finfo.max (genie/attention.py:52)


# ----------------------------------------------------------------------------------------------
# low-precision emulation used to pin the bf16 ("fast") HIP path
# ----------------------------------------------------------------------------------------------
def round_bf16(a: np.ndarray) -> np.ndarray:
    """Round float32 to the nearest-even bfloat16 and return it as float32."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)


# The GELU of the HIP bf16 contract (csrc/common.hpp gelu_erf_poly2, used by EVERY bf16 kernel whose hidden leaves only as bf16):
# gelu(z) = z * (1/2 + zc * P(zc^2)), zc = clamp(z, -4.25, 4.25), P of degree 8 -- |Phi error| <= 1.3e-5 against the reference's erf form
# (st_transformer.py:18); coefficients = tools/fit_gelu_poly.py COEF (tests/test_gelu_poly.py holds all three copies together).
GELU_POLY_Z = 4.25
GELU_POLY_COEF = (3.989023268e-01, -6.634449214e-02, 9.815969504e-03, -1.108560245e-03, 9.341857367e-05, -5.626413895e-06,
                  2.255418963e-07, -5.327728037e-09, 5.564818051e-11)


def gelu_poly(z):
    """What the bf16 kernels compute, in f32 like the kernel's Horner chain (numpy's mul + add where the kernel has one fused
    multiply-add: the same to ~1 ulp per step, which only matters where the bf16 rounding of the result sits on a boundary)."""
    z = np.asarray(z, dtype=np.float32)
    zc = np.clip(z, np.float32(-GELU_POLY_Z), np.float32(GELU_POLY_Z))
    s = zc * zc
    p = np.full_like(s, np.float32(GELU_POLY_COEF[-1]))
    for c in GELU_POLY_COEF[-2::-1]:
        p = p * s + np.float32(c)
    return z * (zc * p + np.float32(0.5))


class Numerics:
    """dtype + optional operand rounding applied at every GEMM/attention-matmul input (+ the GELU form of the contract)."""

    def __init__(self, dtype=np.float32, gemm_in=None, attn_in=False, temporal_qkv=None, gelu=None):
        self.gelu = gelu            # None: the reference's erf form (gelu_erf); the HIP bf16 contract: gelu_poly
        self.dtype = np.dtype(dtype)
        self.gemm_in = gemm_in  # rounding of every nn.Linear operand (e.g. round_bf16)
        self.attn_in = attn_in  # True: the attention matmuls also take rounded operands (bf16 contract);
        #                         False: attention is f32 in the reference's order (exact / f16x3 contract)
        self.temporal_qkv = temporal_qkv  # rounding of the STORED temporal qkv (HIP bf16 keeps it, and the KV cache, in bf16)

    def r(self, a):
        return a if self.gemm_in is None else self.gemm_in(a).astype(self.dtype, copy=False)

    def ra(self, a):
        return self.r(a) if self.attn_in else a


F32 = Numerics(np.float32)
F64 = Numerics(np.float64)
# HIP bf16 contract: Linear operands bf16, the temporal qkv buffer / KV cache stored in bf16, attention arithmetic f32
# (+ the polynomial GELU in front of the bf16 rounding of the MLP hidden)
BF16_MFMA = Numerics(np.float32, round_bf16, temporal_qkv=round_bf16, gelu=gelu_poly)
BF16_ALL = Numerics(np.float32, round_bf16, attn_in=True, temporal_qkv=round_bf16, gelu=gelu_poly)  # attention matmuls on bf16 operands as well


# ----------------------------------------------------------------------------------------------
# a2: token factorisation + embedding  (genie/factorization_utils.py:29-68, genie/st_mask_git.py:257-261)
# ----------------------------------------------------------------------------------------------
def factorize_token_ids(ids, num_factored_vocabs=2, factored_vocab_size=512):
    """(...,) -> (..., num_factored_vocabs); factor j = (id // 512**j) % 512  (factorization_utils.py:55-68)."""
    ids = np.asarray(ids, dtype=np.int64)
    powers = factored_vocab_size ** np.arange(num_factored_vocabs, dtype=np.int64)
    return (ids[..., None] // powers) % factored_vocab_size


def unfactorize_token_ids(factored, num_factored_vocabs=2, factored_vocab_size=512):
    powers = factored_vocab_size ** np.arange(num_factored_vocabs, dtype=np.int64)
    return (np.asarray(factored, dtype=np.int64) * powers).sum(-1)


def embed(ids_BTS, sd, cfg, nm=F32):
    """x = (id==MASK ? mask_embed : sum_j E_j[factor_j(id)]) + pos  -> (B,T,S,d)."""
    dt = nm.dtype
    ids = np.asarray(ids_BTS, dtype=np.int64)
    is_mask = ids == cfg.image_vocab_size
    safe = np.where(is_mask, 0, ids)
    fac = factorize_token_ids(safe, cfg.num_factored_vocabs, cfg.factored_vocab_size)
    e = None
    for j in range(cfg.num_factored_vocabs):
        ej = sd[f"token_embed.factored_embeds.{j}.weight"].astype(dt)[fac[..., j]]
        e = ej if e is None else e + ej
    e = np.where(is_mask[..., None], sd["token_embed.mask_token_embed"].astype(dt)[0], e)
    return e + sd["pos_embed_TSC"].astype(dt)


# ----------------------------------------------------------------------------------------------
# a4: LayerNorm (biased variance, eps 1e-5, affine)  (genie/st_transformer.py:44,67; attention.py:34)
# ----------------------------------------------------------------------------------------------
def layer_norm(x, gamma, beta, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdims=True)
    return xc / np.sqrt(var + x.dtype.type(eps)) * gamma.astype(x.dtype) + beta.astype(x.dtype)


def gelu_erf(z):
    """nn.GELU() default = exact erf form (genie/st_transformer.py:18)."""
    return z * (z.dtype.type(0.5) * (z.dtype.type(1.0) + _erf(z * z.dtype.type(1.0 / math.sqrt(2.0)))))


def _softmax_last(a):
    m = a.max(-1, keepdims=True)
    e = np.exp(a - m)
    return e / e.sum(-1, keepdims=True)


# ----------------------------------------------------------------------------------------------
# a5-a9: SelfAttention.forward (genie/attention.py:36-61, the pure-torch back-end that
# test_attention.py:18 pins equal to the xformers one)
# ----------------------------------------------------------------------------------------------
def self_attention(x_BNC, sd, prefix, cfg, causal, nm=F32, chunk=64):
    dt = nm.dtype
    Bn, N, C = x_BNC.shape
    H, Dh = cfg.num_heads, cfg.head_dim
    Wqkv = nm.r(sd[prefix + "qkv.weight"].astype(dt))
    Wp = nm.r(sd[prefix + "proj.weight"].astype(dt))
    scale = dt.type(cfg.attn_scale)
    out = np.empty((Bn, N, C), dtype=dt)
    for s0 in range(0, Bn, chunk):
        x = x_BNC[s0:s0 + chunk]
        b = x.shape[0]
        qkv = nm.r(x) @ Wqkv.T  # (b,N,3C), no bias unless cfg.qkv_bias
        if cfg.qkv_bias:
            qkv = qkv + sd[prefix + "qkv.bias"].astype(dt)
        qkv = nm.ra(qkv)  # bf16 contract: the qkv buffer itself is stored rounded (f32 otherwise)
        if causal and nm.temporal_qkv is not None:
            qkv = nm.temporal_qkv(qkv).astype(dt, copy=False)
        qkv = qkv.reshape(b, N, 3, H, Dh).transpose(2, 0, 3, 1, 4)  # (3,b,H,N,Dh)  attention.py:38
        q, k, v = qkv[0], qkv[1], qkv[2]
        if cfg.qk_norm:  # one shared affine for q and k  (attention.py:42-47)
            g, bb = sd[prefix + "norm.weight"], sd[prefix + "norm.bias"]
            q = layer_norm(q, g, bb)
            k = layer_norm(k, g, bb)
        if not nm.attn_in:
            q = q * scale  # attention.py:48
            attn = q @ k.transpose(0, 1, 3, 2)  # (b,H,N,N)
        else:
            # 16-bit matrix-core contract of the HIP fast path: q, k are rounded operands and the scale
            # multiplies the f32 scores (same math, different rounding point)
            attn = (nm.r(q) @ nm.r(k).transpose(0, 1, 3, 2)) * scale
        if causal:  # attention.py:51-55
            mask = ~np.tril(np.ones((N, N), dtype=bool))
            attn = np.where(mask, -np.finfo(dt).max, attn)
        if not nm.attn_in:
            attn = _softmax_last(attn)
            o = attn @ v
        else:
            # ... and the un-normalised probabilities are the rounded operand of P.V; the f32 row sum divides after
            e = np.exp(attn - attn.max(-1, keepdims=True))
            o = (nm.r(e) @ nm.r(v)) / e.sum(-1, keepdims=True)
        o = o.transpose(0, 2, 1, 3).reshape(b, N, C)  # attention.py:59
        o = nm.r(o) @ Wp.T
        if cfg.proj_bias:
            o = o + sd[prefix + "proj.bias"].astype(dt)
        out[s0:s0 + chunk] = o
    return out


def mlp(x, sd, prefix, cfg, nm=F32):
    """fc2(gelu(fc1(x)))  (genie/st_transformer.py:16-25)."""
    dt = nm.dtype
    h = nm.r(x) @ nm.r(sd[prefix + "fc1.weight"].astype(dt)).T
    if cfg.mlp_bias:
        h = h + sd[prefix + "fc1.bias"].astype(dt)
    h = nm.gelu(h) if nm.gelu is not None else gelu_erf(h)
    o = nm.r(h) @ nm.r(sd[prefix + "fc2.weight"].astype(dt)).T
    if cfg.mlp_bias:
        o = o + sd[prefix + "fc2.bias"].astype(dt)
    return o


# ----------------------------------------------------------------------------------------------
# a3: STBlock / STTransformerDecoder (genie/st_transformer.py:70-83, 115-120)
# ----------------------------------------------------------------------------------------------
def st_block(x_BTSC, sd, i, cfg, nm=F32):
    B, T, S, C = x_BTSC.shape
    p = f"decoder.layers.{i}."
    # spatial: sequences = (b,t), over S  (st_transformer.py:73-74)
    x = x_BTSC.reshape(B * T, S, C)
    u = x if cfg.qk_norm else layer_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    x = x + self_attention(u, sd, p + "spatial_attn.", cfg, False, nm)
    # temporal: sequences = (b,s), over T, causal, NO pre-norm  (st_transformer.py:77-78)
    x = x.reshape(B, T, S, C).transpose(0, 2, 1, 3).reshape(B * S, T, C)
    x = x + self_attention(x, sd, p + "temporal_attn.", cfg, True, nm, chunk=4096)
    # MLP  (st_transformer.py:81)
    u = x if cfg.qk_norm else layer_norm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    x = x + mlp(u, sd, p + "mlp.", cfg, nm)
    return x.reshape(B, S, T, C).transpose(0, 2, 1, 3)


def decoder_forward(x_BTSC, sd, cfg, nm=F32):
    x = x_BTSC
    for i in range(cfg.num_layers):
        x = st_block(x, sd, i, cfg, nm)
    return np.ascontiguousarray(x)


# ----------------------------------------------------------------------------------------------
# a11: compute_logits (genie/st_mask_git.py:255-265) -> (B, V, T, H, W), V = [vocab0 | vocab1]
# ----------------------------------------------------------------------------------------------
def hidden_states(ids_BTHW, sd, cfg, nm=F32):
    ids = np.asarray(ids_BTHW, dtype=np.int64)
    B, T = ids.shape[:2]
    return decoder_forward(embed(ids.reshape(B, T, -1), sd, cfg, nm), sd, cfg, nm)


def readout(x_BTSC, sd, cfg, nm=F32):
    """token-major logits (B,T,S,V); muP: Linear(output_mult*x/width_mult) (st_mask_git.py:316-323)."""
    dt = nm.dtype
    x = x_BTSC
    if nm.gemm_in is None:
        if cfg.use_mup:
            x = x * dt.type(cfg.readout_mult)
        return x @ sd["out_x_proj.weight"].astype(dt).T + sd["out_x_proj.bias"].astype(dt)
    # 16-bit contract: the muP factor scales the f32 accumulator
    return (nm.r(x) @ nm.r(sd["out_x_proj.weight"].astype(dt)).T) * dt.type(cfg.readout_mult) \
        + sd["out_x_proj.bias"].astype(dt)


def compute_logits(ids_BTHW, sd, cfg, nm=F32):
    ids = np.asarray(ids_BTHW)
    B, T, H, W = ids.shape
    lg = readout(hidden_states(ids, sd, cfg, nm), sd, cfg, nm)  # (B,T,S,V)
    return np.ascontiguousarray(lg.reshape(B, T, H, W, -1).transpose(0, 4, 1, 2, 3))  # B C T H W (:264)


# ----------------------------------------------------------------------------------------------
# a12: losses (genie/st_mask_git.py:231-253, 267-279; eval_utils.py:44-77)
# ----------------------------------------------------------------------------------------------
def _factored_ce_and_hit(factored_logits, targets, cfg):
    """factored_logits (B,Vf,nv,...), targets (B,...) -> per-token summed CE and 'all factors right'."""
    Vf, nv = cfg.factored_vocab_size, cfg.num_factored_vocabs
    fl = np.moveaxis(factored_logits, 1, -1)  # (B,nv,...,Vf)
    m = fl.max(-1, keepdims=True)
    lse = np.log(np.exp(fl - m).sum(-1)) + m[..., 0]  # (B,nv,...)
    ft = np.moveaxis(factorize_token_ids(targets, nv, Vf), -1, 1)  # (B,nv,...)  factorize_labels
    picked = np.take_along_axis(fl, ft[..., None], axis=-1)[..., 0]
    ce = (lse - picked).sum(1)
    hit = (fl.argmax(-1) == ft).all(1)
    return ce, hit


def _to_factored(logits_CTHW, cfg):
    """'b (nv Vf) ... -> b Vf nv ...'  (st_mask_git.py:171-173, 236-239)."""
    B = logits_CTHW.shape[0]
    Vf, nv = cfg.factored_vocab_size, cfg.num_factored_vocabs
    rest = logits_CTHW.shape[2:]
    return np.moveaxis(logits_CTHW.reshape((B, nv, Vf) + rest), 1, 2)


def forward_loss_acc(input_ids_flat, labels_flat, sd, cfg, nm=F32):
    """STMaskGIT.forward -> (loss, acc, logits_CTHW); mean over positions whose INPUT is MASK, frames 1.."""
    H = W = math.isqrt(cfg.S)
    B = np.asarray(input_ids_flat).shape[0]
    x = np.asarray(input_ids_flat, dtype=np.int64).reshape(B, cfg.T, H, W)
    y = np.asarray(labels_flat, dtype=np.int64).reshape(B, cfg.T, H, W)
    logits = compute_logits(x, sd, cfg, nm)
    relevant = x[:, 1:] == cfg.image_vocab_size
    ce, hit = _factored_ce_and_hit(_to_factored(logits[:, :, 1:], cfg), y[:, 1:], cfg)
    n = relevant.sum()
    with np.errstate(invalid="ignore", divide="ignore"):
        loss = (ce * relevant).sum() / n  # 0/0 -> nan, as in the reference (no guard)
        acc = np.float32((hit * relevant).sum()) / n
    return loss, acc, logits


def compute_loss(labels_flat, factored_logits, cfg):
    """eval_utils.compute_loss: plain mean over B*(T-1)*H*W of the summed factored CE -> python float."""
    B = factored_logits.shape[0]
    t = factored_logits.shape[3] + 1
    h, w = factored_logits.shape[-2:]
    y = np.asarray(labels_flat, dtype=np.int64).reshape(B, t, h, w)[:, 1:]
    ce, _ = _factored_ce_and_hit(factored_logits, y, cfg)
    return float(ce.mean())


# ----------------------------------------------------------------------------------------------
# a13/a14: MaskGIT (genie/st_mask_git.py:17-26, 115-229)
# ----------------------------------------------------------------------------------------------
def cosine_schedule(u: float) -> float:
    return math.cos(u * math.pi / 2)


def mask_counts(maskgit_steps: int, S: int):
    """n re-masked after each non-final step (st_mask_git.py:199)."""
    return [math.ceil(cosine_schedule((s + 1) / maskgit_steps) * S) for s in range(maskgit_steps - 1)]


def sample_frame(logits_CHW, cfg, temperature=0.0, uniforms=None):
    """a13: per vocab (hi first) softmax -> argmax / categorical; sample = hi*Vf+lo; conf = prod p[sample].

    logits_CHW: (B, nv*Vf, H, W).  uniforms (T>0 only): (nv, B, H, W) in [0,1), factor order = hi first.
    Categorical(probs / T) renormalises, so T only switches argmax -> sampling (st_mask_git.py:184-186).
    """
    B, _, H, W = logits_CHW.shape
    Vf, nv = cfg.factored_vocab_size, cfg.num_factored_vocabs
    fl = _to_factored(logits_CHW, cfg)  # (B,Vf,nv,H,W)
    m = fl.max(1, keepdims=True)
    e = np.exp(fl - m)
    probs = e / e.sum(1, keepdims=True)
    samples = np.zeros((B, H, W), dtype=np.int64)
    conf = np.ones((B, H, W), dtype=np.float32)
    for k, j in enumerate(range(nv - 1, -1, -1)):  # flip(2): most-significant factor first (:179)
        p = probs[:, :, j]  # (B,Vf,H,W)
        if temperature <= 1e-8:
            s = p.argmax(1)  # first max wins
        else:
            cdf = np.cumsum(np.moveaxis(p, 1, -1).astype(np.float64), -1)
            cdf /= cdf[..., -1:]
            s = (cdf < uniforms[k][..., None]).sum(-1).clip(0, Vf - 1)
        samples = samples * Vf + s
        conf = conf * np.take_along_axis(p, s[:, None], 1)[:, 0].astype(np.float32)
    return samples, conf


def mask_step(samples_flat, keys_flat, unmasked, n, mask_id):
    """a14 mask half for one non-final step; in place on samples_flat / unmasked.

    keys[unmasked]=+inf; order=argsort asc; unmasked[order[n:]]=True; samples[order[:n]]=MASK (:212-216).
    """
    keys = np.where(unmasked, np.float32(np.inf), keys_flat.astype(np.float32))
    order = np.argsort(keys, axis=1, kind="stable")
    np.put_along_axis(unmasked, order[:, n:], True, axis=1)
    np.put_along_axis(samples_flat, order[:, :n], mask_id, axis=1)


def maskgit_generate(prompt_BTHW, out_t, sd, cfg, maskgit_steps=1, temperature=0.0, unmask_mode="random",
                     noise=None, uniforms=None, nm=F32, logits_fn=None):
    """st_mask_git.py:123-229.  Mutates prompt_BTHW[:, out_t] in place; returns (samples_HW, step-0 factored logits).

    noise: (maskgit_steps-1, B, S) float32 draws replacing torch.rand_like in "random" mode.
    """
    assert out_t, "maskgit_generate requires out_t > 0"
    assert np.all(prompt_BTHW[:, out_t:] == cfg.image_vocab_size), \
        f"when generating z{out_t}, frames {out_t} and later must be masked"
    if unmask_mode not in ("greedy", "random"):
        raise NotImplementedError(f"Expected `unmask_mode` to be one of ['greedy', 'random'], got {unmask_mode}")
    B, T, H, W = prompt_BTHW.shape
    S = H * W
    logits_fn = logits_fn or (lambda p: compute_logits(p, sd, cfg, nm))
    unmasked = np.zeros((B, S), dtype=bool)
    logits_CHW = logits_fn(prompt_BTHW)[:, :, out_t]
    orig = logits_CHW.copy()
    counts = mask_counts(maskgit_steps, S)
    samples_HW = None
    for step in range(maskgit_steps):
        if step > 0:
            logits_CHW = logits_fn(prompt_BTHW)[:, :, out_t]
        u = None if uniforms is None else uniforms[step]
        samples_HW, conf = sample_frame(logits_CHW, cfg, temperature, u)
        prev_unmasked = unmasked.copy()
        prev_img = prompt_BTHW[:, out_t].reshape(B, S).copy()
        samples_flat = samples_HW.reshape(B, S).copy()
        if step != maskgit_steps - 1:
            keys = conf.reshape(B, S) if unmask_mode == "greedy" else np.asarray(noise[step]).reshape(B, S)
            mask_step(samples_flat, keys, unmasked, counts[step], cfg.image_vocab_size)
        samples_flat[prev_unmasked] = prev_img[prev_unmasked]
        samples_HW = samples_flat.reshape(B, H, W)
        prompt_BTHW[:, out_t] = samples_HW
    return samples_HW, _to_factored(orig, cfg)


def generate(input_ids_flat, max_new_tokens, sd, cfg, maskgit_steps=1, temperature=0.0, noise=None,
             return_logits=False, nm=F32):
    """STMaskGIT.generate (st_mask_git.py:65-113).  noise: (n_new, steps-1, B, S)."""
    assert max_new_tokens % cfg.S == 0
    H = W = math.isqrt(cfg.S)
    n_new = max_new_tokens // cfg.S
    ids = np.asarray(input_ids_flat, dtype=np.int64)
    B = ids.shape[0]
    x = ids.reshape(B, -1, H, W)
    t0 = x.shape[1]
    p = np.concatenate([x, np.full((B, n_new, H, W), cfg.image_vocab_size, dtype=np.int64)], 1)
    all_logits = []
    for k, t in enumerate(range(t0, t0 + n_new)):
        s, lg = maskgit_generate(p, t, sd, cfg, maskgit_steps, temperature,
                                 noise=None if noise is None else noise[k], nm=nm)
        p[:, t] = s
        all_logits.append(lg)
    out = p.reshape(B, -1)
    return (out, np.stack(all_logits, 3)) if return_logits else out


# ----------------------------------------------------------------------------------------------
# a16: teacher-forced evaluation harness (genie/evaluate.py:82-122, 167-191)
# ----------------------------------------------------------------------------------------------
def predict_zframe_logits(input_ids_flat, sd, cfg, maskgit_steps=2, temperature=0.0, noise=None, nm=F32,
                          unmask_mode="random"):
    """-> samples (B,T-1,H,W), factored logits (B,Vf,nv,T-1,H,W).  noise: (T-1, steps-1, B, S)."""
    H = W = math.isqrt(cfg.S)
    ids = np.asarray(input_ids_flat, dtype=np.int64)
    B = ids.shape[0]
    x = ids.reshape(B, cfg.T, H, W)
    all_s, all_l = [], []
    for k, t in enumerate(range(1, cfg.T)):
        p = x.copy()
        p[:, t:] = cfg.image_vocab_size
        s, lg = maskgit_generate(p, t, sd, cfg, maskgit_steps, temperature, unmask_mode,
                                 noise=None if noise is None else noise[k], nm=nm)
        all_s.append(s)
        all_l.append(lg)
    return np.stack(all_s, 1), np.stack(all_l, 3)


def evaluate_metrics(input_ids_flat, sd, cfg, maskgit_steps=2, noise=None, nm=F32):
    """loss = compute_loss, acc = mean(gt[:,1:] == samples)  (evaluate.py:177-179)."""
    H = W = math.isqrt(cfg.S)
    ids = np.asarray(input_ids_flat, dtype=np.int64)
    samples, fl = predict_zframe_logits(ids, sd, cfg, maskgit_steps, 0.0, noise, nm)
    loss = compute_loss(ids, fl, cfg)
    acc = float((ids.reshape(ids.shape[0], cfg.T, H, W)[:, 1:] == samples).astype(np.float32).mean())
    return loss, acc, samples, fl


# ----------------------------------------------------------------------------------------------
# a18: tokens -> +-1 bits (magvit2/modules/vqvae/lookup_free_quantize.py:181-194 + visualize.py:115)
# ----------------------------------------------------------------------------------------------
def bits_from_tokens(ids_BHW, codebook_dim=18, dtype=np.float32):
    """z[b,c,h,w] = +1 if bit c (LSB first) of id set else -1."""
    ids = np.asarray(ids_BHW, dtype=np.int64)
    bits = (ids[:, None] >> np.arange(codebook_dim, dtype=np.int64)[None, :, None, None]) & 1
    return (bits * 2 - 1).astype(dtype)


# ----------------------------------------------------------------------------------------------
# f16 split-pair emulation used to pin the "f16x3" HIP precision: a ~ hi + lo/2048 with hi, lo in f16
# (hi flushed to zero below the f16 normal range), i.e. a 22-bit operand; the kernel forms
# hi.hi + (hi.lo + lo.hi)/2048 in f32 and drops the lo.lo term (2^-22 relative).
# ----------------------------------------------------------------------------------------------
def round_f16_split(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    with np.errstate(over="ignore"):
        hi = a.astype(np.float16).astype(np.float32)
    hi = np.where(np.abs(hi) < np.float32(6.103515625e-05), np.float32(0), hi)
    lo = ((a - hi) * np.float32(2048.0)).astype(np.float16).astype(np.float32)
    return hi + lo * np.float32(1.0 / 2048.0)


F16X3 = Numerics(np.float32, round_f16_split)


# ----------------------------------------------------------------------------------------------
# a19 epilogue / a20 epilogue: byte and integer ends of the tokenizer path
# ----------------------------------------------------------------------------------------------
def rescale_u8_bf16(x_bf16_as_f32):
    """rescale_magvit_output on a bf16 tensor (visualize.py:84-92): every op rounds to bf16, then clamp, truncate."""
    v = round_bf16(np.asarray(x_bf16_as_f32, np.float32) + np.float32(1.0))
    v = round_bf16(v * np.float32(127.5))
    return np.clip(v, 0, 255).astype(np.uint8)


def rescale_u8_f32(x):
    v = (np.asarray(x, np.float32) + np.float32(1.0)) * np.float32(127.5)
    return np.clip(v, 0, 255).astype(np.uint8)


def tokens_from_bits(h_nchw):
    """Dataset-convention index of an encoder output: id = sum_c [h_c > 0] << c (inverse of bits_from_tokens)."""
    h = np.asarray(h_nchw)
    bits = (h > 0).astype(np.int64)
    return (bits << np.arange(h.shape[1], dtype=np.int64)[None, :, None, None]).sum(1)
