"""CPU restatement of the reference's TRAINING step -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
path (1xgpt_amd/) never does.  Plain NumPy, forward with saved intermediates and a hand-derived backward:

  * MaskGIT collator                      data.py:109-169 (draws injected so a run is replayable)
  * forward + masked factored CE          genie/st_mask_git.py:231-279 (via oracle/genie_oracle.py)
  * backward                              what autograd derives for those lines; pinned against the reference's own
                                          ``loss.backward()`` gradients in tests/golden/train_*.npz
  * gradient clipping                     torch.nn.utils.clip_grad_norm_ as called at train.py:628-629
  * AdamW with the reference's grouping   train.py:426-441 (decay everything whose name has no "bias" /
                                          "layer_norm.weight" substring -- LayerNorm weights therefore DO decay)
  * learning-rate factor                  train.py:468-481 ("custom_cosine") and the linear schedule default

MuAdamW (train.py:439, third-party ``mup`` fork, not vendored) is not restated: parity unpinned, not built.
Dropout is 0 in every shipped config (genie/config.py:32,37) and is not modelled.
"""
import math

import numpy as np

from . import genie_oracle as go

_SQRT_2PI = math.sqrt(2.0 * math.pi)


# ----------------------------------------------------------------------------------------------
# collator (data.py:109-169)
# ----------------------------------------------------------------------------------------------
class ReplayDraws:
    """Replays draws captured from the reference collator, in call order (tools/make_goldens_train.py)."""

    def __init__(self, kinds, arrays):
        self.items = list(zip([str(k) for k in kinds], arrays))
        self.pos = 0

    def _next(self, kind):
        k, a = self.items[self.pos]
        assert k == kind, f"draw {self.pos}: reference drew {k}, restatement asks for {kind}"
        self.pos += 1
        return a

    def rand(self, shape):
        a = self._next("torch.rand")
        assert tuple(a.shape) == tuple(shape), (a.shape, shape)
        return a

    def rand_like(self, shape):
        a = self._next("torch.rand_like")
        assert tuple(a.shape) == tuple(shape), (a.shape, shape)
        return a

    def randint(self, high, shape):
        a = self._next("torch.randint")
        assert tuple(a.shape) == tuple(shape) and a.max() < high
        return a

    def py_random(self):
        return float(self._next("py.random"))

    def py_randint(self, a, b):
        v = int(self._next("py.randint"))
        assert a <= v <= b
        return v

    def py_uniform(self, a, b):
        return float(self._next("py.uniform"))


class NumpyDraws:
    """Fresh draws from a NumPy generator (same protocol)."""

    def __init__(self, seed):
        self.g = np.random.default_rng(seed)

    def rand(self, shape):
        return self.g.random(shape, dtype=np.float32)

    rand_like = rand

    def randint(self, high, shape):
        return self.g.integers(0, high, size=shape, dtype=np.int64)

    def py_random(self):
        return float(self.g.random())

    def py_randint(self, a, b):
        return int(self.g.integers(a, b + 1))

    def py_uniform(self, a, b):
        return float(a + (b - a) * self.g.random())


def maskgit_collate(ids_flat, cfg, draws):
    """(B, T*S) int64 clips -> dict(input_ids, labels), following data.py:112-167 draw for draw."""
    ids = np.asarray(ids_flat, dtype=np.int64)
    B = ids.shape[0]
    h = w = math.isqrt(cfg.S)
    nv, Vf = cfg.num_factored_vocabs, cfg.factored_vocab_size
    x_THW = ids.reshape(B, cfg.T, h, w)
    x_THWC = go.factorize_token_ids(x_THW, nv, Vf)
    labels = x_THW.copy()
    r = draws.rand(x_THWC.shape)
    u01 = np.float32(draws.rand(()))
    thr = np.float32(u01 * np.float32(cfg.max_corrupt_rate))  # python float x 0-dim f32 tensor -> f32 product
    random_values = draws.randint(Vf, x_THWC.shape)
    m = r < thr
    x_THWC = np.where(m, random_values, x_THWC)
    if draws.py_random() < cfg.non_mlm_ratio:
        first = draws.py_randint(cfg.num_prompt_frames, cfg.T - 1)
        correct_rate = draws.py_uniform(0.25, 1.0)
        for i in range(cfg.T - first):
            correct_rate *= draws.py_uniform(0.9, 1.0)
            r = draws.rand((B, h, w, nv))
            m = r > np.float32(correct_rate)
            x_THWC[:, first + i] = np.where(m, random_values[:, first + i], x_THWC[:, first + i])
    else:
        first = 1
    while True:
        u = draws.rand((B, cfg.T - first, 1, 1)).astype(np.float32)
        prob = np.cos(u * np.float32(math.pi) / np.float32(2)).astype(np.float32)
        r = draws.rand_like((B, cfg.T - first, h, w))
        mask = r < prob
        if mask.max():
            break
    x = go.unfactorize_token_ids(x_THWC, nv, Vf)
    x[:, first:][mask] = cfg.image_vocab_size
    return {"input_ids": x.reshape(B, -1), "labels": labels.reshape(B, -1)}


# ----------------------------------------------------------------------------------------------
# forward with saved intermediates, and backward
# ----------------------------------------------------------------------------------------------
def _ln_fwd(x, g, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    xc = x - mu
    rstd = 1.0 / np.sqrt((xc * xc).mean(-1, keepdims=True) + x.dtype.type(eps))
    xh = xc * rstd
    return xh * g + b, (xh, rstd)


def _ln_bwd(dy, cache, g):
    xh, rstd = cache
    red = tuple(range(dy.ndim - 1))
    dg = (dy * xh).sum(red)
    db = dy.sum(red)
    gy = dy * g
    dx = (gy - gy.mean(-1, keepdims=True) - xh * (gy * xh).mean(-1, keepdims=True)) * rstd
    return dx, dg, db


def _gelu_grad(z):
    cdf = 0.5 * (1.0 + go._erf(z / math.sqrt(2.0)))
    pdf = np.exp(-0.5 * z * z) / _SQRT_2PI
    return (cdf + z * pdf).astype(z.dtype)


def _attn_fwd(x, sd, prefix, cfg, causal):
    dt = x.dtype
    Bn, N, C = x.shape
    H, Dh = cfg.num_heads, cfg.head_dim
    Wqkv, Wp = sd[prefix + "qkv.weight"].astype(dt), sd[prefix + "proj.weight"].astype(dt)
    qkv = x @ Wqkv.T
    if cfg.qkv_bias:
        qkv = qkv + sd[prefix + "qkv.bias"].astype(dt)
    qkv = qkv.reshape(Bn, N, 3, H, Dh).transpose(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    cq = ck = None
    if cfg.qk_norm:
        g, b = sd[prefix + "norm.weight"].astype(dt), sd[prefix + "norm.bias"].astype(dt)
        q, cq = _ln_fwd(q, g, b)
        k, ck = _ln_fwd(k, g, b)
    qs = q * dt.type(cfg.attn_scale)
    s = qs @ k.transpose(0, 1, 3, 2)
    if causal:
        s = np.where(~np.tril(np.ones((N, N), dtype=bool)), -np.finfo(dt).max, s)
    p = go._softmax_last(s)
    a = (p @ v).transpose(0, 2, 1, 3).reshape(Bn, N, C)
    o = a @ Wp.T
    if cfg.proj_bias:
        o = o + sd[prefix + "proj.bias"].astype(dt)
    return o, (x, qs, k, v, p, a, cq, ck)


def _attn_bwd(do, cache, sd, prefix, cfg, grads):
    x, qs, k, v, p, a, cq, ck = cache
    dt = x.dtype
    Bn, N, C = x.shape
    H, Dh = cfg.num_heads, cfg.head_dim
    Wqkv, Wp = sd[prefix + "qkv.weight"].astype(dt), sd[prefix + "proj.weight"].astype(dt)
    _acc(grads, prefix + "proj.weight", do.reshape(-1, C).T @ a.reshape(-1, C))
    if cfg.proj_bias:
        _acc(grads, prefix + "proj.bias", do.reshape(-1, C).sum(0))
    da = (do @ Wp).reshape(Bn, N, H, Dh).transpose(0, 2, 1, 3)  # (Bn,H,N,Dh)
    dv = p.transpose(0, 1, 3, 2) @ da
    dp = da @ v.transpose(0, 1, 3, 2)
    ds = p * (dp - (p * dp).sum(-1, keepdims=True))
    dqs = ds @ k
    dk = ds.transpose(0, 1, 3, 2) @ qs
    dq = dqs * dt.type(cfg.attn_scale)
    if cfg.qk_norm:
        g = sd[prefix + "norm.weight"].astype(dt)
        dq, dg1, db1 = _ln_bwd(dq, cq, g)
        dk, dg2, db2 = _ln_bwd(dk, ck, g)
        _acc(grads, prefix + "norm.weight", dg1 + dg2)
        _acc(grads, prefix + "norm.bias", db1 + db2)
    dqkv = np.stack([dq, dk, dv]).transpose(1, 3, 0, 2, 4).reshape(Bn, N, 3 * C)
    _acc(grads, prefix + "qkv.weight", dqkv.reshape(-1, 3 * C).T @ x.reshape(-1, C))
    if cfg.qkv_bias:
        _acc(grads, prefix + "qkv.bias", dqkv.reshape(-1, 3 * C).sum(0))
    return dqkv @ Wqkv


def _acc(grads, key, val):
    grads[key] = val if key not in grads else grads[key] + val


def forward_backward(input_ids_flat, labels_flat, sd, cfg, dtype=np.float32):
    """STMaskGIT.forward (st_mask_git.py:267-279) and d loss / d every parameter.

    Returns (loss, acc, grads) with grads keyed and shaped like the state dict."""
    dt = np.dtype(dtype)
    nm = go.Numerics(dtype=dt.type)
    H_ = W_ = math.isqrt(cfg.S)
    ids = np.asarray(input_ids_flat, dtype=np.int64)
    B = ids.shape[0]
    T, S, d = cfg.T, cfg.S, cfg.d_model
    x_in = ids.reshape(B, T, S)
    y = np.asarray(labels_flat, dtype=np.int64).reshape(B, T, S)
    sdt = {k: np.asarray(v).astype(dt) for k, v in sd.items()}
    x = go.embed(x_in, sdt, cfg, nm).astype(dt)
    caches = []
    for i in range(cfg.num_layers):
        p = f"decoder.layers.{i}."
        c = {}
        xs = x.reshape(B * T, S, d)
        if cfg.qk_norm:
            u = xs
        else:
            u, c["ln1"] = _ln_fwd(xs, sdt[p + "norm1.weight"], sdt[p + "norm1.bias"])
        o, c["sp"] = _attn_fwd(u, sdt, p + "spatial_attn.", cfg, False)
        x = (xs + o).reshape(B, T, S, d)
        xt = x.transpose(0, 2, 1, 3).reshape(B * S, T, d)
        o, c["tp"] = _attn_fwd(xt, sdt, p + "temporal_attn.", cfg, True)
        x = (xt + o).reshape(B, S, T, d).transpose(0, 2, 1, 3)
        if cfg.qk_norm:
            u = x
        else:
            u, c["ln2"] = _ln_fwd(x, sdt[p + "norm2.weight"], sdt[p + "norm2.bias"])
        z = u @ sdt[p + "mlp.fc1.weight"].T
        if cfg.mlp_bias:
            z = z + sdt[p + "mlp.fc1.bias"]
        hh = go.gelu_erf(z)
        o = hh @ sdt[p + "mlp.fc2.weight"].T
        if cfg.mlp_bias:
            o = o + sdt[p + "mlp.fc2.bias"]
        c["mlp"] = (u, z, hh)
        x = x + o
        caches.append(c)
    rho = dt.type(cfg.readout_mult) if cfg.use_mup else dt.type(1.0)
    Wo, bo = sdt["out_x_proj.weight"], sdt["out_x_proj.bias"]
    logits = (x * rho) @ Wo.T + bo  # (B,T,S,V)

    # ---- masked factored CE over frames 1.. (st_mask_git.py:231-253) and d loss / d logits
    Vf, nv = cfg.factored_vocab_size, cfg.num_factored_vocabs
    relevant = x_in[:, 1:] == cfg.image_vocab_size  # (B,T-1,S)
    n = relevant.sum()
    fl = logits[:, 1:].reshape(B, T - 1, S, nv, Vf)
    m = fl.max(-1, keepdims=True)
    e = np.exp(fl - m)
    se = e.sum(-1, keepdims=True)
    prob = e / se
    lse = np.log(se[..., 0]) + m[..., 0]
    ft = go.factorize_token_ids(y[:, 1:], nv, Vf)  # (B,T-1,S,nv)
    picked = np.take_along_axis(fl, ft[..., None], axis=-1)[..., 0]
    ce = (lse - picked).sum(-1)
    hit = (fl.argmax(-1) == ft).all(-1)
    with np.errstate(invalid="ignore", divide="ignore"):
        loss = (ce * relevant).sum() / n
        acc = np.float32((hit * relevant).sum()) / n
    onehot = np.zeros_like(prob)
    np.put_along_axis(onehot, ft[..., None], 1.0, axis=-1)
    dfl = (prob - onehot) * (relevant[..., None, None] / dt.type(n)).astype(dt)
    dlogits = np.zeros_like(logits)
    dlogits[:, 1:] = dfl.reshape(B, T - 1, S, nv * Vf)

    # ---- backward
    grads = {}
    V = nv * Vf
    grads["out_x_proj.weight"] = dlogits.reshape(-1, V).T @ (x * rho).reshape(-1, d)
    grads["out_x_proj.bias"] = dlogits.reshape(-1, V).sum(0)
    dx = (dlogits @ Wo) * rho
    for i in reversed(range(cfg.num_layers)):
        p = f"decoder.layers.{i}."
        c = caches[i]
        u, z, hh = c["mlp"]
        hid = z.shape[-1]
        _acc(grads, p + "mlp.fc2.weight", dx.reshape(-1, d).T @ hh.reshape(-1, hid))
        if cfg.mlp_bias:
            _acc(grads, p + "mlp.fc2.bias", dx.reshape(-1, d).sum(0))
        dz = (dx @ sdt[p + "mlp.fc2.weight"]) * _gelu_grad(z)
        _acc(grads, p + "mlp.fc1.weight", dz.reshape(-1, hid).T @ u.reshape(-1, d))
        if cfg.mlp_bias:
            _acc(grads, p + "mlp.fc1.bias", dz.reshape(-1, hid).sum(0))
        du = dz @ sdt[p + "mlp.fc1.weight"]
        if cfg.qk_norm:
            dx = dx + du
        else:
            dxl, dg, db = _ln_bwd(du, c["ln2"], sdt[p + "norm2.weight"])
            grads[p + "norm2.weight"], grads[p + "norm2.bias"] = dg, db
            dx = dx + dxl
        # temporal (no pre-norm)
        dxt = dx.transpose(0, 2, 1, 3).reshape(B * S, T, d)
        dxt = dxt + _attn_bwd(dxt, c["tp"], sdt, p + "temporal_attn.", cfg, grads)
        dx = dxt.reshape(B, S, T, d).transpose(0, 2, 1, 3)
        # spatial
        dxs = dx.reshape(B * T, S, d)
        du = _attn_bwd(dxs, c["sp"], sdt, p + "spatial_attn.", cfg, grads)
        if cfg.qk_norm:
            dxs = dxs + du
        else:
            dxl, dg, db = _ln_bwd(du, c["ln1"], sdt[p + "norm1.weight"])
            grads[p + "norm1.weight"], grads[p + "norm1.bias"] = dg, db
            dxs = dxs + dxl
        dx = dxs.reshape(B, T, S, d)
    # embedding (factorization_utils.py:29-52, st_mask_git.py:257-261)
    grads["pos_embed_TSC"] = dx.sum(0, keepdims=True)
    is_mask = x_in == cfg.image_vocab_size
    grads["token_embed.mask_token_embed"] = dx[is_mask].sum(0, keepdims=True)
    fac = go.factorize_token_ids(np.where(is_mask, 0, x_in), nv, Vf)
    keep = ~is_mask
    for j in range(nv):
        gE = np.zeros((Vf, d), dtype=dt)
        np.add.at(gE, fac[..., j][keep], dx[keep])
        grads[f"token_embed.factored_embeds.{j}.weight"] = gE
    return float(loss), float(acc), {k: np.ascontiguousarray(v.reshape(np.asarray(sd[k]).shape)) for k, v in grads.items()}


# ----------------------------------------------------------------------------------------------
# optimizer step (train.py:426-441, 628-633)
# ----------------------------------------------------------------------------------------------
def decays(name: str) -> bool:
    """train.py:427-437: parameters whose NAME contains "bias" or "layer_norm.weight" are excluded from decay.
    No GENIE parameter is named "layer_norm.*" (they are norm1/norm2/norm), so LayerNorm weights decay."""
    return not ("bias" in name or "layer_norm.weight" in name)


def grad_norm(grads):
    """Global L2 norm as torch.nn.utils.clip_grad_norm_ computes it (norm of per-tensor norms)."""
    return float(np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads.values())))


def clip_coef(total_norm, max_norm):
    return min(1.0, max_norm / (total_norm + 1e-6))


def adamw_step(params, grads, state, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, grad_scale=1.0):
    """torch.optim.AdamW (decoupled decay), in place on `params` (dict of f32 arrays); `step` counts from 1."""
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    for k, p in params.items():
        g = grads[k].astype(np.float32) * np.float32(grad_scale)
        m, v = state.setdefault(k, (np.zeros_like(p), np.zeros_like(p)))
        if decays(k):
            p *= np.float32(1.0 - lr * weight_decay)
        m *= np.float32(beta1)
        m += np.float32(1.0 - beta1) * g
        v *= np.float32(beta2)
        v += np.float32(1.0 - beta2) * g * g
        denom = np.sqrt(v) / np.float32(math.sqrt(bc2)) + np.float32(eps)
        p -= np.float32(lr / bc1) * (m / denom)


def lr_factor_custom_cosine(step, warmup_steps, max_steps, end_ratio=0.1):
    """train.py:468-477."""
    if step < warmup_steps:
        return (step + 1) / warmup_steps
    remaining = max_steps - warmup_steps
    return ((1 + math.cos(math.pi * (step - warmup_steps) / remaining)) / 2) * (1 - end_ratio) + end_ratio


def lr_factor_linear(step, warmup_steps, max_steps):
    """transformers.get_scheduler("linear") -- train.py's default --lr_scheduler_type."""
    if step < warmup_steps:
        return step / max(1, warmup_steps)
    return max(0.0, (max_steps - step) / max(1, max_steps - warmup_steps))
