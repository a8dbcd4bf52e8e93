"""Synthetic weights and clips, bit-reproducible on any box with this image's numpy.

There are no trained checkpoints offline, so parity and throughput are measured on synthetic
weights.  Two laws:

* ``law="init"``  -- the reference's ``init_weights`` law (genie/st_mask_git.py:281-296): N(0, 0.02)
  Linear/Embedding weights, zero biases, LayerNorm gamma=1 beta=0; plus N(0, 0.02) pos-embed and
  mask-embed (zeros at construction in the reference -- a degenerate case).
* ``law="conditioned"`` -- every activation O(1), non-zero biases / LN affine, logits with std ~2 so
  that argmax, CE and the confidence ordering are numerically meaningful.  This is the law used by
  the golden fixtures, the parity tests and the bench.

Each tensor draws from its own PCG64 stream keyed by (seed, crc32(state-dict key)), so tensors are
independent of iteration order and of which other tensors exist.  Keys/shapes follow the reference
state dict.
"""
import zlib

import numpy as np

from .config import GenieConfig


def state_dict_spec(cfg: GenieConfig):
    """[(key, shape, kind, fan_in)] in reference state-dict order."""
    d, L, Dh = cfg.d_model, cfg.num_layers, cfg.head_dim
    hid = int(d * cfg.mlp_ratio)
    V = cfg.factored_vocab_size * cfg.num_factored_vocabs
    spec = [("pos_embed_TSC", (1, cfg.T, cfg.S, d), "pos", 1)]
    for i in range(L):
        p = f"decoder.layers.{i}."
        if not cfg.qk_norm:
            spec += [(p + "norm1.weight", (d,), "gamma", 1), (p + "norm1.bias", (d,), "beta", 1)]
        for a in ("spatial_attn", "temporal_attn"):
            spec.append((p + a + ".qkv.weight", (3 * d, d), "lin", d))
            if cfg.qkv_bias:
                spec.append((p + a + ".qkv.bias", (3 * d,), "bias", 1))
            spec.append((p + a + ".proj.weight", (d, d), "lin_res", d))
            if cfg.proj_bias:
                spec.append((p + a + ".proj.bias", (d,), "bias", 1))
            if cfg.qk_norm:
                spec += [(p + a + ".norm.weight", (Dh,), "gamma", 1), (p + a + ".norm.bias", (Dh,), "beta", 1)]
        if not cfg.qk_norm:
            spec += [(p + "norm2.weight", (d,), "gamma", 1), (p + "norm2.bias", (d,), "beta", 1)]
        spec.append((p + "mlp.fc1.weight", (hid, d), "lin", d))
        if cfg.mlp_bias:
            spec.append((p + "mlp.fc1.bias", (hid,), "bias", 1))
        spec.append((p + "mlp.fc2.weight", (d, hid), "lin_res", hid))
        if cfg.mlp_bias:
            spec.append((p + "mlp.fc2.bias", (d,), "bias", 1))
    spec.append(("token_embed.mask_token_embed", (1, d), "emb", 1))
    for j in range(cfg.num_factored_vocabs):
        spec.append((f"token_embed.factored_embeds.{j}.weight", (cfg.factored_vocab_size, d), "emb", 1))
    spec.append(("out_x_proj.weight", (V, d), "readout", d))
    spec.append(("out_x_proj.bias", (V,), "bias", 1))
    return spec


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, zlib.crc32(key.encode())])))


def make_state_dict(cfg: GenieConfig, seed: int = 0, law: str = "conditioned") -> dict:
    """{key: float32 ndarray} for ``cfg``."""
    L = cfg.num_layers
    out = {}
    for key, shape, kind, fan_in in state_dict_spec(cfg):
        g = _rng(seed, key)
        z = g.standard_normal(shape, dtype=np.float32)
        if law == "init":
            if kind in ("lin", "lin_res", "readout", "emb", "pos"):
                w = z * np.float32(0.02)
            elif kind == "gamma":
                w = np.ones(shape, np.float32)
            else:  # bias, beta
                w = np.zeros(shape, np.float32)
        elif law == "conditioned":
            if kind == "lin":
                w = z * np.float32(1.0 / np.sqrt(fan_in))
            elif kind == "lin_res":  # residual branches sum to O(1) over 3L additions
                w = z * np.float32(1.0 / np.sqrt(fan_in) / np.sqrt(1.5 * L))
            elif kind == "readout":
                w = z * np.float32(1.0 / np.sqrt(fan_in))
            elif kind == "emb":
                w = z * np.float32(0.7)
            elif kind == "pos":
                w = z * np.float32(0.5)
            elif kind == "gamma":
                w = np.float32(1.0) + z * np.float32(0.1)
            else:  # bias, beta
                w = z * np.float32(0.05)
        else:
            raise ValueError(f"unknown law {law!r}")
        out[key] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def make_clips(n_clips: int, cfg: GenieConfig, seed: int = 1234) -> np.ndarray:
    """(n_clips, T*S) int64 token ids ~ U{0..image_vocab_size-1}; labels = ids (reference: data.py:102-105)."""
    g = np.random.default_rng(seed)
    return g.integers(0, cfg.image_vocab_size, size=(n_clips, cfg.T * cfg.S), dtype=np.int64)


def make_noise(shape, seed: int = 42) -> np.ndarray:
    """U[0,1) float32 draws standing in for ``torch.rand_like`` (reference: genie/st_mask_git.py:206)."""
    g = np.random.default_rng(seed)
    return g.random(shape, dtype=np.float32)
