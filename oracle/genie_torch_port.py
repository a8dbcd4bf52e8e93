"""GENIE forward restated with torch CPU ops -- TEST / BASELINE INFRASTRUCTURE, not product code.

Why it exists beside the NumPy oracle (oracle/genie_oracle.py): bench.py's `cpu_baseline` leg should time what the
reference's CPU path actually executes -- torch f32 ops (addmm / bmm / softmax / layer_norm / gelu, oneDNN + MKL threads;
genie/evaluate.py with device="cpu") -- and the reference's Python does not travel to the GPU box.  The NumPy oracle is
3-4x slower than that on the same cores (OpenBLAS, NumPy softmax / erf), which understated the CPU path (VERDICT r1,
weak 8).  Same formulas as SURVEY.md Appendix A; each step cites the reference line it follows.  Pinned to the NumPy
oracle (itself pinned to the reference's outputs) by tests/test_oracle_golden.py::test_torch_port_matches_oracle.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np
import torch
import torch.nn.functional as F


def to_torch(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def _attention(x, sd, p, cfg, causal):
    """attention.py:36-61 (BasicSelfAttention): x (B', N, C)."""
    Bn, N, C = x.shape
    H, Dh = cfg.num_heads, cfg.head_dim
    qkv = F.linear(x, sd[p + "qkv.weight"], sd.get(p + "qkv.bias") if cfg.qkv_bias else None)        # :38
    q, k, v = qkv.reshape(Bn, N, 3, H, Dh).permute(2, 0, 3, 1, 4)                                     # (3,B',H,N,Dh)
    if cfg.qk_norm:                                                                                      # :42-47
        q = F.layer_norm(q, (Dh,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)
        k = F.layer_norm(k, (Dh,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)
    attn = (q * cfg.attn_scale) @ k.transpose(-2, -1)                                                   # :48-49
    if causal:                                                                                           # :51-55
        mask = torch.ones(N, N, dtype=torch.bool).tril_().logical_not_()
        attn = attn.masked_fill(mask, -torch.finfo(attn.dtype).max)
    o = (attn.softmax(-1) @ v).transpose(1, 2).reshape(Bn, N, C)                                        # :57-59
    return F.linear(o, sd[p + "proj.weight"], sd.get(p + "proj.bias") if cfg.proj_bias else None)     # :60


def _block(x, sd, i, cfg):
    """st_transformer.py:70-83."""
    B, T, S, C = x.shape
    p = f"decoder.layers.{i}."
    xs = x.reshape(B * T, S, C)
    u = xs if cfg.qk_norm else F.layer_norm(xs, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
    xs = xs + _attention(u, sd, p + "spatial_attn.", cfg, False)                                        # :73-74
    xt = xs.reshape(B, T, S, C).transpose(1, 2).reshape(B * S, T, C)                                    # :77
    xt = xt + _attention(xt, sd, p + "temporal_attn.", cfg, True)                                       # :78 (no pre-norm)
    u = xt if cfg.qk_norm else F.layer_norm(xt, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    h = F.gelu(F.linear(u, sd[p + "mlp.fc1.weight"], sd.get(p + "mlp.fc1.bias") if cfg.mlp_bias else None))
    xt = xt + F.linear(h, sd[p + "mlp.fc2.weight"], sd.get(p + "mlp.fc2.bias") if cfg.mlp_bias else None)  # :81
    return xt.reshape(B, S, T, C).transpose(1, 2)                                                      # :82


@torch.no_grad()
def compute_logits(ids_BTHW, sd, cfg):
    """st_mask_git.py:255-265 -> (B, V, T, H, W) float32 ndarray.  sd: dict of torch tensors (to_torch)."""
    ids = torch.from_numpy(np.asarray(ids_BTHW, dtype=np.int64))
    B, T, H, W = ids.shape
    ids = ids.reshape(B, T, H * W)
    is_mask = ids == cfg.image_vocab_size                                                               # st_mask_git.py:51
    safe = torch.where(is_mask, torch.zeros_like(ids), ids)
    e = None
    for j in range(cfg.num_factored_vocabs):                                                            # factorization_utils.py:55-68
        f = (safe // cfg.factored_vocab_size ** j) % cfg.factored_vocab_size
        ej = sd[f"token_embed.factored_embeds.{j}.weight"][f]
        e = ej if e is None else e + ej
    x = torch.where(is_mask[..., None], sd["token_embed.mask_token_embed"][0], e) + sd["pos_embed_TSC"]  # :257-261
    for i in range(cfg.num_layers):
        x = _block(x, sd, i, cfg)
    if cfg.use_mup:
        x = x * cfg.readout_mult                                                                        # :316-323
    lg = F.linear(x, sd["out_x_proj.weight"], sd["out_x_proj.bias"])
    return lg.reshape(B, T, H, W, -1).permute(0, 4, 1, 2, 3).contiguous().numpy()                      # :264
