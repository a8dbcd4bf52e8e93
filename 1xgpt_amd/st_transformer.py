"""Mlp / STBlock / STTransformerDecoder -- drop-ins for genie/st_transformer.py (:7-120).

Same constructors and parameter names as the reference.  ``forward`` takes and returns the
reference's ``(B, T, S, C)`` tensor; internally nothing is permuted: the temporal kernel reads
frame-strided rows instead of materialising ``(B S) T C`` (reference :77,:82).
"""
import os
import torch
import torch.nn as nn

from . import _lib
from .attention import SelfAttention, _ptr, _require_cuda, _stream, hip_linear


class Mlp(nn.Module):
    def __init__(self, d_model: int, mlp_ratio: float = 4.0, mlp_bias: bool = True, mlp_drop: float = 0.0) -> None:
        super().__init__()
        if mlp_drop != 0.0:
            raise NotImplementedError("inference path: mlp_drop must be 0")
        hidden_dim = int(d_model * mlp_ratio)
        self.fc1 = nn.Linear(d_model, hidden_dim, bias=mlp_bias)
        self.fc2 = nn.Linear(hidden_dim, d_model, bias=mlp_bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _require_cuda(x)
        shp = x.shape
        x2 = x.contiguous().view(-1, shp[-1]).float()
        h = hip_linear(x2, self.fc1.weight, self.fc1.bias, gelu=True)  # fc1 + exact-erf GELU fused
        return hip_linear(h, self.fc2.weight, self.fc2.bias).view(shp)


def _attn_struct(a: SelfAttention, packed=None, temporal=False) -> _lib.AttnWeights:
    s = _lib.AttnWeights()
    s.qkv_w, s.qkv_b = a.qkv.weight.data_ptr(), _ptr(a.qkv.bias)
    s.proj_w, s.proj_b = a.proj.weight.data_ptr(), _ptr(a.proj.bias)
    if a.qk_norm:
        s.norm_w, s.norm_b = a.norm.weight.data_ptr(), a.norm.bias.data_ptr()
    if packed is not None:
        s.qkv_w16, s.proj_w16 = packed(a.qkv.weight), packed(a.proj.weight)
        s.w16_wide = (_lib.WIDE_QKV if packed.is_wide(s.qkv_w16) else 0) | (_lib.WIDE_PROJ if packed.is_wide(s.proj_w16) else 0)
        s.fused_w16 = packed.temporal_fused(a) if temporal else packed.spatial_fused(a)
        if not temporal and s.fused_w16 and os.environ.get("GENIE_NO_FUSED_QKV", "0") != "1":
            s.w16_wide |= _lib.FUSED_QKV_STREAM
        s.frame_w16 = packed.frame_stream(a.qkv.weight, a.proj.weight)
    return s


class STBlock(nn.Module):
    def __init__(self, num_heads: int, d_model: int, qkv_bias: bool = False, proj_bias: bool = True,
                 qk_norm: bool = True, use_mup: bool = True, attn_drop: float = 0.0, mlp_ratio: float = 4.0,
                 mlp_bias: bool = True, mlp_drop: float = 0.0) -> None:
        super().__init__()
        self.norm1 = nn.Identity() if qk_norm else nn.LayerNorm(d_model, eps=1e-05)
        self.spatial_attn = SelfAttention(num_heads=num_heads, d_model=d_model, qkv_bias=qkv_bias,
                                          proj_bias=proj_bias, qk_norm=qk_norm, use_mup=use_mup, attn_drop=attn_drop)
        self.temporal_attn = SelfAttention(num_heads=num_heads, d_model=d_model, qkv_bias=qkv_bias,
                                           proj_bias=proj_bias, qk_norm=qk_norm, use_mup=use_mup, attn_drop=attn_drop)
        self.norm2 = nn.Identity() if qk_norm else nn.LayerNorm(d_model, eps=1e-05)
        self.mlp = Mlp(d_model=d_model, mlp_ratio=mlp_ratio, mlp_bias=mlp_bias, mlp_drop=mlp_drop)
        self._meta = dict(num_heads=num_heads, d_model=d_model, qkv_bias=qkv_bias, proj_bias=proj_bias,
                          qk_norm=qk_norm, use_mup=use_mup, mlp_ratio=mlp_ratio, mlp_bias=mlp_bias)

    def layer_struct(self, packed=None) -> _lib.LayerWeights:
        lw = _lib.LayerWeights()
        if not self._meta["qk_norm"]:
            lw.norm1_w, lw.norm1_b = self.norm1.weight.data_ptr(), self.norm1.bias.data_ptr()
            lw.norm2_w, lw.norm2_b = self.norm2.weight.data_ptr(), self.norm2.bias.data_ptr()
        lw.spatial = _attn_struct(self.spatial_attn, packed)
        lw.temporal = _attn_struct(self.temporal_attn, packed, temporal=True)
        lw.fc1_w, lw.fc1_b = self.mlp.fc1.weight.data_ptr(), _ptr(self.mlp.fc1.bias)
        lw.fc2_w, lw.fc2_b = self.mlp.fc2.weight.data_ptr(), _ptr(self.mlp.fc2.bias)
        if packed is not None:
            lw.fc1_w16, lw.fc2_w16 = packed(self.mlp.fc1.weight), packed(self.mlp.fc2.weight)
            lw.w16_wide = (_lib.WIDE_FC1 if packed.is_wide(lw.fc1_w16) else 0) | (_lib.WIDE_FC2 if packed.is_wide(lw.fc2_w16) else 0)
            lw.mlp_fused_w16 = packed.mlp_fused(self.mlp)
            lw.mlp_frame_w16 = packed.frame_stream(self.mlp.fc1.weight, self.mlp.fc2.weight)
        return lw

    def _cfg(self, T, S, precision=_lib.PREC_EXACT) -> _lib.GenieCfg:
        m = self._meta
        dh = m["d_model"] // m["num_heads"]
        return _lib.GenieCfg(num_layers=1, num_heads=m["num_heads"], head_dim=dh, d_model=m["d_model"], T=T, S=S,
                             hidden=int(m["d_model"] * m["mlp_ratio"]), factored_vocab=512, num_factored=2,
                             image_vocab_size=262144, qk_norm=int(m["qk_norm"]), use_mup=int(m["use_mup"]),
                             qkv_bias=int(m["qkv_bias"]), proj_bias=int(m["proj_bias"]), mlp_bias=int(m["mlp_bias"]),
                             attn_scale=float(self.spatial_attn.scale), readout_mult=1.0, precision=precision)

    def forward(self, x_TSC: torch.Tensor) -> torch.Tensor:
        """x += SpAttn(norm1(x)); x += TmpAttn(x, causal); x += Mlp(norm2(x))  (reference :70-83)."""
        _require_cuda(x_TSC)
        lib = _lib.load()
        B, T, S, C = x_TSC.shape
        x = x_TSC.contiguous().float().clone()
        cfg = self._cfg(T, S)
        nbytes = lib.genie_workspace_bytes(cfg, B)
        if nbytes == 0:
            _lib.check(lib.genie_check_config(cfg), "genie_check_config")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        lw = self.layer_struct()
        _lib.check(lib.genie_st_block_forward(cfg, lw, x.data_ptr(), B, ws.data_ptr(), nbytes, _stream()),
                   "genie_st_block_forward")
        return x


class STTransformerDecoder(nn.Module):
    def __init__(self, num_layers: int, num_heads: int, d_model: int, qkv_bias: bool = False, proj_bias: bool = True,
                 qk_norm: bool = True, use_mup: bool = True, attn_drop: float = 0.0, mlp_ratio: float = 4.0,
                 mlp_bias: bool = True, mlp_drop: float = 0.0):
        super().__init__()
        self.layers = nn.ModuleList([STBlock(num_heads=num_heads, d_model=d_model, qkv_bias=qkv_bias,
                                             proj_bias=proj_bias, qk_norm=qk_norm, use_mup=use_mup,
                                             attn_drop=attn_drop, mlp_ratio=mlp_ratio, mlp_bias=mlp_bias,
                                             mlp_drop=mlp_drop) for _ in range(num_layers)])

    def forward(self, tgt: torch.Tensor) -> torch.Tensor:
        x = tgt
        for layer in self.layers:
            x = layer(x)
        return x
