"""SelfAttention -- QKV linear, optional per-head qk-LayerNorm, softmax attention, out-proj.

Drop-in for the reference's ``genie.attention.SelfAttention`` (attention.py:9-61): same constructor,
same parameter names (``qkv.weight``, ``proj.weight/bias``, ``norm.weight/bias``), same
``forward(x (B,N,C), causal=False)``.  The reference switches between a torch and an xformers back-end
with the XFORMERS_DISABLED env var; here there is exactly one back-end: the gfx950 kernels behind
libgenie_hip.so (f32-MFMA GEMMs + LDS-staged attention).  ``nn.Linear``/``nn.LayerNorm`` are used as
parameter holders only.
"""
import torch
import torch.nn as nn

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _require_cuda(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("1xgpt_amd runs on the GPU only (no CPU fallback): move module and inputs to cuda")


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def hip_linear(x2d: torch.Tensor, weight: torch.Tensor, bias, out: torch.Tensor = None, gelu=False,
               accumulate=False) -> torch.Tensor:
    """y = x W^T (+b) [gelu] [y += ...] through genie_linear (f32 MFMA)."""
    lib = _lib.load()
    for t in (x2d, weight, bias, out):
        if t is not None and (t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous()):
            raise TypeError("hip_linear operands must be contiguous float32 cuda tensors")
    M, K = x2d.shape
    N = weight.shape[0]
    if weight.shape[1] != K:
        raise ValueError(f"hip_linear: x is (M,{K}) but weight is {tuple(weight.shape)}")
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x2d.device)
    _lib.check(lib.genie_linear(x2d.data_ptr(), weight.data_ptr(), _ptr(bias), out.data_ptr(), M, N, K, int(gelu),
                                int(accumulate), _stream()), "genie_linear")
    return out


class SelfAttention(nn.Module):
    def __init__(self, num_heads: int, d_model: int, qkv_bias: bool = False, proj_bias: bool = True,
                 qk_norm: bool = True, use_mup: bool = True, attn_drop: float = 0.0) -> None:
        super().__init__()
        if attn_drop != 0.0:
            raise NotImplementedError("attn_drop is unused by the reference's forward (attention.py:28) and must be 0")
        self.num_heads = num_heads
        self.head_dim = d_model // num_heads
        self.scale = 8 / self.head_dim if use_mup else self.head_dim ** -0.5  # attention.py:26
        self.qkv = nn.Linear(d_model, d_model * 3, bias=qkv_bias)
        self.proj = nn.Linear(d_model, d_model, bias=proj_bias)
        self.qk_norm = qk_norm
        if self.qk_norm:
            self.norm = nn.LayerNorm(self.head_dim, eps=1e-05)

    def forward(self, x: torch.Tensor, causal: bool = False) -> torch.Tensor:
        _require_cuda(x)
        lib = _lib.load()
        B, N, C = x.shape
        x2 = x.contiguous().view(B * N, C).float()
        qkv = hip_linear(x2, self.qkv.weight, self.qkv.bias)
        ao = torch.empty(B * N, C, dtype=torch.float32, device=x.device)
        nw = self.norm.weight if self.qk_norm else None
        nb = self.norm.bias if self.qk_norm else None
        _lib.check(lib.genie_attention_core(qkv.data_ptr(), ao.data_ptr(), B, N, self.num_heads, self.head_dim,
                                            float(self.scale), int(causal), _ptr(nw), _ptr(nb), _stream()),
                   "genie_attention_core")
        return hip_linear(ao, self.proj.weight, self.proj.bias).view(B, N, C)


# the reference exports both names; there is a single implementation here
BasicSelfAttention = SelfAttention
MemoryEfficientAttention = SelfAttention
