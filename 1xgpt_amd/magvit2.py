"""MAGVIT2 frame tokenizer, inference subset, on the same device as the world model.

Counterpart of the reference's magvit2/ package restricted to what the hot path needs:
``Decoder`` (18 x 16 x 16 +-1 bits -> 3 x 256 x 256), ``Encoder`` (the mirror), the LFQ bit (un)packing and the
``decode_latents_wrapper`` of visualize.py:95-122.  State-dict keys equal the reference's ``encoder.*`` /
``decoder.*`` keys, so a Lightning ``magvit2.ckpt`` loads with ``load_tokenizer_ckpt``.

One execution: ``HipDecoder`` / ``HipEncoder`` -- hand-written gfx950 kernels behind the C ABI (NHWC bf16, implicit-GEMM
3x3 convolutions on the bf16 matrix cores with a gathered A operand and fused bias / ResBlock skip / depth-to-space
epilogue, fused GroupNorm+swish with order-fixed statistics, bit/byte ends ``genie_bits_from_tokens*``,
``genie_rescale_u8_*``, ``genie_tokens_from_*``).  Tokens and frames never leave HBM (the reference round-trips through
numpy and PIL, eval_utils.py:39-41).  The ``nn.Module`` classes below are PARAMETER HOLDERS with the reference's
state-dict keys (so ``magvit2.ckpt`` loads by name); they have no arithmetic of their own -- calling one raises.  There is
no second backend: a geometry the kernels do not cover (ResBlock widths must be multiples of 64; the shipped config is
128/256/512) is an error, not a fallback.  The plain-torch formulation lives in ``oracle/magvit2_oracle.py`` (test
infrastructure).
"""
import json
import math
import zlib
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn as nn

from . import _lib


@dataclass
class VQConfig:
    """Architecture fields of the reference's magvit2/config.py:9-43 (loss/training fields are accepted and ignored)."""
    in_channels: int = 3
    z_channels: int = 18
    out_channels: int = 3
    base_channels: int = 128
    ch_mult: tuple = (1, 1, 2, 2, 4)
    num_res_blocks: int = 2
    num_codebooks: int = 1
    codebook_size: int = 262144
    token_factorization: bool = False

    @classmethod
    def from_pretrained(cls, json_path):
        with open(json_path) as f:
            raw = json.load(f)
        known = {"in_channels", "z_channels", "out_channels", "base_channels", "ch_mult", "num_res_blocks",
                 "num_codebooks", "codebook_size", "token_factorization"}
        kw = {k: v for k, v in raw.items() if k in known}
        if "ch_mult" in kw:
            kw["ch_mult"] = tuple(kw["ch_mult"])
        return cls(**kw)


def _holder_forward(self, *args, **kwargs):
    raise RuntimeError(f"{type(self).__name__} is a parameter holder (reference state-dict keys); the arithmetic runs in "
                       "HipDecoder / HipEncoder (VQModel.decode_tokens / encode_tokens)")


class ResBlock(nn.Module):
    """GN32 -> swish -> conv3x3 -> GN32 -> swish -> conv3x3 (+ 1x1 shortcut on channel change), no conv biases
    (improved_model.py:12-51)."""

    def __init__(self, in_filters, out_filters):
        super().__init__()
        self.in_filters, self.out_filters = in_filters, out_filters
        self.norm1 = nn.GroupNorm(32, in_filters, eps=1e-6)
        self.norm2 = nn.GroupNorm(32, out_filters, eps=1e-6)
        self.conv1 = nn.Conv2d(in_filters, out_filters, kernel_size=(3, 3), padding=1, bias=False)
        self.conv2 = nn.Conv2d(out_filters, out_filters, kernel_size=(3, 3), padding=1, bias=False)
        if in_filters != out_filters:
            self.nin_shortcut = nn.Conv2d(in_filters, out_filters, kernel_size=(1, 1), padding=0, bias=False)

    forward = _holder_forward


class Upsampler(nn.Module):
    """conv3x3 C -> 4C (bias) whose output the kernel stores depth-to-space (DCR order, improved_model.py:185-237)."""

    def __init__(self, dim):
        super().__init__()
        self.conv1 = nn.Conv2d(dim, dim * 4, (3, 3), padding=1)

    forward = _holder_forward


class _Level(nn.Module):
    pass


class Encoder(nn.Module):
    """improved_model.py:54-121."""

    def __init__(self, config: VQConfig):
        super().__init__()
        self.num_res_blocks, self.num_blocks = config.num_res_blocks, len(config.ch_mult)
        self.conv_in = nn.Conv2d(config.in_channels, config.base_channels, kernel_size=(3, 3), padding=1, bias=False)
        self.down = nn.ModuleList()
        in_ch_mult = (1,) + tuple(config.ch_mult)
        block_in = block_out = config.base_channels
        for i_level in range(self.num_blocks):
            block = nn.ModuleList()
            block_in = config.base_channels * in_ch_mult[i_level]
            block_out = config.base_channels * config.ch_mult[i_level]
            for _ in range(self.num_res_blocks):
                block.append(ResBlock(block_in, block_out))
                block_in = block_out
            down = _Level()
            down.block = block
            if i_level < self.num_blocks - 1:
                down.downsample = nn.Conv2d(block_out, block_out, kernel_size=(3, 3), stride=(2, 2), padding=1)
            self.down.append(down)
        self.mid_block = nn.ModuleList([ResBlock(block_in, block_in) for _ in range(self.num_res_blocks)])
        self.norm_out = nn.GroupNorm(32, block_out, eps=1e-6)
        self.conv_out = nn.Conv2d(block_out, config.z_channels, kernel_size=(1, 1))

    forward = _holder_forward


class Decoder(nn.Module):
    """improved_model.py:124-182."""

    def __init__(self, config: VQConfig):
        super().__init__()
        self.num_blocks, self.num_res_blocks = len(config.ch_mult), config.num_res_blocks
        block_in = config.base_channels * config.ch_mult[self.num_blocks - 1]
        self.conv_in = nn.Conv2d(config.z_channels, block_in, kernel_size=(3, 3), padding=1, bias=True)
        self.mid_block = nn.ModuleList([ResBlock(block_in, block_in) for _ in range(self.num_res_blocks)])
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_blocks)):
            block = nn.ModuleList()
            block_out = config.base_channels * config.ch_mult[i_level]
            for _ in range(self.num_res_blocks):
                block.append(ResBlock(block_in, block_out))
                block_in = block_out
            up = _Level()
            up.block = block
            if i_level > 0:
                up.upsample = Upsampler(block_in)
            self.up.insert(0, up)
        self.norm_out = nn.GroupNorm(32, block_in, eps=1e-6)
        self.conv_out = nn.Conv2d(block_in, config.out_channels, kernel_size=(3, 3), padding=1)

    forward = _holder_forward


def _stream():
    return torch.cuda.current_stream().cuda_stream


def bits_from_tokens(ids: torch.LongTensor, codebook_dim: int = 18) -> torch.Tensor:
    """(n, h, w) int64 -> (n, codebook_dim, h, w) f32 in {-1,+1}, channel c = bit c (LSB first): what
    ``LFQ.get_codebook_entry(...).flip(1)`` yields (lookup_free_quantize.py:181-194, visualize.py:114-115)."""
    if not ids.is_cuda:
        raise RuntimeError("1xgpt_amd runs on the GPU only (no CPU fallback): move tokens to cuda")
    ids = ids.to(torch.int64).contiguous()
    n, h, w = ids.shape
    z = torch.empty(n, codebook_dim, h, w, dtype=torch.float32, device=ids.device)
    lib = _lib.load()
    _lib.check(lib.genie_bits_from_tokens(ids.data_ptr(), z.data_ptr(), n, h * w, codebook_dim, _stream()),
               "genie_bits_from_tokens")
    return z


def tokens_from_bits(hcode: torch.Tensor) -> torch.LongTensor:
    """Encoder output (n, bits, h, w) -> dataset-convention ids (n, h, w): bit c = [h_c > 0] (LSB first)."""
    if not hcode.is_cuda:
        raise RuntimeError("1xgpt_amd runs on the GPU only (no CPU fallback)")
    hc = hcode.float().contiguous()
    n, bits, h, w = hc.shape
    ids = torch.empty(n, h, w, dtype=torch.int64, device=hc.device)
    lib = _lib.load()
    _lib.check(lib.genie_tokens_from_bits(hc.data_ptr(), ids.data_ptr(), n, h * w, bits, _stream()),
               "genie_tokens_from_bits")
    return ids


def rescale_magvit_output(x: torch.Tensor) -> torch.Tensor:
    """[-1,1] -> uint8 [0,255], clamp then truncate, on the device (visualize.py:84-92).  bf16 input reproduces the
    reference's bf16 intermediate roundings bit for bit."""
    if not x.is_cuda:
        raise RuntimeError("1xgpt_amd runs on the GPU only (no CPU fallback)")
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    lib = _lib.load()
    if x.dtype == torch.bfloat16:
        _lib.check(lib.genie_rescale_u8_bf16(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "genie_rescale_u8")
    else:
        x = x.float()
        _lib.check(lib.genie_rescale_u8_f32(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "genie_rescale_u8")
    return out


class VQModel(nn.Module):
    """Encoder + Decoder parameters + LFQ bit packing (inference subset of models/lfqgan.py:21-133).  ``decode_tokens`` /
    ``encode_tokens`` run the hand-written conv stacks (built lazily from the current parameters; call ``refresh()`` after
    loading new weights or moving the module)."""

    def __init__(self, config: VQConfig = None):
        super().__init__()
        self.config = config or VQConfig()
        self.encoder = Encoder(self.config)
        self.decoder = Decoder(self.config)
        self.codebook_dim = int(math.log2(self.config.codebook_size))
        self.requires_grad_(False)
        self._hip_dec = self._hip_enc = None

    def refresh(self):
        self._hip_dec = self._hip_enc = None
        return self

    def _apply(self, fn, *a, **k):  # .to() / .cuda() / .half(): the packed copies are stale
        self._hip_dec = self._hip_enc = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._hip_dec = self._hip_enc = None
        return super().load_state_dict(*a, **k)

    def hip_decoder(self):
        if self._hip_dec is None:
            self._hip_dec = HipDecoder(self.decoder, self.codebook_dim)
        return self._hip_dec

    def hip_encoder(self):
        if self._hip_enc is None:
            self._hip_enc = HipEncoder(self.encoder, self.codebook_dim)
        return self._hip_enc

    @torch.no_grad()
    def decode_tokens(self, ids_nhw: torch.LongTensor) -> torch.Tensor:
        """(n, h, w) token ids -> (n, 3, H, W) uint8 on the device (visualize.py:111-121 without the host round trip)."""
        return self.hip_decoder().decode_tokens(ids_nhw)

    @torch.no_grad()
    def encode_tokens(self, frames_u8: torch.Tensor) -> torch.LongTensor:
        """(n, 3, H, W) uint8 -> (n, h, w) ids with the dataset bit convention."""
        return self.hip_encoder().encode_tokens(frames_u8)


def _check_widths(module, what):
    bad = sorted({m.out_filters for m in module.modules() if isinstance(m, ResBlock) and m.out_filters % 64} |
                 {m.in_filters for m in module.modules() if isinstance(m, ResBlock) and m.in_filters % 64})
    if bad:
        raise ValueError(f"{what}: ResBlock widths {bad} are not multiples of 64 -- the implicit-GEMM conv kernels do not "
                         "cover this geometry and there is no fallback backend")


class HipDecoder:
    """The MAGVIT2 ``Decoder`` (improved_model.py:124-182) on hand-written gfx950 kernels: NHWC bf16 activations,
    3x3 convolutions as implicit GEMMs on the bf16 matrix cores (``genie_conv3x3_bf16``: gathered A operand, fused
    bias / ResBlock skip / depth-to-space epilogue), fused GroupNorm+swish (``genie_group_norm_swish_bf16``), 1x1
    shortcuts as GEMMs, direct kernels for the two edge layers, tokens in and uint8 frames out without leaving HBM.
    Built from a ``Decoder`` module's parameters (weights are re-packed tap-major bf16 once).
    Requires every ResBlock width to be a multiple of 64 (true for the shipped config: 128/256/512)."""

    def __init__(self, decoder: "Decoder", codebook_dim: int = 18):
        self.lib = _lib.load()
        dev = next(decoder.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("1xgpt_amd runs on the GPU only (no CPU fallback): move the decoder to cuda")
        _check_widths(decoder, "HipDecoder")
        self.dev, self.bits = dev, codebook_dim
        self.zero = torch.zeros(64, dtype=torch.bfloat16, device=dev)
        self._keep = []
        pack, f32, block = self._pack, self._f32, self._block
        self.cin_pad = 64
        self.conv_in = pack(decoder.conv_in, pad_in=self.cin_pad)
        self.mid = [block(b) for b in decoder.mid_block]
        self.levels = []
        for i_level in reversed(range(decoder.num_blocks)):
            up = decoder.up[i_level]
            self.levels.append({"blocks": [block(b) for b in up.block],
                                "up": pack(up.upsample.conv1) if i_level > 0 else None})
        self.norm_out = (f32(decoder.norm_out.weight), f32(decoder.norm_out.bias))
        self.c_out = decoder.conv_out.weight.shape[0]
        self.conv_out = pack(decoder.conv_out, pad_out=(self.c_out + 7) // 8 * 8)
        torch.cuda.synchronize()

    # -- weight packing (tap-major bf16 once per model)
    def _pack(self, conv, pad_in=None, pad_out=None):
        w = conv.weight.detach().float()
        bsrc = conv.bias
        if pad_in is not None and w.shape[1] < pad_in:   # zero input channels: C_in % 64 == 0 for the implicit GEMM
            w = torch.cat([w, w.new_zeros(w.shape[0], pad_in - w.shape[1], *w.shape[2:])], 1)
        if pad_out is not None and w.shape[0] < pad_out:  # zero output channels: C_out % 8 == 0
            w = torch.cat([w, w.new_zeros(pad_out - w.shape[0], *w.shape[1:])], 0)
            if bsrc is not None:
                bsrc = torch.cat([bsrc.detach().float(), bsrc.new_zeros(pad_out - bsrc.shape[0]).float()])
        w = w.contiguous()
        co, ci, kh, kw = w.shape
        out = torch.empty(co, kh * kw, ci, dtype=torch.bfloat16, device=self.dev)
        _lib.check(self.lib.genie_pack_conv_weight(w.data_ptr(), out.data_ptr(), co, ci, kh * kw, _stream()), "pack_conv")
        b = None if bsrc is None else bsrc.detach().float().contiguous()
        self._keep += [w, out, b]
        return out, b, ci, co

    def _f32(self, p):
        t = p.detach().float().contiguous()
        self._keep.append(t)
        return t

    def _block(self, rb):
        d = {"n1": (self._f32(rb.norm1.weight), self._f32(rb.norm1.bias)),
             "n2": (self._f32(rb.norm2.weight), self._f32(rb.norm2.bias)),
             "c1": self._pack(rb.conv1), "c2": self._pack(rb.conv2), "cin": rb.in_filters, "cout": rb.out_filters}
        if rb.in_filters != rb.out_filters:
            d["nin"] = self._pack(rb.nin_shortcut)
        return d

    # -- primitive wrappers (x: (n, H, W, C) bf16 contiguous)
    def _gn(self, x, gb, swish=True):
        """GroupNorm(32, 1e-6) [+ swish].  When x came out of ``_conv3`` with fused statistics (``x._gn``: the conv's
        per-tile partial sums and geometry) no statistics pass over x runs."""
        n, H, W, C = x.shape
        y = torch.empty_like(x)
        fused = getattr(x, "_gn", None)
        if fused is not None:
            part, ch, cw, cout, d2s = fused
            stats = torch.empty(n * 32 * 2, dtype=torch.float32, device=self.dev)
            _lib.check(self.lib.genie_group_norm_swish_fused_bf16(x.data_ptr(), gb[0].data_ptr(), gb[1].data_ptr(), y.data_ptr(),
                                                                  part.data_ptr(), stats.data_ptr(), n, ch, cw, cout, int(d2s),
                                                                  32, 1e-6, int(swish), _stream()),
                       "genie_group_norm_swish_fused_bf16")
            return y
        stats = torch.empty(self.lib.genie_group_norm_scratch_floats(n, H * W, 32), dtype=torch.float32, device=self.dev)
        _lib.check(self.lib.genie_group_norm_swish_bf16(x.data_ptr(), gb[0].data_ptr(), gb[1].data_ptr(), y.data_ptr(),
                                                        stats.data_ptr(), n, H * W, C, 32, 1e-6, int(swish), _stream()),
                   "genie_group_norm_swish_bf16")
        return y

    def _conv3(self, x, wp, residual=None, d2s=False, stride=1, gn=True):
        """3x3 conv (stride 1, or 2 for the encoder's downsample; H, W below are the OUTPUT size).  gn=True asks the conv
        to also emit the GroupNorm statistics of its output (every conv here but the last feeds a GroupNorm); geometries
        the fused form does not cover (tiles that straddle images, narrow layers) take the plain conv and the separate
        statistics pass."""
        w, b, ci, co = wp
        n, Hi, Wi, C = x.shape
        assert C == ci
        H, W = Hi // stride, Wi // stride
        y = torch.empty((n, 2 * H, 2 * W, co // 4) if d2s else (n, H, W, co), dtype=torch.bfloat16, device=self.dev)
        bp = 0 if b is None else b.data_ptr()
        rp = 0 if residual is None else residual.data_ptr()
        if gn:
            part = torch.empty(self.lib.genie_conv_gn_part_floats(n, H, W, co), dtype=torch.float32, device=self.dev)
            rc = self.lib.genie_conv3x3_gn_bf16(x.data_ptr(), w.data_ptr(), bp, rp, y.data_ptr(), self.zero.data_ptr(), n, H, W,
                                                ci, co, int(d2s), stride, part.data_ptr(), 32, _stream())
            if rc == 0:
                y._gn = (part, H, W, co, d2s)
                return y
            if rc != _lib.E_UNSUPPORTED:
                _lib.check(rc, "genie_conv3x3_gn_bf16")
        if stride == 2:
            assert residual is None and not d2s
            _lib.check(self.lib.genie_conv3x3_s2_bf16(x.data_ptr(), w.data_ptr(), bp, y.data_ptr(), self.zero.data_ptr(), n, H, W,
                                                      ci, co, _stream()), "genie_conv3x3_s2_bf16")
        else:
            _lib.check(self.lib.genie_conv3x3_bf16(x.data_ptr(), w.data_ptr(), bp, rp, y.data_ptr(), self.zero.data_ptr(), n, H, W,
                                                   ci, co, int(d2s), _stream()), "genie_conv3x3_bf16")
        return y

    def _conv1(self, x, wp):
        w, b, ci, co = wp
        n, H, W, C = x.shape
        y = torch.empty(n, H, W, co, dtype=torch.bfloat16, device=self.dev)
        _lib.check(self.lib.genie_conv1x1_bf16(x.data_ptr(), w.data_ptr(), 0 if b is None else b.data_ptr(), y.data_ptr(),
                                               n * H * W, ci, co, _stream()), "genie_conv1x1_bf16")
        return y

    def _direct(self, x, wp, out_mode):
        w, b, ci, co = wp
        n, H, W, C = x.shape
        y = torch.empty((n, H, W, co), dtype=torch.bfloat16, device=self.dev) if out_mode == 0 else \
            torch.empty((n, co, H, W), dtype=torch.float32, device=self.dev)
        _lib.check(self.lib.genie_conv_direct_bf16(x.data_ptr(), w.data_ptr(), 0 if b is None else b.data_ptr(),
                                                   y.data_ptr(), n, H, W, ci, co, out_mode, _stream()),
                   "genie_conv_direct_bf16")
        return y

    def _res(self, x, blk):
        t = self._conv3(self._gn(x, blk["n1"]), blk["c1"])
        t = self._gn(t, blk["n2"])
        res = x if "nin" not in blk else self._conv1(x, blk["nin"])
        return self._conv3(t, blk["c2"], residual=res)

    @torch.no_grad()
    def decode_bits_nhwc(self, z_nhwc: torch.Tensor) -> torch.Tensor:
        """(n, h, w, 64) bf16 (+-1 bits, zero padded) -> decoder output NHWC bf16 (n, H, W, 4) (channel 3 is padding)."""
        x = self._conv3(z_nhwc, self.conv_in)
        for blk in self.mid:
            x = self._res(x, blk)
        for lvl in self.levels:
            for blk in lvl["blocks"]:
                x = self._res(x, blk)
            if lvl["up"] is not None:
                x = self._conv3(x, lvl["up"], d2s=True)
        x = self._gn(x, self.norm_out)
        return self._conv3(x, self.conv_out, gn=False)

    @torch.no_grad()
    def decode_tokens(self, ids_nhw: torch.LongTensor, return_float=False):
        """(n, h, w) token ids -> (n, 3, H, W) uint8 (truncating rescale, visualize.py:84-92)."""
        ids = ids_nhw.to(self.dev).to(torch.int64).contiguous()
        n, h, w = ids.shape
        z = torch.empty(n, h, w, self.cin_pad, dtype=torch.bfloat16, device=self.dev)
        _lib.check(self.lib.genie_bits_from_tokens_nhwc_bf16(ids.data_ptr(), z.data_ptr(), n * h * w, self.bits,
                                                             self.cin_pad, _stream()), "genie_bits_from_tokens_nhwc_bf16")
        y = self.decode_bits_nhwc(z)  # (n, H, W, cpad) bf16
        H, W, cpad = y.shape[1], y.shape[2], y.shape[3]
        if return_float:
            return y[..., :self.c_out].permute(0, 3, 1, 2).float().contiguous()
        out = torch.empty(n, self.c_out, H, W, dtype=torch.uint8, device=self.dev)
        _lib.check(self.lib.genie_rescale_u8_nhwc_bf16(y.data_ptr(), out.data_ptr(), n, H * W, cpad, self.c_out, _stream()),
                   "genie_rescale_u8_nhwc_bf16")
        return out


class HipEncoder(HipDecoder):
    """The MAGVIT2 ``Encoder`` (improved_model.py:54-121) on the same hand-written kernels: uint8 frames in, token ids
    (dataset bit convention) out.  conv_in 3->128 runs on the implicit GEMM with the input zero-padded
    to 64 channels; the stride-2 downsample convs use the strided A-gather (``genie_conv3x3_s2_bf16``); the 1x1
    conv_out 512->18 is a GEMM padded to 20 outputs whose sign bits are packed by ``genie_tokens_from_code_nhwc_bf16``."""

    def __init__(self, encoder: "Encoder", codebook_dim: int = 18):  # noqa: super().__init__ builds decoder tables
        self.lib = _lib.load()
        dev = next(encoder.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("1xgpt_amd runs on the GPU only (no CPU fallback): move the encoder to cuda")
        _check_widths(encoder, "HipEncoder")
        self.dev, self.bits = dev, codebook_dim
        self.zero = torch.zeros(64, dtype=torch.bfloat16, device=dev)
        self._keep = []
        self.cin = encoder.conv_in.weight.shape[1]
        self.cin_pad = 64
        self.conv_in = self._pack(encoder.conv_in, pad_in=self.cin_pad)
        self.levels = []
        for i_level in range(encoder.num_blocks):
            dn = encoder.down[i_level]
            self.levels.append({"blocks": [self._block(b) for b in dn.block],
                                "down": self._pack(dn.downsample) if i_level < encoder.num_blocks - 1 else None})
        self.mid = [self._block(b) for b in encoder.mid_block]
        self.norm_out = (self._f32(encoder.norm_out.weight), self._f32(encoder.norm_out.bias))
        self.code_pad = (codebook_dim + 3) // 4 * 4
        self.conv_out = self._pack(encoder.conv_out, pad_out=self.code_pad)
        torch.cuda.synchronize()

    @torch.no_grad()
    def encode_tokens(self, frames_u8: torch.Tensor) -> torch.LongTensor:
        """(n, 3, H, W) uint8 -> (n, H/16, W/16) int64 token ids."""
        f = frames_u8.to(self.dev).contiguous()
        n, c, H, W = f.shape
        x = torch.empty(n, H, W, self.cin_pad, dtype=torch.bfloat16, device=self.dev)
        _lib.check(self.lib.genie_frames_to_nhwc_bf16(f.data_ptr(), x.data_ptr(), n, H * W, c, self.cin_pad, _stream()),
                   "genie_frames_to_nhwc_bf16")
        x = self._conv3(x, self.conv_in)
        for lvl in self.levels:
            for blk in lvl["blocks"]:
                x = self._res(x, blk)
            if lvl["down"] is not None:
                x = self._conv3(x, lvl["down"], stride=2)
        for blk in self.mid:
            x = self._res(x, blk)
        x = self._gn(x, self.norm_out)
        code = self._conv1(x, self.conv_out)  # (n, h, w, code_pad) bf16
        nn_, h_, w_, _ = code.shape
        ids = torch.empty(nn_, h_, w_, dtype=torch.int64, device=self.dev)
        _lib.check(self.lib.genie_tokens_from_code_nhwc_bf16(code.data_ptr(), ids.data_ptr(), nn_ * h_ * w_, self.bits,
                                                             self.code_pad, _stream()), "genie_tokens_from_code_nhwc_bf16")
        return ids


def load_tokenizer_ckpt(model: VQModel, path: str):
    """Lightning checkpoint -> encoder.* / decoder.* (lfqgan.py:85-119; EMA == raw weights at inference)."""
    sd = torch.load(path, map_location="cpu")["state_dict"]
    own = model.state_dict()
    picked = {k: v for k, v in sd.items() if k in own}
    missing = [k for k in own if k not in picked]
    if missing:
        raise KeyError(f"tokenizer checkpoint lacks {len(missing)} keys, e.g. {missing[:3]}")
    model.load_state_dict(picked, strict=True)
    return model


def decode_latents_wrapper(batch_size=16, tokenizer_ckpt="data/magvit2.ckpt", max_images=None, model: VQModel = None,
                           device="cuda"):
    """visualize.py:95-122, device-resident: returns ``decode_latents(tokens (b,h,w)) -> uint8 (b,3,H,W)`` tensor on the
    hand-written conv stack (bf16 operands like the reference's ``.to(dtype=torch.bfloat16)`` module)."""
    if model is None:
        model = load_tokenizer_ckpt(VQModel(VQConfig()), tokenizer_ckpt)
    model = model.to(device=device).eval()
    hip_dec = model.hip_decoder()

    @torch.no_grad()
    def decode_latents(video_data):
        if isinstance(video_data, np.ndarray):
            video_data = torch.from_numpy(video_data.astype(np.int64))
        video_data = video_data.to(device)
        outs = []
        for s in range(0, video_data.shape[0], batch_size):
            outs.append(hip_dec.decode_tokens(video_data[s:s + batch_size]))
            if max_images and len(outs) * batch_size >= max_images:
                break
        return torch.cat(outs)

    return decode_latents


# ---- synthetic tokenizer weights (no magvit2.ckpt offline): PCG64 streams keyed by (seed, crc32(key)) ------------
def make_vq_state_dict(model: VQModel, seed: int = 1) -> dict:
    """{key: f32 ndarray}: conv weights N(0, 1/sqrt(fan_in)), biases N(0, 0.05), GroupNorm gamma 1 + 0.1 N, beta 0.05 N."""
    out = {}
    for k, v in model.state_dict().items():
        g = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, zlib.crc32(k.encode())])))
        z = g.standard_normal(tuple(v.shape), dtype=np.float32)
        if v.dim() == 4:
            fan_in = v.shape[1] * v.shape[2] * v.shape[3]
            w = z * np.float32(1.0 / math.sqrt(fan_in))
        elif k.endswith("weight"):  # GroupNorm gamma
            w = np.float32(1.0) + z * np.float32(0.1)
        else:
            w = z * np.float32(0.05)
        out[k] = np.ascontiguousarray(w, dtype=np.float32)
    return out
