"""STMaskGIT -- the GENIE world model (embed -> ST transformer -> factored readout) and its MaskGIT decoder.

Drop-in for the reference's ``genie.st_mask_git.STMaskGIT`` (st_mask_git.py:29-313): same constructor,
attributes (``mask_token_id``, ``config``, ``h``, ``w``), state-dict keys and methods
(``compute_logits``, ``forward``, ``compute_loss_and_acc``, ``maskgit_generate``, ``generate``,
``init_weights``, ``from_pretrained``/``save_pretrained``).  All arithmetic runs in libgenie_hip.so
(hand-written gfx950 kernels) on the module's device; PyTorch only owns the buffers and the stream.
There is no CPU fallback: calling a method with CPU tensors raises.

Differences that are deliberate and visible:
  * ``maskgit_generate`` takes two optional extras, ``noise`` (the U[0,1) draws the reference takes from
    ``torch.rand_like``, :206) and ``uniforms`` (for temperature > 0), so a test can replay an exact stream.
  * ``precision``: "exact" (f32 MFMA), "f16x3" (split-f16 operands on the f16 matrix cores, f32-class
    results: the parity-grade fast mode) or "bf16" (bf16 MFMA operands, f32 accumulate: throughput mode).
"""
import json
import math
import os

import torch
import torch.nn as nn

from . import _lib
from .config import GenieConfig
from .factorization_utils import FactorizedEmbedding
from .st_transformer import STTransformerDecoder

_PRECISIONS = {"exact": _lib.PREC_EXACT, "f32": _lib.PREC_EXACT, "bf16": _lib.PREC_BF16, "fast": _lib.PREC_BF16,
               "f16x3": _lib.PREC_F16X3}


def cosine_schedule(u):
    """Fraction of tokens still masked after a share ``u`` of the MaskGIT steps: cos(pi u / 2), for a Python float
    (the decoder's own use) or a tensor; anything else is refused like the reference does (st_mask_git.py:17-26)."""
    if isinstance(u, float):
        return math.cos(0.5 * math.pi * u)
    if isinstance(u, torch.Tensor):
        return torch.cos(u * (0.5 * torch.pi))
    raise NotImplementedError(f"Unexpected {type(u)=} {u=}")


class _Packer:
    """16-bit copies of the Linear weights in the layout the kernels of one precision read, made once per weight table.

    ``packer(weight)`` returns the device pointer of the packed copy (bf16, or the [hi | lo] f16 planes of f16x3);
    ``is_wide(ptr)`` tells whether that f16x3 tensor breaks the |w| < 32 range contract of the fast split GEMM (the caller
    writes the answer into the ``w16_wide`` field next to the pointer: the flag travels with the table, the library keeps no
    registry); ``temporal_fused`` / ``mlp_fused`` build the fragment streams of the fused sub-block kernels
    (csrc/kernels_fused.hip) for the geometry they cover and return 0 otherwise."""

    def __init__(self, lib, prec, dev, config):
        self.lib, self.prec, self.dev, self.config = lib, prec, dev, config
        self.keep, self.wide = [], set()

    def _stream(self):
        return torch.cuda.current_stream().cuda_stream

    def __call__(self, wt):
        if self.prec == _lib.PREC_BF16:
            t = torch.empty(wt.shape, dtype=torch.bfloat16, device=self.dev)
            _lib.check(self.lib.genie_pack_bf16(wt.data_ptr(), t.data_ptr(), wt.numel(), self._stream()), "genie_pack_bf16")
        else:  # f16x3: [hi plane | lo plane], wt ~ hi + lo / 2048
            t = torch.empty((2,) + tuple(wt.shape), dtype=torch.float16, device=self.dev)
            _lib.check(self.lib.genie_pack_split_f16(wt.data_ptr(), t.data_ptr(), wt.numel(), self._stream()),
                       "genie_pack_split_f16")
            # Range contract, checked on the packed hi plane at load time.  Beyond the f16 range the precision cannot represent
            # the tensor at all; from 32 up the 256x256 GEMM's 2^11 scaling of the hi plane would overflow, so the tensor is
            # flagged and its Linear runs on the two-accumulator kernels (same f32-class result, slower for that tensor only).
            if not bool(torch.isfinite(t[0]).all()):
                raise ValueError("f16x3 precision needs finite |weight| < 65504 (the f16 range of the split's hi plane); this "
                                 "checkpoint has a larger or non-finite weight -- use precision='exact' or 'bf16'")
            if float(t[0].abs().max()) >= 31.98:
                import warnings
                self.wide.add(t.data_ptr())
                warnings.warn(f"f16x3: a weight tensor of shape {tuple(wt.shape)} has |w| >= 32 (max {float(wt.abs().max()):.3g}): "
                              "its Linear runs on the two-accumulator split GEMM instead of the 256x256 kernel (same results, "
                              "lower throughput for that layer)")
        self.keep.append(t)
        return t.data_ptr()

    def is_wide(self, ptr):
        return ptr in self.wide

    def _fused_geometry(self):
        c = self.config
        return (self.prec == _lib.PREC_BF16 and c.d_model == 256 and c.num_heads == 8 and not c.qk_norm
                and os.environ.get("GENIE_NO_FUSED", "0") != "1")

    def temporal_fused(self, attn):
        c = self.config
        if (self.prec == _lib.PREC_F16X3 and c.d_model == 256 and c.num_heads == 8 and not c.qk_norm and c.T <= 16
                and os.environ.get("GENIE_NO_FUSED", "0") != "1"):
            # f16x3: the temporal qkv Linear + attention run as one kernel (csrc/kernels_fused_f16x3.hip) on this split-f16 stream
            t = torch.empty(_lib.TEMPORAL_QKV_F16X3_ELEMS, dtype=torch.float16, device=self.dev)
            _lib.check(self.lib.genie_pack_temporal_qkv_f16x3(attn.qkv.weight.data_ptr(), t.data_ptr(), self._stream()),
                       "genie_pack_temporal_qkv_f16x3")
            self.keep.append(t)
            return t.data_ptr()
        if not (self._fused_geometry() and self.config.T == 16):
            return 0
        t = torch.empty(_lib.TEMPORAL_FUSED_ELEMS, dtype=torch.bfloat16, device=self.dev)
        _lib.check(self.lib.genie_pack_temporal_fused_bf16(attn.qkv.weight.data_ptr(), attn.proj.weight.data_ptr(), t.data_ptr(),
                                                           self._stream()), "genie_pack_temporal_fused_bf16")
        self.keep.append(t)
        return t.data_ptr()

    def spatial_fused(self, attn):
        if not (self._fused_geometry() and self.config.S == 256):
            return 0
        # [out-projection stream | qkv stream]: the second part lets the previous block's fused MLP kernel write this block's operand planes
        t = torch.empty(_lib.SPATIAL_PROJ_FUSED_ELEMS + _lib.SPATIAL_QKV_FUSED_ELEMS, dtype=torch.bfloat16, device=self.dev)
        _lib.check(self.lib.genie_pack_spatial_proj_fused_bf16(attn.proj.weight.data_ptr(), t.data_ptr(), self._stream()),
                   "genie_pack_spatial_proj_fused_bf16")
        _lib.check(self.lib.genie_pack_spatial_qkv_fused_bf16(attn.qkv.weight.data_ptr(), t.data_ptr() + 2 * _lib.SPATIAL_PROJ_FUSED_ELEMS,
                                                              self._stream()), "genie_pack_spatial_qkv_fused_bf16")
        self.keep.append(t)
        return t.data_ptr()

    def _frame_geometry(self):
        # the fragment-order kernels of the one-frame passes (csrc/kernels_frame.hip): f16x3, LayerNorm or qk-norm blocks, heads of 64 or 32
        c = self.config
        return (self.prec == _lib.PREC_F16X3 and c.S == 256 and c.d_model in (128, 256, 512)
                and c.d_model in (64 * c.num_heads, 32 * c.num_heads) and os.environ.get("GENIE_NO_FRAME_KERNELS", "0") != "1")

    def frame_stream(self, *weights):
        """The given Linear weights back to back as split f16 in fragment order (genie_pack_frame_w16), or 0."""
        if not self._frame_geometry() or any(w.shape[0] % 64 or w.shape[1] % 64 for w in weights):
            return 0   # (output widths in 64-column tiles -- no tail handling in gemm16_fr --, contraction in 64-k blocks)
        t = torch.empty(sum(2 * w.numel() for w in weights), dtype=torch.float16, device=self.dev)
        off = 0
        for w in weights:
            _lib.check(self.lib.genie_pack_frame_w16(w.data_ptr(), t.data_ptr() + 2 * off, w.shape[0], w.shape[1], self._stream()),
                       "genie_pack_frame_w16")
            off += 2 * w.numel()
        self.keep.append(t)
        return t.data_ptr()

    def mlp_fused(self, mlp):
        if not (self._fused_geometry() and mlp.fc1.weight.shape[0] == 1024):
            return 0
        t = torch.empty(_lib.MLP_FUSED_ELEMS, dtype=torch.bfloat16, device=self.dev)
        _lib.check(self.lib.genie_pack_mlp_fused_bf16(mlp.fc1.weight.data_ptr(), mlp.fc2.weight.data_ptr(), t.data_ptr(),
                                                      self._stream()), "genie_pack_mlp_fused_bf16")
        self.keep.append(t)
        return t.data_ptr()


class GenieOutput(dict):
    """Minimal stand-in for transformers' ModelOutput: attribute and key access to loss / acc / logits."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class STMaskGIT(nn.Module):
    def __init__(self, config: GenieConfig, precision: str = "exact"):
        super().__init__()
        self.h = self.w = math.isqrt(config.S)
        assert self.h ** 2 == config.S, "Expected S to be square"
        self.decoder = STTransformerDecoder(
            num_layers=config.num_layers, num_heads=config.num_heads, d_model=config.d_model,
            qkv_bias=config.qkv_bias, proj_bias=config.proj_bias, qk_norm=config.qk_norm, use_mup=config.use_mup,
            attn_drop=config.attn_drop, mlp_ratio=config.mlp_ratio, mlp_bias=config.mlp_bias,
            mlp_drop=config.mlp_drop)
        self.pos_embed_TSC = nn.Parameter(torch.zeros(1, config.T, config.S, config.d_model))
        self.mask_token_id = config.image_vocab_size
        self.token_embed = FactorizedEmbedding(
            factored_vocab_size=config.factored_vocab_size, num_factored_vocabs=config.num_factored_vocabs,
            d_model=config.d_model, mask_token_id=self.mask_token_id)
        # nn.Linear holds the readout parameters for both the plain and the muP readout; the muP factor
        # output_mult/width_mult (reference :316-323) is applied inside the readout GEMM (cfg.readout_mult).
        self.out_x_proj = nn.Linear(config.d_model, config.factored_vocab_size * config.num_factored_vocabs)
        self.config = config
        self._table = None
        self._wide = []
        self.set_precision(precision)
        self._ws = None
        self.requires_grad_(False)

    # ------------------------------------------------------------------ plumbing
    def set_precision(self, precision: str):
        if precision not in _PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}, got {precision!r}")
        self.precision = precision
        self._prec = _PRECISIONS[precision]
        self._invalidate()
        return self

    def _apply(self, fn, *a, **k):  # .to()/.cuda()/.float() move parameters: cached pointers are stale
        self._invalidate()
        self._ws = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._invalidate()
        return super().load_state_dict(*a, **k)

    def _device(self):
        dev = self.pos_embed_TSC.device
        if dev.type != "cuda":
            raise RuntimeError("1xgpt_amd.STMaskGIT runs on the GPU only (no CPU fallback): call .to('cuda') first")
        return dev

    def refresh_weights(self):
        """Rebuild the device pointer table (and bf16 copies).  Call after mutating parameters in place."""
        self._invalidate()

    def _invalidate(self):
        """Forget the cached pointer table (packed copies, fused streams and range flags go with it)."""
        self._table = None
        self._wide = []

    def _weights(self):
        if self._table is not None:
            return self._table
        dev = self._device()
        lib = _lib.load()
        for n, p in self.named_parameters():
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError(f"parameter {n} must be contiguous float32 (got {p.dtype})")
        cfg = _lib.make_cfg(self.config, self._prec)
        _lib.check(lib.genie_check_config(cfg), "genie_check_config")
        packed = _Packer(lib, self._prec, dev, self.config) if self._prec != _lib.PREC_EXACT else None
        L = self.config.num_layers
        layers = (_lib.LayerWeights * L)(*[blk.layer_struct(packed) for blk in self.decoder.layers])
        w = _lib.Weights()
        w.pos_embed = self.pos_embed_TSC.data_ptr()
        w.mask_embed = self.token_embed.mask_token_embed.data_ptr()
        for j, e in enumerate(self.token_embed.factored_embeds):
            w.embed[j] = e.weight.data_ptr()
        w.out_w, w.out_b = self.out_x_proj.weight.data_ptr(), self.out_x_proj.bias.data_ptr()
        if packed is not None:
            w.out_w16 = packed(self.out_x_proj.weight)
            w.out_w16_wide = int(packed.is_wide(w.out_w16))
            w.out_frame_w16 = packed.frame_stream(self.out_x_proj.weight)
        w.layers_host = layers
        self._table = (cfg, w, layers, packed)
        self._wide = sorted(packed.wide) if packed is not None else []
        return self._table

    def _workspace(self, B, generate_prompt_frames=0):
        """The model's workspace for B clips; generate_prompt_frames = P > 0: large enough for genie_generate_cached with P prompt frames too."""
        cfg = self._weights()[0]
        need = _lib.load().genie_workspace_bytes(cfg, B)
        if generate_prompt_frames:
            need = max(need, _lib.load().genie_generate_workspace_bytes(cfg, B, generate_prompt_frames))
        if self._ws is None or self._ws.numel() < need or self._ws.device != self._device():
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self._device())
        return self._ws

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def _ids(self, t, whole_clips=False):
        if not t.is_cuda:
            raise RuntimeError("1xgpt_amd runs on the GPU only (no CPU fallback): move inputs to cuda")
        if whole_clips and (t.dim() < 2 or t.numel() != t.shape[0] * self.config.T * self.config.S):
            # (the reference fails on such an input too -- the positional table does not broadcast, st_mask_git.py:261 --; the
            # library takes B from dim 0 and would read past the tensor)
            raise RuntimeError(f"expected token ids of {self.config.T} frames x {self.config.S} tokens per clip (B, T, H, W) / "
                               f"(B, T*S), got {tuple(t.shape)}")
        return t.to(torch.int64).contiguous()

    # ------------------------------------------------------------------ forward pieces
    def hidden_states(self, x_THW: torch.LongTensor):
        """Run embed + decoder; the (B,T,S,d) result stays at offset 0 of the workspace (returned as a view)."""
        lib = _lib.load()
        cfg, w = self._weights()[:2]
        ids = self._ids(x_THW, whole_clips=True)
        B = ids.shape[0]
        ws = self._workspace(B)
        _lib.check(lib.genie_compute_logits(cfg, w, ids.data_ptr(), B, 0, 0, 0, ws.data_ptr(), ws.data_ptr(),
                                            ws.numel(), self._stream()), "genie_compute_logits(hidden)")
        n = B * self.config.T * self.config.S * self.config.d_model
        return ws[: n * 4].view(torch.float32).view(B, self.config.T, self.config.S, self.config.d_model)

    def compute_logits_frames(self, x_THW, t0, t1, layout="bcthw"):
        """Logits of frames [t0,t1): (B, V, t1-t0, H, W) for layout='bcthw', (B, t1-t0, S, V) for 'token'."""
        lib = _lib.load()
        cfg, w = self._weights()[:2]
        ids = self._ids(x_THW, whole_clips=True)
        B = ids.shape[0]
        ws = self._workspace(B)
        nt, V = t1 - t0, self.config.factored_vocab_size * self.config.num_factored_vocabs
        if layout == "bcthw":
            out = torch.empty(B, V, nt, self.h, self.w, dtype=torch.float32, device=ids.device)
            lay = _lib.LAYOUT_BCTHW
        else:
            out = torch.empty(B, nt, self.config.S, V, dtype=torch.float32, device=ids.device)
            lay = _lib.LAYOUT_TOKEN_MAJOR
        _lib.check(lib.genie_compute_logits(cfg, w, ids.data_ptr(), B, t0, t1, lay, out.data_ptr(), ws.data_ptr(),
                                            ws.numel(), self._stream()), "genie_compute_logits")
        return out

    def compute_logits(self, x_THW):
        """ids (B,T,H,W) -> logits (B, V, T, H, W), channels [vocab0(512) | vocab1(512)]  (reference :255-265)."""
        return self.compute_logits_frames(x_THW, 0, self.config.T, "bcthw")

    def ce_sums(self, x_THW, labels_THW, t0=1, t1=None, masked_only=True):
        """Fused forward + readout + factored CE: returns a (3,) float64 device tensor
        [sum CE, sum all-factors-correct, n counted] over frames [t0,t1); no logits leave the workspace."""
        lib = _lib.load()
        cfg, w = self._weights()[:2]
        ids, lab = self._ids(x_THW, whole_clips=True), self._ids(labels_THW, whole_clips=True)
        B = ids.shape[0]
        t1 = self.config.T if t1 is None else t1
        ws = self._workspace(B)
        self.hidden_states(ids)
        sums = torch.zeros(3, dtype=torch.float64, device=ids.device)
        _lib.check(lib.genie_readout_ce(cfg, w, ws.data_ptr(), lab.data_ptr(), ids.data_ptr() if masked_only else 0,
                                        B, t0, t1, sums.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()),
                   "genie_readout_ce")
        return sums

    def compute_loss_and_acc(self, logits_CTHW, targets_THW, relevant_mask_THW):
        """Masked mean of the summed factored CE and of 'all factors argmax-correct' (reference :231-253)."""
        lib = _lib.load()
        cfg = self._weights()[0]
        B, T = targets_THW.shape[0], self.config.T
        lg = logits_CTHW[:, :, 1:].contiguous().float()
        tg = self._ids(targets_THW).view(B, T, -1)
        wid = torch.zeros_like(tg)
        wid[:, 1:] = torch.where(relevant_mask_THW.reshape(B, T - 1, -1).bool(), self.mask_token_id, 0)
        sums = torch.zeros(3, dtype=torch.float64, device=lg.device)
        _lib.check(lib.genie_factored_ce(cfg, lg.data_ptr(), _lib.LAYOUT_BCTHW, tg.data_ptr(), wid.data_ptr(), B, 1, T,
                                         sums.data_ptr(), self._stream()), "genie_factored_ce")
        return (sums[0] / sums[2]).float(), (sums[1] / sums[2]).float()  # 0/0 -> nan like the reference

    def forward(self, input_ids, labels):
        """(B, T*H*W) ids + labels -> GenieOutput(loss, acc, logits (B,V,T,H,W))  (reference :267-279)."""
        T, H, W = self.config.T, self.h, self.w
        x_THW = self._ids(input_ids).view(-1, T, H, W)
        lab = self._ids(labels).view(-1, T, H, W)
        if lab.shape != x_THW.shape:
            raise RuntimeError(f"labels {tuple(labels.shape)} do not match input_ids {tuple(input_ids.shape)}")
        logits = self.compute_logits(x_THW)  # leaves the hidden state in the workspace
        lib = _lib.load()
        cfg, w = self._weights()[:2]
        B = x_THW.shape[0]
        ws = self._workspace(B)
        sums = torch.zeros(3, dtype=torch.float64, device=x_THW.device)
        _lib.check(lib.genie_readout_ce(cfg, w, ws.data_ptr(), lab.data_ptr(), x_THW.data_ptr(), B, 1, T,
                                        sums.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()),
                   "genie_readout_ce")
        loss, acc = (sums[0] / sums[2]).float(), (sums[1] / sums[2]).float()
        return GenieOutput(loss=loss, acc=acc, logits=logits)

    # ------------------------------------------------------------------ MaskGIT
    @staticmethod
    def init_mask(prompt_THW):
        H, W = prompt_THW.size(2), prompt_THW.size(3)
        return torch.zeros(prompt_THW.size(0), H * W, dtype=torch.bool, device=prompt_THW.device)

    @torch.no_grad()
    def maskgit_generate(self, prompt_THW, out_t, maskgit_steps=1, temperature=0.0, unmask_mode="random",
                         noise=None, uniforms=None, return_logits=True, check=True):
        """MaskGIT decode of frame ``out_t`` (reference :123-229): the whole loop runs on the device.

        Mutates ``prompt_THW[:, out_t]`` in place (reference :223) and returns
        ``(samples (B,H,W) int64, step-0 factored logits (B, 512, 2, H, W))``.
        noise: optional (maskgit_steps-1, B, S) float32 draws for "random" unmasking (default: torch.rand).
        uniforms: (maskgit_steps, num_factored_vocabs, B, S) for temperature > 1e-8 (default: torch.rand).
        """
        if unmask_mode not in ("greedy", "random"):
            raise NotImplementedError(f"Expected `unmask_mode` to be one of ['greedy', 'random'], got {unmask_mode}")
        assert out_t, "maskgit_generate requires out_t > 0"
        lib = _lib.load()
        cfg, w = self._weights()[:2]
        if not prompt_THW.is_cuda:
            raise RuntimeError("1xgpt_amd runs on the GPU only (no CPU fallback): move inputs to cuda")
        in_place = prompt_THW.dtype == torch.int64 and prompt_THW.is_contiguous()
        prompt = prompt_THW if in_place else prompt_THW.to(torch.int64).contiguous()
        B, T = prompt.shape[0], prompt.shape[1]
        if T != self.config.T or prompt.numel() != B * T * self.config.S:
            # (the reference fails here too: `x_TSC + self.pos_embed_TSC` does not broadcast, st_mask_git.py:261; the library would
            # read past the tensor)
            raise RuntimeError(f"maskgit_generate expects a (B, T={self.config.T}, H, W) prompt, got {tuple(prompt_THW.shape)}")
        S, V = self.config.S, self.config.factored_vocab_size * self.config.num_factored_vocabs
        dev = prompt.device
        ws = self._workspace(B)
        if unmask_mode == "random" and maskgit_steps > 1:
            if noise is None:
                noise = torch.rand(maskgit_steps - 1, B, S, dtype=torch.float32, device=dev)
            noise = noise.to(device=dev, dtype=torch.float32).contiguous()
            assert noise.numel() == (maskgit_steps - 1) * B * S, "noise must be (maskgit_steps-1, B, S)"
        else:
            noise = None
        if temperature > 1e-8:
            if uniforms is None:
                uniforms = torch.rand(maskgit_steps, self.config.num_factored_vocabs, B, S, device=dev)
            uniforms = uniforms.to(device=dev, dtype=torch.float32).contiguous()
        else:
            uniforms = None
        samples = torch.empty(B, self.h, self.w, dtype=torch.int64, device=dev)
        logits0 = torch.empty(B, V, self.h, self.w, dtype=torch.float32, device=dev) if return_logits else None
        status = torch.zeros(1, dtype=torch.int32, device=dev) if check else None
        rc = lib.genie_maskgit_generate(
            cfg, w, prompt.data_ptr(), B, int(out_t), int(maskgit_steps), float(temperature),
            _lib.UNMASK_GREEDY if unmask_mode == "greedy" else _lib.UNMASK_RANDOM,
            0 if noise is None else noise.data_ptr(), 0 if uniforms is None else uniforms.data_ptr(),
            samples.data_ptr(), 0 if logits0 is None else logits0.data_ptr(), _lib.LAYOUT_BCTHW,
            0 if status is None else status.data_ptr(), ws.data_ptr(), ws.numel(), self._stream())
        _lib.check(rc, "genie_maskgit_generate")
        if check and int(status.item()) != 0:  # host sync, like the reference's assert torch.all(...) (:155)
            raise AssertionError(f"when generating z{out_t}, frames {out_t} and later must be masked")
        if not in_place:
            prompt_THW.copy_(prompt.view_as(prompt_THW))
        if logits0 is None:
            return samples, None
        nv, vf = self.config.num_factored_vocabs, self.config.factored_vocab_size
        # "B (num_vocabs vocab_size) H W -> B vocab_size num_vocabs H W"  (reference :226-229)
        return samples, logits0.view(B, nv, vf, self.h, self.w).permute(0, 2, 1, 3, 4)

    def generate(self, input_ids, attention_mask=None, max_new_tokens=None, min_new_tokens=None, return_logits=False,
                 maskgit_steps=1, temperature=0.0, noise=None, kv_cache=True):
        """Autoregressive frame generation behind the reference's Llama-style signature (st_mask_git.py:65-113):
        ``input_ids`` (B, n_prompt_frames * S) holds the prompt frames; ``max_new_tokens // S`` further frames are decoded one
        after the other with ``maskgit_generate``, each seeing every frame before it.  Returns the (B, (n_prompt + n_new) * S)
        token ids, plus -- with ``return_logits`` -- the step-0 factored logits of the new frames stacked on dim 3.
        ``attention_mask`` is accepted and ignored, as in the reference.  ``min_new_tokens`` may only repeat ``max_new_tokens``.
        noise: optional (n_new_frames, maskgit_steps - 1, B, S) unmasking draws to replay (the reference draws them itself).
        kv_cache: True (default) = the frames are decoded by one-frame passes against a temporal KV cache; False = the reference's
        own schedule, a full forward over the canvas per MaskGIT step (same frames up to f32 accumulation order)."""
        S = self.config.S
        if min_new_tokens is not None and min_new_tokens != max_new_tokens:
            raise AssertionError("Expecting `min_new_tokens`, if specified, to match `max_new_tokens`.")
        if max_new_tokens % S:
            raise AssertionError("Expecting `max_new_tokens` to be a multiple of `self.config.S`.")
        ids = self._ids(input_ids)
        B, n_new = ids.size(0), max_new_tokens // S
        n_prompt = ids.numel() // (B * S)
        if kv_cache and n_new >= 1 and n_prompt >= 1 and n_prompt + n_new <= self.config.T:
            # the same frames on the temporal KV cache, the whole loop one library call (genie_generate_cached): every MaskGIT step
            # runs the rows of the frame being decoded instead of a full forward over the canvas (causal in time: the all-MASK
            # frames behind it never reach it).  Equal to the loop below up to f32 accumulation order.
            lib = _lib.load()
            cfg, w = self._weights()[:2]
            dev = ids.device
            steps = int(maskgit_steps)
            V = self.config.factored_vocab_size * self.config.num_factored_vocabs
            clip = torch.full((B, n_prompt + n_new, S), self.mask_token_id, dtype=torch.int64, device=dev)
            clip[:, :n_prompt] = ids.view(B, n_prompt, S)
            nz = None
            if steps > 1:   # torch.rand_like of st_mask_git.py:204-206: the caller's draws, or fresh ones
                nz = (torch.rand(n_new, steps - 1, B, S, device=dev) if noise is None
                      else noise.to(dev)[:, :steps - 1].reshape(n_new, steps - 1, B, S).float().contiguous())
            uni = torch.rand(n_new, steps, self.config.num_factored_vocabs, B, S, device=dev) if temperature > 1e-8 else None
            gen = torch.empty(B, n_new, S, dtype=torch.int64, device=dev)
            lg0 = torch.empty(B, n_new, S, V, dtype=torch.float32, device=dev) if return_logits else None
            nbytes = lib.genie_prefix_cache_bytes(cfg, B)
            cache = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            ws = self._workspace(B, generate_prompt_frames=n_prompt)
            _lib.check(lib.genie_generate_cached(cfg, w, clip.data_ptr(), B, n_prompt, n_new, steps, float(temperature), _lib.UNMASK_RANDOM,
                                                 0 if nz is None else nz.data_ptr(), 0 if uni is None else uni.data_ptr(), 0, 1,
                                                 gen.data_ptr(), 0 if lg0 is None else lg0.data_ptr(), cache.data_ptr(), nbytes,
                                                 ws.data_ptr(), ws.numel(), self._stream()), "genie_generate_cached")
            tokens = torch.cat([ids.view(B, n_prompt * S), gen.view(B, n_new * S)], dim=1)
            if not return_logits:
                return tokens
            nv, vf = self.config.num_factored_vocabs, self.config.factored_vocab_size
            # (B, n, S, [vocab0 | vocab1]) -> "B vocab_size num_vocabs n H W": the per-frame (B, 512, 2, H, W) logits stacked on dim 3
            return tokens, lg0.view(B, n_new, self.h, self.w, nv, vf).permute(0, 5, 4, 1, 2, 3)
        # one (B, n_prompt + n_new, H, W) canvas: prompt frames up front, the frames to come all-MASK; maskgit_generate fills
        # canvas[:, t] in place, which is exactly what the next frame's context must contain
        canvas = torch.full((B, n_prompt + n_new, self.h, self.w), self.mask_token_id, dtype=torch.long, device=ids.device)
        canvas[:, :n_prompt] = ids.view(B, n_prompt, self.h, self.w)
        step0_logits = []
        for k in range(n_new):
            frame, logits = self.maskgit_generate(canvas, n_prompt + k, maskgit_steps=maskgit_steps, temperature=temperature,
                                                  noise=None if noise is None else noise[k], return_logits=return_logits)
            canvas[:, n_prompt + k] = frame
            step0_logits.append(logits)
        tokens = canvas.view(B, -1)
        return (tokens, torch.stack(step0_logits, dim=3)) if return_logits else tokens

    # ------------------------------------------------------------------ weights
    def init_weights(self):
        """N(0, 0.02) Linear/Embedding weights, zero biases (reference :281-296; the muP branch of the
        reference needs the un-vendored ``mup`` package and is not reproduced)."""
        std = 0.02
        for module in self.modules():
            if isinstance(module, nn.Linear):
                module.weight.data.normal_(mean=0.0, std=std)
                if module.bias is not None:
                    module.bias.data.zero_()
            elif isinstance(module, nn.Embedding):
                module.weight.data.normal_(mean=0.0, std=std)
        self._invalidate()

    def set_mup_shapes(self, rescale_params=False):
        """The reference attaches muP infshapes for training (:298-304).  At inference the only effect is the
        readout factor output_mult/width_mult with base width 256, which cfg.readout_mult already carries."""
        return None

    def save_pretrained(self, save_directory):
        """config.json (flat GenieConfig dict) + model.safetensors, the layout the reference's
        PyTorchModelHubMixin writes."""
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, "config.json"), "w") as f:
            json.dump(vars(self.config), f)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()},
                  os.path.join(save_directory, "model.safetensors"))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, precision="exact", **kwargs):
        """Load a local HF-style checkpoint directory (hub ids need a network and are not supported offline)."""
        from safetensors.torch import load_file
        d = str(pretrained_model_name_or_path)
        if not os.path.isdir(d):
            raise FileNotFoundError(f"{d} is not a local checkpoint directory (no network access for hub ids)")
        with open(os.path.join(d, "config.json")) as f:
            raw = json.load(f)
        raw = raw.get("config", raw)
        from dataclasses import fields
        known = {f.name for f in fields(GenieConfig)}
        config = GenieConfig(**{k: v for k, v in raw.items() if k in known})
        if config.use_mup:
            import warnings
            warnings.warn("use_mup=True: the readout factor follows mup's documented formula output_mult * x / width_mult "
                          "(base width 256); the reference's own mup fork is not vendored, so this factor is not pinned "
                          "against reference outputs (DESIGN.md section 7, 'parity unpinned')")
        model = cls(config, precision=precision)
        model.load_state_dict(load_file(os.path.join(d, "model.safetensors")), strict=True)
        model.eval()
        return model

    def load_numpy_state_dict(self, sd: dict):
        """Load {key: ndarray} (e.g. from 1xgpt_amd.synthetic.make_state_dict)."""
        self.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        return self


class FixedMuReadout(nn.Linear):
    """Name kept for import compatibility (reference :316-323): y = Linear(output_mult * x / width_mult).
    Inside STMaskGIT the factor is fused into the readout GEMM; this standalone module applies it through
    the same HIP GEMM."""
    output_mult = 1.0

    def width_mult(self):
        return self.in_features / 256

    def forward(self, x):
        from .attention import hip_linear
        shp = x.shape
        x2 = (x.contiguous().view(-1, shp[-1]).float() * (self.output_mult / self.width_mult()))
        return hip_linear(x2, self.weight, self.bias).view(*shp[:-1], -1)
