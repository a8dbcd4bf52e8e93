"""Factored vocabulary helpers: 2^18 ids <-> (id % 512, id // 512).

Host-side mirror of the reference's genie/factorization_utils.py (same function names and argument
meaning).  The device-side arithmetic of the embedding lookup lives in csrc/kernels_exact.hip
(``embed_kernel``); these torch helpers are index bookkeeping used by callers and tests.
"""
import torch
import torch.nn as nn

from . import _lib


def factorize_token_ids(token_ids: torch.LongTensor, num_factored_vocabs: int = 2,
                        factored_vocab_size: int = 512) -> torch.LongTensor:
    """(...,) -> (..., num_factored_vocabs), factor j in [0, factored_vocab_size)  (reference :55-68)."""
    powers = factored_vocab_size ** torch.arange(num_factored_vocabs, device=token_ids.device)
    return (token_ids.unsqueeze(-1) // powers) % factored_vocab_size


def unfactorize_token_ids(factored_token_ids: torch.LongTensor, num_factored_vocabs: int = 2,
                          factored_vocab_size: int = 512) -> torch.LongTensor:
    """Inverse of factorize_token_ids (reference :71-84)."""
    powers = factored_vocab_size ** torch.arange(num_factored_vocabs, device=factored_token_ids.device)
    return (factored_token_ids * powers).sum(dim=-1)


def factorize_labels(labels_THW: torch.LongTensor, num_factored_vocabs: int = 2,
                     factored_vocab_size: int = 512) -> torch.LongTensor:
    """(B,T,H,W) -> (B, num_factored_vocabs, T, H, W)  (reference :87-100)."""
    return factorize_token_ids(labels_THW, num_factored_vocabs, factored_vocab_size).permute(0, 4, 1, 2, 3)


class FactorizedEmbedding(nn.Module):
    """Sum of the per-factor embeddings, a separate learned vector for the mask token (reference :6-52).

    Parameter names match the reference state dict (``factored_embeds.{j}.weight``, ``mask_token_embed``).
    ``forward`` runs the HIP gather kernel (without the positional term).
    """

    def __init__(self, factored_vocab_size: int, num_factored_vocabs: int, d_model: int, mask_token_id: int):
        super().__init__()
        self.factored_vocab_size = factored_vocab_size
        self.num_factored_vocabs = num_factored_vocabs
        self.d_model = d_model
        self.mask_token_id = mask_token_id
        self.factored_embeds = nn.ModuleList([nn.Embedding(factored_vocab_size, d_model)
                                              for _ in range(num_factored_vocabs)])
        self.mask_token_embed = nn.Parameter(torch.zeros(1, d_model))

    def forward(self, input_ids: torch.LongTensor) -> torch.FloatTensor:
        """ids (B, T, H*W) -> (B, T, H*W, d_model)."""
        if not input_ids.is_cuda:
            raise RuntimeError("1xgpt_amd runs on the GPU only (no CPU fallback): move inputs to cuda")
        lib = _lib.load()
        B, T, S = input_ids.shape
        ids = input_ids.contiguous()
        out = torch.empty(B, T, S, self.d_model, dtype=torch.float32, device=ids.device)
        zero_pos = torch.zeros(T * S, self.d_model, dtype=torch.float32, device=ids.device)
        cfg = _lib.GenieCfg(num_layers=1, num_heads=1, head_dim=16, d_model=self.d_model, T=T, S=S, hidden=16,
                            factored_vocab=self.factored_vocab_size, num_factored=self.num_factored_vocabs,
                            image_vocab_size=self.mask_token_id, precision=_lib.PREC_EXACT)
        w = _lib.Weights()
        w.pos_embed = zero_pos.data_ptr()
        w.mask_embed = self.mask_token_embed.data_ptr()
        for j, e in enumerate(self.factored_embeds):
            w.embed[j] = e.weight.data_ptr()
        # embed only checks d_model/T/S/vocab fields
        cfg.head_dim = 16 if self.d_model % 16 == 0 else self.d_model
        cfg.num_heads = self.d_model // cfg.head_dim
        _lib.check(lib.genie_embed(cfg, w, ids.data_ptr(), B, out.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream), "genie_embed")
        return out
