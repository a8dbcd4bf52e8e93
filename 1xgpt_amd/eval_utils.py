"""Metric helpers of the evaluation harness -- mirrors of the reference's eval_utils.py.

``compute_loss`` is the challenge metric (eval_utils.py:44-77): sum of the two 512-way cross-entropies,
plain mean over B*(T-1)*H*W, returned as a Python float.  The reduction runs in the HIP factored-CE
kernel; LPIPS (eval_utils.py:80-88) is out of scope.
"""
import torch

from . import _lib


class AvgMetric:
    """Weighted running mean with the reference's interface (eval_utils.py:10-25): ``update(val, batch_size)`` adds a batch mean
    with its weight, ``update_list(values)`` adds per-sample values with weight one each, ``mean()`` is sum / weight.  Kept as
    (weighted sum, weight) so that partial sums from several ranks can be added before dividing (distributed.means_from_sums)."""

    def __init__(self):
        self.total, self.count = 0, 0

    def _add(self, weighted_sum, weight):
        self.total, self.count = self.total + weighted_sum, self.count + weight

    def update(self, val, batch_size=1):
        self._add(val * batch_size, batch_size)

    def update_list(self, flat_vals):
        vals = list(flat_vals)
        self._add(sum(vals), len(vals))

    def mean(self):
        return self.total / self.count


def _cfg_for_ce(T, S, num_factored_vocabs, factored_vocab_size):
    return _lib.GenieCfg(num_layers=1, num_heads=1, head_dim=16, d_model=16, T=T, S=S, hidden=16,
                         factored_vocab=factored_vocab_size, num_factored=num_factored_vocabs,
                         image_vocab_size=factored_vocab_size ** num_factored_vocabs, precision=_lib.PREC_EXACT)


def factored_ce_sums(labels_flat, factored_logits, num_factored_vocabs=2, factored_vocab_size=512):
    """-> (3,) float64 device tensor [sum CE, sum all-factors-correct, n tokens] over frames 1..T-1."""
    B, n_tok = labels_flat.shape[0], labels_flat.shape[1]
    want = (B, factored_vocab_size, num_factored_vocabs)
    if factored_logits.dim() != 6 or tuple(factored_logits.shape[:3]) != want:   # same complaint as eval_utils.py:62-64
        raise AssertionError(f"Shape of `logits` should be (B, {factored_vocab_size}, {num_factored_vocabs}, T-1, H, W)")
    if not factored_logits.is_cuda:
        raise RuntimeError("1xgpt_amd runs on the GPU only (no CPU fallback): logits must be on cuda")
    t, (h, w) = factored_logits.shape[3] + 1, factored_logits.shape[-2:]
    if t * h * w != n_tok:                                                        # eval_utils.py:66-67
        raise AssertionError("Shape of `factored_logits` does not match flattened latent image size.")
    # (B, Vf, nv, T-1, H, W) -> kernel layout (B, nv*Vf, T-1, S)
    lg = factored_logits.permute(0, 2, 1, 3, 4, 5).contiguous().float()
    labels = labels_flat.to(device=lg.device, dtype=torch.int64).contiguous()
    sums = torch.zeros(3, dtype=torch.float64, device=lg.device)
    # frames-of-power-of-two restriction of check_cfg does not matter here: only T/S/vocab are read
    cfg = _cfg_for_ce(t, h * w, num_factored_vocabs, factored_vocab_size)
    cfg.T = t
    lib = _lib.load()
    _lib.check(lib.genie_factored_ce(cfg, lg.data_ptr(), _lib.LAYOUT_BCTHW, labels.data_ptr(), 0, B, 1, t,
                                     sums.data_ptr(), torch.cuda.current_stream().cuda_stream), "genie_factored_ce")
    return sums


def compute_loss(labels_flat, factored_logits, num_factored_vocabs: int = 2, factored_vocab_size: int = 512) -> float:
    """Cross entropy summed over the factored vocabularies, mean over tokens (eval_utils.py:44-77)."""
    s = factored_ce_sums(labels_flat, factored_logits, num_factored_vocabs, factored_vocab_size)
    return (s[0] / s[2]).item()


def decode_tokens(reshaped_token_ids, decode_latents):
    """(B,T,H,W) token ids -> (B,T,3,256,256) uint8 via the on-device MAGVIT2 decoder (eval_utils.py:28-41).
    Unlike the reference nothing is routed through numpy/PIL: tokens and frames stay in HBM."""
    B, T = reshaped_token_ids.shape[:2]
    imgs = decode_latents(reshaped_token_ids.reshape(B * T, *reshaped_token_ids.shape[2:]))
    return imgs.reshape(B, T, *imgs.shape[1:])
