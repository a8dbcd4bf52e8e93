"""GPU (-m gpu): geometries at the edges of what genie_check_config admits -- the shortest and longest clips (T = 2 ... 64), the
smallest and largest frames (S = 1 ... 1024 tokens), heads of 16, an odd batch -- through the drop-in module against the oracle, in
every precision the geometry allows.  None of them has a specialised kernel: they run the generic attention kernels, the ragged-tile
GEMM paths and the T > 16 temporal path, which the shipped shapes never touch (st_transformer.py:70-83, attention.py:36-61,
st_mask_git.py:123-229 are shape-agnostic; so is this library, or it refuses)."""
import math

import numpy as np
import pytest

from conftest import pkg
from oracle import genie_oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


GEOMETRIES = [
    # T, S, d, heads, layers, qk_norm, use_mup
    (8, 64, 64, 2, 2, False, False),      # half-length clips, 8x8 frames, heads of 32
    (32, 16, 64, 4, 2, True, False),      # T > 16: the generic temporal path; heads of 16; qk-norm
    (64, 4, 32, 2, 1, False, True),       # the longest clip the config check admits, 2x2 frames, muP scale
    (2, 1024, 32, 2, 1, False, False),    # the largest frame (32x32 tokens) that fits the LDS bound with heads of 16
    (4, 1, 64, 1, 2, True, True),         # one token per frame: spatial attention over a single key
]


@pytest.mark.parametrize("T,S,d,heads,layers,qk_norm,use_mup", GEOMETRIES)
@pytest.mark.parametrize("precision", ["exact", "f16x3", "bf16"])
def test_edge_geometry_logits_loss_and_maskgit_vs_oracle(T, S, d, heads, layers, qk_norm, use_mup, precision):
    if precision == "bf16" and d % 64:
        pytest.skip("bf16 needs d_model % 64 == 0 (genie_check_config)")
    cfg = pkg("config").GenieConfig(num_layers=layers, num_heads=heads, d_model=d, T=T, S=S, num_factored_vocabs=2, qk_norm=qk_norm,
                                    use_mup=use_mup)
    sd = pkg("synthetic").make_state_dict(cfg, seed=7 + T + S, law="conditioned")
    m = pkg("st_mask_git").STMaskGIT(cfg, precision=precision).load_numpy_state_dict(sd).to("cuda")
    H = W = math.isqrt(S)
    B = 3
    ids = pkg("synthetic").make_clips(B, cfg, seed=11 + S)
    x = ids.reshape(B, T, H, W).copy()
    x[:, T // 2:] = cfg.image_vocab_size                      # second half of every clip masked
    nm = {"exact": O.F32, "f16x3": O.F32, "bf16": O.BF16_MFMA}[precision]
    ref = O.compute_logits(x, sd, cfg, nm)
    lg = m.compute_logits(dev(x)).cpu().numpy()
    assert lg.shape == ref.shape == (B, 1024, T, H, W)
    scale = max(1.0, float(np.abs(ref).max()) / 8)
    err = np.abs(lg - ref)
    if precision == "bf16":
        assert np.median(err) < 4e-3 * scale and err.max() < 0.1 * scale, (np.median(err), err.max())
    else:
        assert err.max() < 5e-5 * scale, err.max()
    # forward loss / acc (masked mean over frames >= 1) and one 2-step MaskGIT decode of the first masked frame
    out = m(dev(x.reshape(B, -1)), dev(ids))
    loss_o, acc_o, _ = O.forward_loss_acc(x.reshape(B, -1), ids, sd, cfg, nm)
    assert abs(out.loss.item() - loss_o) < (1e-4 if precision != "bf16" else 3e-3), (out.loss.item(), loss_o)
    t0 = T // 2
    noise = pkg("synthetic").make_noise((1, B, S), seed=5)
    p_dev = dev(x)
    s_dev, _ = m.maskgit_generate(p_dev, t0, maskgit_steps=2, temperature=0.0, noise=dev(noise))
    s_o, _ = O.maskgit_generate(x.copy(), t0, sd, cfg, 2, 0.0, "random", noise=noise, nm=nm)
    same = (s_dev.cpu().numpy() == s_o).mean()
    assert same > (0.98 if precision != "bf16" else 0.9), same
    assert p_dev[:, t0].equal(s_dev)                         # in-place write-back (st_mask_git.py:223)
