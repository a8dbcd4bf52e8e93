"""GPU (-m gpu): the fused sub-block kernels of the shipped geometry (csrc/kernels_fused.hip; d 256, 8 heads of 32, T 16, bf16).

Each fused kernel must (a) agree with the launches it replaces -- same rounding points, so the difference is accumulation order
and one-ulp bf16 flips -- and (b) hold the bf16 bar of tests/test_hip_configs.py against the bf16-contract oracle
(oracle BF16_MFMA: st_transformer.py:70-83 / attention.py:36-61 with bf16 Linear operands)."""
import numpy as np
import pytest

from conftest import pkg
from oracle import genie_oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def _model(cfg, sd, fused, monkeypatch):
    monkeypatch.setenv("GENIE_NO_FUSED", "0" if fused else "1")
    m = pkg("st_mask_git").STMaskGIT(cfg, precision="bf16").load_numpy_state_dict(sd).to("cuda")
    m._weights()  # the opt-out is read when the weight table is built
    return m


def _uses_fused(m):
    return any(l.temporal.fused_w16 or l.mlp_fused_w16 for l in m._weights()[2])


@pytest.mark.parametrize("qkv_bias", [False, True])
def test_fused_blocks_match_unfused_and_oracle(monkeypatch, qkv_bias):
    cfg = pkg("config").GenieConfig(num_layers=2, num_heads=8, d_model=256, T=16, S=256, num_factored_vocabs=2, qk_norm=False,
                                    use_mup=False, qkv_bias=qkv_bias)
    synth = pkg("synthetic")
    sd = synth.make_state_dict(cfg, seed=77, law="conditioned")
    if qkv_bias:  # the synthetic law leaves biases at zero: make them count
        g = np.random.default_rng(5)
        for k in sd:
            if k.endswith("qkv.bias"):
                sd[k] = (0.05 * g.standard_normal(sd[k].shape)).astype(np.float32)
    B = 19  # 19 clips = 608 temporal blocks / 608 MLP blocks for 512 persistent workgroups: every ring carries over from one
            # block to the next, and the last round is ragged
    ids = synth.make_clips(B, cfg, seed=78)
    x = ids.reshape(B, 16, 16, 16).copy()
    x[:, 8:] = cfg.image_vocab_size
    mf = _model(cfg, sd, True, monkeypatch)
    assert _uses_fused(mf)
    hf = mf.hidden_states(dev(x)).cpu().numpy().copy()
    mu = _model(cfg, sd, False, monkeypatch)
    assert not _uses_fused(mu)
    hu = mu.hidden_states(dev(x)).cpu().numpy().copy()
    scale = np.abs(hu).max()
    d = np.abs(hf - hu)
    print("fused vs unfused hidden: max", d.max(), "median", np.median(d), "scale", scale)
    assert np.isfinite(hf).all()
    assert np.median(d) < 2e-3 * scale and d.max() < 5e-2 * scale
    # against the bf16-contract oracle, first and last clip (the last clip's blocks sit in the ragged round)
    for b in (0, B - 1):
        ref = O.hidden_states(x[b:b + 1], sd, cfg, O.BF16_MFMA)[0]
        ef, eu = np.abs(hf[b] - ref), np.abs(hu[b] - ref)
        print("clip", b, "fused vs oracle max/median", ef.max(), np.median(ef), " unfused vs oracle", eu.max(), np.median(eu))
        assert np.median(ef) < 2e-3 * scale and ef.max() < 5e-2 * scale
        assert np.median(ef) < 2.0 * np.median(eu) + 1e-6     # no worse than the launches it replaces (same contract)
    # batch independence: a clip alone takes the unfused launches (too few blocks), in a batch the fused kernel
    h1 = mf.hidden_states(dev(x[4:5])).cpu().numpy()
    assert np.abs(h1[0] - hf[4]).max() < 5e-2 * scale
    # the readout takes the bf16 shadow of x that the last layer's fused MLP writes: logits of the last frames, fused vs unfused
    lf = mf.compute_logits_frames(dev(x), 14, 16, "token").cpu().numpy()
    lu = mu.compute_logits_frames(dev(x), 14, 16, "token").cpu().numpy()
    dl = np.abs(lf - lu)
    print("fused vs unfused logits: max", dl.max(), "median", np.median(dl))
    assert np.median(dl) < 4e-3 and dl.max() < 8e-2
