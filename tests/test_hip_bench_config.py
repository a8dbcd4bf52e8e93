"""GPU (-m gpu): the EXACT configuration bench.py reports -- GENIE_138M shape (L=32, d=512; H=8 and the H=16 variant),
f16x3, teacher-forced evaluate with prefix reuse at chip-filling batches (every GEMM on gemm16_pp_kernel, the QKV operand-plane
epilogue, attn_spatial_dma_kernel, attn_temporal_prefix_f32_mfma_kernel together) -- against

  * the REFERENCE's own run of that workload on bench.py's weights and clip 0 (tests/golden/ev_c138*.npz, made by
    tools/make_goldens.py c138_ev / c138_ev_h16 from genie/evaluate.py:82-122 + eval_utils.compute_loss),
  * this library's full-forward schedule (the reference's 15 x maskgit_steps forwards) on the same clips and draws,
  * itself at other batch sizes (64 clips = BASELINE config 4's per-GPU shard: batch independence, CE = mean of per-clip CEs).

ids are held bit-exact on every timestep whose smallest top-2 logit gap in the reference run (ev_frame_gap) exceeds the f32
accumulation-order noise; the fragile timesteps (a gap of 1e-5 cannot survive ANY reordering of an f32 sum) to near-equality."""
import math
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import pkg, record_measure

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

BF16_CE_BAR = 5e-3   # |CE(bf16) - CE(reference f32)| of one clip over 15 timesteps: ~10x the measured deltas (profiles/r06_bf16_deltas.txt)
ROBUST = 6e-5   # top-2 gap of the reference run below which f32 accumulation order may flip an argmax (as test_hip_configs.py)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def make_ev(cfg, sd, precision, steps=2):
    m = pkg("st_mask_git").STMaskGIT(cfg, precision=precision).load_numpy_state_dict(sd).to("cuda")
    H = W = math.isqrt(cfg.S)
    args = SimpleNamespace(maskgit_steps=steps, temperature=0, latent_h=H, latent_w=W)
    return pkg("evaluate").GenieEvaluator(args, None, "cuda", model=m)


def batch_with_golden(z, cfg, B, seed):
    """B clips: the reference's clip at positions 0 and B-1 (with the reference's unmasking draws), synthetic clips between."""
    synth = pkg("synthetic")
    ids = np.concatenate([z["ids"], synth.make_clips(B - 2, cfg, seed=seed), z["ids"]], 0)
    noise = synth.make_noise((cfg.T - 1, 1, B, cfg.S), seed=seed + 1)
    noise[:, :, 0] = z["ev_noise"][:, :, 0]
    noise[:, :, B - 1] = z["ev_noise"][:, :, 0]
    return ids, noise


def check_samples_against_reference(z, samples_clip, what):
    """samples_clip (T-1, H, W) of the golden clip: exact on robust timesteps, >= 97 % on fragile ones."""
    ref = z["ev_samples"][0].astype(np.int64)
    for k, gap in enumerate(z["ev_frame_gap"]):
        same = samples_clip[k] == ref[k]
        if gap > ROBUST:
            assert same.all(), (what, "timestep", k + 1, int((~same).sum()), "mismatches at gap", gap)
        else:
            assert same.mean() > 0.97, (what, "timestep", k + 1, same.mean(), gap)


@pytest.mark.parametrize("name,precision", [("ev_c138", "f16x3"), ("ev_c138_h16", "f16x3"), ("ev_c138", "exact"),
                                            # the SHIPPED config (genie/configs/magvit_n32_h8_d256.json) through the same
                                            # 15-timestep loop at full depth: BASELINE config 1's workload (tools/make_goldens.py c35_ev)
                                            ("ev_c35", "f16x3"), ("ev_c35", "exact"),
                                            # the reference's DEFAULT attention variant on the GENIE_138M shape (qk_norm=True, genie/config.py:33;
                                            # tools/make_goldens.py c138_ev_qknorm), and the dataclass defaults proper (qk_norm + use_mup:
                                            # c138_ev_default; the muP readout factor is the documented formula, mup_pinned = 0)
                                            ("ev_c138_qknorm", "f16x3"), ("ev_c138_qknorm", "exact"), ("ev_c138_default", "f16x3")])
def test_bench_config_against_reference_and_full_schedule(golden, name, precision):
    z, cfg, sd = golden(name)
    B = 12 if precision == "f16x3" else 4   # 12 clips x 15 frames: >= 192 tiles of 256x256 in every GEMM of the masked passes
    ids, noise = batch_with_golden(z, cfg, B, seed=9100)
    ev = make_ev(cfg, sd, precision)
    eu = pkg("eval_utils")
    d_ids, d_noise = dev(ids), dev(noise)
    s_reuse, fl_reuse = ev.predict_zframe_logits_reuse(d_ids, noise=d_noise)
    s_full, fl_full = ev.predict_zframe_logits(d_ids, noise=d_noise)
    # (1) the two schedules agree on every clip: logits to f32 accumulation-order noise, ids up to fragile gaps
    dl = (fl_full - fl_reuse).abs().max().item()
    assert dl < 5e-5, dl
    assert (s_full != s_reuse).float().mean().item() < 2e-3
    # (2) the reference's clip, first and last in the batch, through both schedules, against the reference's own run
    W = 16
    for b in (0, B - 1):
        for nm_, s_, fl_ in (("reuse", s_reuse, fl_reuse), ("full", s_full, fl_full)):
            check_samples_against_reference(z, s_[b].cpu().numpy(), (name, precision, nm_, b))
            ce = eu.compute_loss(d_ids[b:b + 1], fl_[b:b + 1].contiguous())
            assert abs(ce - float(z["ev_loss"])) < 1e-4, (nm_, b, ce, float(z["ev_loss"]))
            probe = np.stack([fl_[b, :, :, k, s // W, s % W].cpu().numpy() for k, s in zip(z["probe_k"], z["probe_s"])], 0)
            assert np.abs(probe - z["probe_logits"]).max() < 5e-5, (nm_, b, np.abs(probe - z["probe_logits"]).max())
        # per-timestep CE (localises a deviation that the 15-frame mean would dilute)
        lab = d_ids[b].view(cfg.T, cfg.S)
        for k in range(cfg.T - 1):
            two = torch.cat([lab[:1], lab[k + 1:k + 2]], 0).reshape(1, -1)
            ce_k = eu.compute_loss(two, fl_reuse[b:b + 1, :, :, k:k + 1].contiguous())
            assert abs(ce_k - float(z["ev_loss_per_t"][k])) < 2e-4, (b, k, ce_k, float(z["ev_loss_per_t"][k]))
    # (3) evaluate_metric_sums_reuse (what bench.py times) = the same numbers as sums
    sums = ev.evaluate_metric_sums_reuse(d_ids, noise=d_noise).tolist()
    ce_all = eu.compute_loss(d_ids, fl_reuse.contiguous())
    assert abs(sums[0] / sums[1] - ce_all) < 1e-5
    assert sums[5] == B and sums[4] == B * (cfg.T - 1)


@pytest.mark.parametrize("name,precision", [("ev_c138_robust", "f16x3"), ("ev_c138_robust", "exact"), ("ev_c35_robust", "f16x3"),
                                            ("ev_c35_robust", "exact")])
def test_every_sampled_id_bit_exact_on_a_robust_clip(golden, name, precision):
    """No fragile branch: tools/make_goldens.py c138_ev_robust / c35_ev_robust searched clip seeds for a clip whose 15 timesteps ALL have a
    smallest top-2 logit gap above ROBUST in the reference's own run (30 forwards x 512 argmax decisions), so EVERY one of the 15 x 256
    sampled ids must equal the reference's -- through the prefix-reuse schedule (what bench.py times) and the reference's full-forward
    schedule, the clip first and last in a chip-filling batch -- and the CE sits within 1e-4 (genie/evaluate.py:82-122, st_mask_git.py:154-229)."""
    z, cfg, sd = golden(name)
    assert (z["ev_frame_gap"] > ROBUST).all(), z["ev_frame_gap"]
    B = 12 if precision == "f16x3" else 4
    ids, noise = batch_with_golden(z, cfg, B, seed=9500)
    ev = make_ev(cfg, sd, precision)
    eu = pkg("eval_utils")
    d_ids, d_noise = dev(ids), dev(noise)
    ref = z["ev_samples"][0].astype(np.int64)
    for sched in (ev.predict_zframe_logits_reuse, ev.predict_zframe_logits):
        s, fl = sched(d_ids, noise=d_noise)
        for b in (0, B - 1):
            got = s[b].cpu().numpy()
            assert np.array_equal(got, ref), (name, precision, sched.__name__, b, int((got != ref).sum()), "id mismatches")
            ce = eu.compute_loss(d_ids[b:b + 1], fl[b:b + 1].contiguous())
            assert abs(ce - float(z["ev_loss"])) < 1e-4, (sched.__name__, b, ce, float(z["ev_loss"]))


def test_bench_config_64_clip_shard(golden):
    """BASELINE config 4's per-GPU shard (512 clips / 8 GPUs = 64 clips per rank) on one GPU, f16x3 + prefix reuse: the
    reference's clip at both ends of the shard reproduces the reference's ids / CE, the shard's CE is the mean of its parts'
    CEs (the six f64 sums add: what the RCCL all-reduce relies on), and a clip's result does not depend on its batch."""
    z, cfg, sd = golden("ev_c138")
    B = 64
    ids, noise = batch_with_golden(z, cfg, B, seed=9200)
    ev = make_ev(cfg, sd, "f16x3")
    d_ids, d_noise = dev(ids), dev(noise)
    total = ev.evaluate_metric_sums_reuse(d_ids, noise=d_noise)
    samples, _ = ev.predict_zframe_logits_reuse(d_ids, noise=d_noise, return_logits=False)
    for b in (0, B - 1):
        check_samples_against_reference(z, samples[b].cpu().numpy(), ("shard64", b))
    parts = torch.zeros_like(total)
    ce_clip = {}
    for lo, hi in ((0, 1), (1, 16), (16, 32), (32, 63), (63, 64)):
        s = ev.evaluate_metric_sums_reuse(d_ids[lo:hi], noise=d_noise[:, :, lo:hi].contiguous())
        parts += s
        if hi - lo == 1:
            ce_clip[lo] = (s[0] / s[1]).item()
    t, p = total.tolist(), parts.tolist()
    assert t[1] == p[1] and t[3] == p[3] and t[4] == p[4] == B * 15 and t[5] == p[5] == B
    assert abs(t[0] / t[1] - p[0] / p[1]) < 1e-5, (t[0] / t[1], p[0] / p[1])     # different GEMM tilings at 1 / 15 / 31 / 64 clips
    assert abs(t[2] - p[2]) <= 2e-3 * max(1.0, t[3] / 256)                        # sampled-token hits (ids up to fragile gaps)
    for b in (0, B - 1):                                                           # the reference's clip alone
        assert abs(ce_clip[b] - float(z["ev_loss"])) < 1e-4, (b, ce_clip[b], float(z["ev_loss"]))


@pytest.mark.parametrize("name,B", [("ev_c138", 12), ("ev_c35", 12), ("ev_c138_qknorm", 12)])
def test_bench_config_bf16_schedules_agree(golden, name, B):
    """The throughput precision reported beside the headline: reuse and full-forward schedules agree with each other to bf16
    noise, and sit within bf16 noise of the f32 reference (CE within BF16_CE_BAR; DESIGN.md section 2 -- not a parity mode).  For the shipped
    d = 256 config the full-forward schedule runs the fused temporal + MLP kernels (csrc/kernels_fused.hip), the reuse schedule
    the unfused prefix attention with the fused MLP: the comparison crosses both."""
    z, cfg, sd = golden(name)
    ids, noise = batch_with_golden(z, cfg, B, seed=9300)
    ev = make_ev(cfg, sd, "bf16")
    eu = pkg("eval_utils")
    d_ids, d_noise = dev(ids), dev(noise)
    s_reuse, fl_reuse = ev.predict_zframe_logits_reuse(d_ids, noise=d_noise)
    s_full, fl_full = ev.predict_zframe_logits(d_ids, noise=d_noise)
    ce_r, ce_f = eu.compute_loss(d_ids, fl_reuse.contiguous()), eu.compute_loss(d_ids, fl_full.contiguous())
    assert abs(ce_r - ce_f) < 2e-3, (ce_r, ce_f)
    err = (fl_full - fl_reuse).abs()
    assert err.median().item() < 1e-2 and err.max().item() < 0.5, (err.median().item(), err.max().item())
    assert (s_full == s_reuse).float().mean().item() > 0.85
    ce0 = eu.compute_loss(d_ids[:1], fl_reuse[:1].contiguous())
    record_measure(f"bf16_schedules_agree[{name}].ce0_minus_reference", ce0 - float(z["ev_loss"]))
    assert abs(ce0 - float(z["ev_loss"])) < BF16_CE_BAR, (ce0, float(z["ev_loss"]))
    # clip 0 and clip B-1 are the same clip with the same draws: identical results whatever sits between them
    assert abs(eu.compute_loss(d_ids[-1:], fl_reuse[-1:].contiguous()) - ce0) < 2e-3


def test_fused_subblocks_effect_on_sampled_ids_at_depth(golden, monkeypatch):
    """The fused sub-block kernels of the shipped config (csrc/kernels_fused.hip) against the launches they replace, THROUGH ALL 32
    LAYERS and the MaskGIT sampling (st_transformer.py:70-83 x 32, st_mask_git.py:154-229): 12 clips of the reference's full-forward
    schedule (the schedule in which all three sub-blocks run fused), same clips and draws, fused vs GENIE_NO_FUSED.  bf16 is not a parity
    mode: what is held is that fusing changes the sampled ids no more than bf16's own noise does -- the two runs agree with each other at
    least as well as either agrees with the f32 reference run of clip 0 -- and that CE moves by less than 1e-3."""
    z, cfg, sd = golden("ev_c35")
    B = 12
    ids, noise = batch_with_golden(z, cfg, B, seed=9400)
    eu = pkg("eval_utils")
    d_ids, d_noise = dev(ids), dev(noise)
    out = {}
    for tag, env in (("fused", "0"), ("unfused", "1")):
        monkeypatch.setenv("GENIE_NO_FUSED", env)
        ev = make_ev(cfg, sd, "bf16")
        ev.model._weights()
        uses = any(l.temporal.fused_w16 or l.mlp_fused_w16 for l in ev.model._weights()[2])
        assert uses == (tag == "fused")
        s, fl = ev.predict_zframe_logits(d_ids, noise=d_noise)
        out[tag] = (s.cpu().numpy(), eu.compute_loss(d_ids, fl.contiguous()), eu.compute_loss(d_ids[:1], fl[:1].contiguous()))
        del ev
        torch.cuda.empty_cache()
    (sf, cef, cef0), (su, ceu, ceu0) = out["fused"], out["unfused"]
    agree = float((sf == su).mean())
    ref = z["ev_samples"][0].astype(np.int64)
    agree_f_ref, agree_u_ref = float((sf[0] == ref).mean()), float((su[0] == ref).mean())
    print(f"ids: fused vs unfused {agree:.4f}; clip 0 vs the f32 reference: fused {agree_f_ref:.4f}, unfused {agree_u_ref:.4f}; "
          f"CE fused {cef:.6f} unfused {ceu:.6f} (clip 0: {cef0:.6f} / {ceu0:.6f} / reference {float(z['ev_loss']):.6f})")
    assert agree > 0.95 and agree >= min(agree_f_ref, agree_u_ref) - 0.01
    assert abs(agree_f_ref - agree_u_ref) < 0.02          # fusing neither helps nor hurts the agreement with the reference
    record_measure("fused_subblocks_at_depth.cef0_minus_reference", cef0 - float(z["ev_loss"]))
    assert abs(cef - ceu) < 1e-3 and abs(cef0 - float(z["ev_loss"])) < BF16_CE_BAR
