"""Teacher-forced prefix reuse (genie_clean_pass + genie_masked_frames_logits): same outputs as the reference's
15 x maskgit_steps full forwards -- ids bit-exact, CE within 1e-4 of the reference goldens.  Needs a GPU: -m gpu."""
import math
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def make_ev(golden, name, precision, steps=2):
    z, cfg, sd = golden(name)
    m = pkg("st_mask_git").STMaskGIT(cfg, precision=precision).load_numpy_state_dict(sd).to("cuda")
    H = W = math.isqrt(cfg.S)
    args = SimpleNamespace(maskgit_steps=steps, temperature=0, latent_h=H, latent_w=W)
    return z, cfg, pkg("evaluate").GenieEvaluator(args, None, "cuda", model=m)


@pytest.mark.parametrize("precision", ["exact", "f16x3"])
@pytest.mark.parametrize("name", ["tiny_ln", "tiny_qknorm", "tiny_mup", "tiny_qknorm_mup"])
def test_reuse_matches_reference_tiny(golden, name, precision):
    z, cfg, ev = make_ev(golden, name, precision)
    samples, fl = ev.predict_zframe_logits_reuse(dev(z["ids"]), noise=dev(z["ev_noise"]))
    assert np.array_equal(samples.cpu().numpy(), z["ev_samples"])
    assert np.abs(fl.cpu().numpy() - z["ev_logits"]).max() < 1e-4
    sums = ev.evaluate_metric_sums_reuse(dev(z["ids"]), noise=dev(z["ev_noise"])).tolist()
    assert abs(sums[0] / sums[1] - float(z["ev_loss"])) < 1e-4
    assert abs(sums[2] / sums[3] - float(z["ev_acc"])) < 1e-7


@pytest.mark.parametrize("precision", ["exact", "f16x3"])
@pytest.mark.parametrize("name", ["shape_dh32", "shape_dh64"])
def test_reuse_matches_reference_real_geometry(golden, name, precision):
    z, cfg, ev = make_ev(golden, name, precision)
    ids = dev(z["ids"])
    sums = ev.evaluate_metric_sums_reuse(ids, noise=dev(z["ev_noise"])).tolist()
    assert abs(sums[0] / sums[1] - float(z["ev_loss"])) < 1e-4
    assert abs(sums[2] / sums[3] - float(z["ev_acc"])) < 1e-7
    samples, _ = ev.predict_zframe_logits_reuse(ids, noise=dev(z["ev_noise"]), return_logits=False)
    assert np.array_equal(samples.cpu().numpy(), z["ev_samples"])


@pytest.mark.parametrize("steps", [1, 3, 8])
def test_reuse_equals_full_forward_path(golden, steps):
    """Against this library's own full-forward evaluator on a batch of clips, several step counts, qk-norm geometry."""
    z, cfg, ev = make_ev(golden, "shape_dh64_qknorm", "exact", steps)
    ids = dev(pkg("synthetic").make_clips(3, cfg, seed=11))
    noise = torch.rand(cfg.T - 1, max(steps - 1, 1), 3, cfg.S, device="cuda")
    s_full, fl_full = ev.predict_zframe_logits(ids, noise=noise)
    s_reuse, fl_reuse = ev.predict_zframe_logits_reuse(ids, noise=noise)
    assert (fl_full - fl_reuse).abs().max().item() < 2e-5
    mism = (s_full != s_reuse).float().mean().item()
    assert mism < 2e-3  # identical up to top-2 logit gaps below the f32 accumulation-order noise (random clips)


def test_reuse_bf16_close(golden):
    z, cfg, ev = make_ev(golden, "shape_dh64", "bf16")
    sums = ev.evaluate_metric_sums_reuse(dev(z["ids"]), noise=dev(z["ev_noise"])).tolist()
    from conftest import record_measure
    record_measure("prefix_reuse.bf16_ce_minus_reference", sums[0] / sums[1] - float(z["ev_loss"]))
    assert abs(sums[0] / sums[1] - float(z["ev_loss"])) < 5e-3   # measured -4.1e-4 (profiles/r06_bf16_deltas.txt)


@pytest.mark.parametrize("precision", ["exact", "f16x3"])
def test_generate_with_kv_cache_matches_reference(golden, precision):
    """generate.py semantics through genie_frame_pass (temporal KV cache) against the reference harness golden."""
    import ast
    from conftest import GOLDEN
    z = np.load(f"{GOLDEN}/harness.npz")
    cfg = pkg("config").GenieConfig(**ast.literal_eval(str(z["cfg"])))
    sd = pkg("synthetic").make_state_dict(cfg, seed=int(z["weight_seed"]))
    m = pkg("st_mask_git").STMaskGIT(cfg, precision=precision).load_numpy_state_dict(sd).to("cuda")
    G = pkg("generate")
    for tf, key in [(False, "gen_ar"), (True, "gen_tf")]:
        out = G.generate_frames_cached(m, dev(z["gen_example"]), num_prompt_frames=2, maskgit_steps=2,
                                       teacher_force_time=tf, noise=dev(z[key + "_noise"]))
        assert np.array_equal(out.cpu().numpy(), z[key + "_outputs"])


@pytest.mark.parametrize("name,steps", [("shape_dh64", 2), ("shape_dh32", 8), ("shape_dh64_qknorm", 3)])
def test_generate_kv_cache_equals_full_forward(golden, name, steps):
    z, cfg, sd = golden(name)
    m = pkg("st_mask_git").STMaskGIT(cfg, precision="f16x3").load_numpy_state_dict(sd).to("cuda")
    G = pkg("generate")
    ex = dev(pkg("synthetic").make_clips(2, cfg, seed=21)).view(2, 16, 16, 16)
    noise = torch.rand(8, max(steps - 1, 1), 2, cfg.S, device="cuda")
    full = G.generate_frames(m, ex, 8, steps, 0.0, False, noise=noise)
    cached = G.generate_frames_cached(m, ex, 8, steps, 0.0, False, noise=noise)
    mism = (full != cached).float().mean().item()
    assert mism < 2e-3  # autoregressive: identical unless a top-2 logit gap sits below f32 accumulation noise


@pytest.mark.parametrize("precision,B", [("exact", 2), ("f16x3", 2), ("f16x3", 1)])
def test_prompt_pass_fills_cache_like_frame_passes(golden, precision, B):
    """generate()'s prompt: ONE P-frame genie_clean_pass writing into the T-frame KV-cache layout must leave slots 0..P-1 of
    every layer exactly as P single-frame genie_frame_pass calls do (same per-row arithmetic; different GEMM tilings)."""
    z, cfg, sd = golden("shape_dh64")
    _lib = pkg("_lib")
    lib = _lib.load()
    m = pkg("st_mask_git").STMaskGIT(cfg, precision=precision).load_numpy_state_dict(sd).to("cuda")
    c, w = m._weights()[:2]
    T, S, d, L, P = cfg.T, cfg.S, cfg.d_model, cfg.num_layers, 8
    ids = dev(pkg("synthetic").make_clips(B, cfg, seed=31)).view(B, T, S)
    ws = m._workspace(B)
    nbytes = lib.genie_prefix_cache_bytes(c, B)
    st = torch.cuda.current_stream().cuda_stream
    ca = torch.zeros(nbytes // 4, dtype=torch.float32, device="cuda")
    cb = torch.zeros(nbytes // 4, dtype=torch.float32, device="cuda")
    _lib.check(lib.genie_clean_pass(c, w, ids[:, :P].contiguous().data_ptr(), B, P, T, ca.data_ptr(), nbytes, ws.data_ptr(),
                                    ws.numel(), st), "genie_clean_pass")
    for t in range(P):
        _lib.check(lib.genie_frame_pass(c, w, ids[:, t].contiguous().data_ptr(), B, t, cb.data_ptr(), nbytes, 0, ws.data_ptr(),
                                        ws.numel(), st), "genie_frame_pass")
    a = ca.view(L, B, T, S, 3 * d)[:, :, :P]
    b = cb.view(L, B, T, S, 3 * d)[:, :, :P]
    assert torch.isfinite(a).all()
    scale = b.abs().max().item()
    assert (a - b).abs().max().item() < 3e-5 * max(1.0, scale)
    assert (ca.view(L, B, T, S, 3 * d)[:, :, P:] == 0).all()   # slots >= P untouched


@pytest.mark.parametrize("name", ["shape_dh64", "shape_dh32"])
@pytest.mark.parametrize("precision,tol_max,tol_med", [("f16x3", 3e-5, 1e-5), ("bf16", 8e-2, 6e-3), ("exact", 3e-5, 1e-5)])
def test_single_frame_pass_equals_full_forward_frame(golden, name, precision, tol_max, tol_med):
    """One-frame passes run their own kernels (split-K small GEMM with the LayerNorm in its fragment path, key-split spatial
    attention, register-resident temporal attention over the cache): the logits of frames 0 and 1 from genie_frame_pass must
    equal those frames of the full 16-frame forward (temporal attention is causal) -- to f32 noise in exact / f16x3, to bf16
    noise in bf16."""
    z, cfg, sd = golden(name)
    _lib = pkg("_lib")
    lib = _lib.load()
    m = pkg("st_mask_git").STMaskGIT(cfg, precision=precision).load_numpy_state_dict(sd).to("cuda")
    c, w = m._weights()[:2]
    T, S = cfg.T, cfg.S
    V = cfg.factored_vocab_size * cfg.num_factored_vocabs
    ids = dev(pkg("synthetic").make_clips(1, cfg, seed=77)).view(1, T, S)
    full = m.compute_logits(ids.view(1, T, 16, 16))                      # (1, V, T, 16, 16)
    ws = m._workspace(1)
    nbytes = lib.genie_prefix_cache_bytes(c, 1)
    cache = torch.zeros(nbytes // 4, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for t in (0, 1):
        lg = torch.empty(1, S, V, dtype=torch.float32, device="cuda")
        _lib.check(lib.genie_frame_pass(c, w, ids[:, t].contiguous().data_ptr(), 1, t, cache.data_ptr(), nbytes, lg.data_ptr(),
                                        ws.data_ptr(), ws.numel(), st), "genie_frame_pass")
        ref = full[0, :, t].reshape(V, S).T                               # (S, V)
        err = (lg[0] - ref).abs()
        assert err.max().item() < tol_max and err.median().item() < tol_med, (t, err.max().item(), err.median().item())
