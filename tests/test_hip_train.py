"""Training step on the HIP path (through the C ABI via 1xgpt_amd/train.py) against the reference's own autograd
gradients / AdamW results (tests/golden/train_*.npz) and the CPU oracle.  Needs a real MI355X: ``-m gpu``.

Tolerances (f32 MFMA, f32 accumulation order only): loss within 1e-5 relative; every gradient tensor within
1e-4 of its largest element (reference autograd itself is f32); parameters after two clipped AdamW steps within
a mean error of 1e-3 of the update, with Adam's sign sensitivity on near-zero gradients bounded separately (see the test).
"""
import numpy as np
import pytest

from conftest import pkg
from oracle import genie_train_oracle as TO

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

GRAD_TOL = 1e-4


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def make_trainer(cfg, sd, precision="exact", **kw):
    model = pkg("st_mask_git").STMaskGIT(cfg, precision=precision).load_numpy_state_dict(sd).to("cuda")
    return pkg("train").GenieTrainer(model, **kw)


def fro_err(a, ref):
    a, ref = a.astype(np.float64), ref.astype(np.float64)
    return float(np.sqrt(((a - ref) ** 2).sum()) / (np.sqrt((ref ** 2).sum()) + 1e-30))


def rel_err(a, ref):
    return float(np.abs(a - ref).max() / (np.abs(ref).max() + 1e-30))


@pytest.mark.parametrize("precision", ["exact", "f16x3"])
@pytest.mark.parametrize("name", ["train_tiny_ln", "train_tiny_qknorm"])
def test_every_gradient_vs_reference_autograd(golden, name, precision):
    """f16x3 (split-f16 matrix cores, 22-bit operands) is held to the SAME bar as the f32 path."""
    z, cfg, sd = golden(name)
    tr = make_trainer(cfg, sd, precision)
    loss, acc = tr.forward_backward(dev(z["s0_input_ids"]), dev(z["s0_labels"]))
    assert abs(float(loss) - float(z["s0_loss"])) < 1e-5 * float(z["s0_loss"])
    assert abs(float(acc) - float(z["s0_acc"])) < 1e-7
    worst = {}
    for k, g in tr.gradients().items():
        worst[k] = rel_err(g.cpu().numpy(), z[f"s0_grad/{k}"])
    bad = {k: v for k, v in worst.items() if v > GRAD_TOL}
    assert not bad, bad
    gn = float(torch.sqrt(tr.grad_sumsq()[0]))
    assert abs(gn - float(z["s0_grad_norm"])) < 1e-5 * gn


@pytest.mark.parametrize("name", ["train_tiny_ln", "train_tiny_qknorm"])
def test_bf16_gradients_close_to_reference(golden, name):
    """bf16 operands (what the reference computes under `--mixed_precision bf16`): every gradient tensor within 3 %
    (Frobenius) of the f32 autograd gradient, loss within 1e-2."""
    z, cfg, sd = golden(name)
    tr = make_trainer(cfg, sd, "bf16")
    loss, _ = tr.forward_backward(dev(z["s0_input_ids"]), dev(z["s0_labels"]))
    assert abs(float(loss) - float(z["s0_loss"])) < 1e-2
    bad = {}
    for k, g in tr.gradients().items():
        e = fro_err(g.cpu().numpy(), z[f"s0_grad/{k}"])
        if e > 3e-2:
            bad[k] = e
    assert not bad, bad
    gn = float(torch.sqrt(tr.grad_sumsq()[0]))
    assert abs(gn - float(z["s0_grad_norm"])) < 1e-2 * gn


@pytest.mark.parametrize("precision", ["exact", "f16x3"])
@pytest.mark.parametrize("name", ["train_tiny_ln", "train_tiny_qknorm"])
def test_two_optimizer_steps_vs_reference(golden, name, precision):
    """collate (replayed draws) -> forward/backward -> clip_grad_norm_ -> AdamW with the reference's grouping -> the
    custom_cosine schedule, twice; compare with torch.optim.AdamW's parameters."""
    z, cfg, sd = golden(name)
    tm = pkg("train")
    tr = make_trainer(cfg, sd, precision, lr=float(z["lr"]), betas=(float(z["beta1"]), float(z["beta2"])), eps=float(z["eps"]),
                      weight_decay=float(z["weight_decay"]), max_grad_norm=float(z["max_grad_norm"]),
                      lr_lambda=tm.lr_factor_custom_cosine(1, 4))
    data = pkg("data")
    for step in range(2):
        kinds = z[f"s{step}_draw_kinds"]
        draws = TO.ReplayDraws(kinds, [z[f"s{step}_draw_{i}"] for i in range(len(kinds))])
        batch = data.maskgit_collate(dev(z[f"s{step}_ids"]), cfg, draws)
        assert np.array_equal(batch["input_ids"].cpu().numpy(), z[f"s{step}_input_ids"])
        out = tr.train_step(batch)
        assert abs(float(out["loss"]) - float(z[f"s{step}_loss"])) < 2e-5 * float(z[f"s{step}_loss"])
        assert abs(float(out["grad_norm"]) - float(z[f"s{step}_grad_norm"])) < 2e-5 * float(z[f"s{step}_grad_norm"])
        assert abs(out["lr"] - float(z[f"s{step}_lr"])) < 1e-12
    state = tr.model.state_dict()
    for k in sd:
        # Adam's first updates are lr * g / (|g| + eps) ~ lr * sign(g): an element whose gradient is ~1e-7 turns an f32
        # rounding difference into a visible fraction of lr (1e-3), whatever computes it.  So: the typical element must
        # agree to 1e-3 of the two-step update, at most 0.1 % of a tensor's elements may differ by more than 1 % of
        # it, and none by more than the update itself.
        err = np.abs(state[k].cpu().numpy() - z[f"final_param/{k}"])
        assert err.mean() < 2e-6, k
        assert (err > 2.5e-5).mean() <= 1e-3, k
        assert err.max() < 2e-3, k


@pytest.mark.parametrize("precision", ["exact", "f16x3"])
def test_real_geometry_vs_reference_samples(golden, precision):
    """T=16, S=256, Dh=64: the production tile shapes (MFMA attention forward, 256x256 score GEMMs backward)."""
    z, cfg, sd = golden("train_shape_dh64")
    tr = make_trainer(cfg, sd, precision)
    loss, _ = tr.forward_backward(dev(z["s0_input_ids"]), dev(z["s0_labels"]))
    assert abs(float(loss) - float(z["s0_loss"])) < 1e-5 * float(z["s0_loss"])
    for k, g in tr.gradients().items():
        g = g.cpu().numpy()
        n_ref = float(z[f"s0_gradnorm/{k}"])
        # 4096-token contractions with heavy cancellation (bias / LayerNorm gradients): f32 reference noise ~1e-4;
        # f16x3 operands carry 22 bits instead of 24
        tol = 2e-4 if precision == "exact" else 5e-4
        assert abs(np.sqrt((g.astype(np.float64) ** 2).sum()) - n_ref) <= tol * n_ref + 1e-12, k
        samp = g.reshape(-1)[:: max(1, g.size // 64)][:64]
        assert np.abs(samp - z[f"s0_gradsample/{k}"]).max() <= tol * np.abs(g).max() + 1e-12, k


def test_real_geometry_bf16(golden):
    """The bf16 precision at the production geometry (T=16, S=256, Dh=64): forward on the bf16 GEMMs, spatial attention BACKWARD
    on the bf16 matrix cores (kernels_attn_bwd16.hip: dQ / dK / dV and the softmax backward from bf16 operands): every tensor's
    gradient norm within 2 % of the f32 reference's, sampled entries within 3 % of the tensor's largest."""
    z, cfg, sd = golden("train_shape_dh64")
    tr = make_trainer(cfg, sd, "bf16")
    loss, _ = tr.forward_backward(dev(z["s0_input_ids"]), dev(z["s0_labels"]))
    assert abs(float(loss) - float(z["s0_loss"])) < 1e-2
    for k, g in tr.gradients().items():
        g = g.cpu().numpy()
        n_ref = float(z[f"s0_gradnorm/{k}"])
        assert abs(np.sqrt((g.astype(np.float64) ** 2).sum()) - n_ref) <= 2e-2 * n_ref + 1e-12, k
        samp = g.reshape(-1)[:: max(1, g.size // 64)][:64]
        assert np.abs(samp - z[f"s0_gradsample/{k}"]).max() <= 3e-2 * np.abs(g).max() + 1e-12, k


@pytest.mark.parametrize("H,d,qk_norm", [(4, 128, False), (2, 128, True), (4, 256, False), (8, 512, False)])
def test_bf16_attention_backward_vs_exact(H, d, qk_norm):
    """S = 256 with head_dim 32 and 64, with and without qk-norm: the bf16 trainer (bf16-MFMA spatial attention backward; at
    d = 256 / 512 every weight gradient on the TN kernel with its transposing LDS reads, kernels_gemm_tn.hip, several column
    tiles and token slabs) against the exact trainer on the same weights and batch -- every gradient tensor within 3 %
    (Frobenius)."""
    cfg = pkg("config").GenieConfig(num_layers=2, num_heads=H, d_model=d, T=4, S=256, num_factored_vocabs=2, qk_norm=qk_norm,
                                    use_mup=False, num_prompt_frames=2)
    synth = pkg("synthetic")
    sd = synth.make_state_dict(cfg, seed=17, law="conditioned")
    ids = synth.make_clips(2, cfg, seed=18)
    x = ids.reshape(2, 4, 256).copy()
    x[:, 2:] = cfg.image_vocab_size
    grads = {}
    for prec in ("exact", "bf16"):
        tr = make_trainer(cfg, sd, prec)
        tr.forward_backward(dev(x.reshape(2, -1)), dev(ids))
        grads[prec] = {k: g.cpu().numpy().astype(np.float64) for k, g in tr.gradients().items()}
    bad = {k: fro_err(grads["bf16"][k], grads["exact"][k]) for k in grads["exact"]}
    bad = {k: v for k, v in bad.items() if v > 3e-2}
    assert not bad, bad


@pytest.mark.parametrize("H,d,B,qk_norm,T", [(4, 128, 3, False, 4), (2, 64, 1, False, 4), (2, 128, 2, True, 4),
                                              (4, 128, 1, True, 4), (4, 128, 2, False, 16), (2, 128, 1, True, 16),
                                              (4, 256, 2, False, 4), (8, 512, 1, False, 4)])
@pytest.mark.parametrize("use_mup", [False, True])
def test_gradients_vs_oracle(H, d, B, qk_norm, T, use_mup):
    """Other widths / head sizes (Dh = 32, 64), an odd batch, T = 16 (the MFMA temporal kernels, forward and backward,
    with and without qk-norm) and the shipped widths d = 256 / 512 (their bandwidth-tuned LayerNorm forward / backward
    kernels) against the NumPy restatement."""
    if use_mup and (B != 1 or T != 4):
        pytest.skip("muP scaling (attention scale 8/Dh, readout multiplier 256/d) is covered on the single-clip cases")
    cfg = pkg("config").GenieConfig(num_layers=2, num_heads=H, d_model=d, T=T, S=16, num_factored_vocabs=2,
                                    qk_norm=qk_norm, num_prompt_frames=2, use_mup=use_mup)
    syn = pkg("synthetic")
    sd = syn.make_state_dict(cfg, seed=77 + H, law="conditioned")
    ids = syn.make_clips(B, cfg, seed=900 + d)
    batch = TO.maskgit_collate(ids, cfg, TO.NumpyDraws(5 + B))
    loss_o, acc_o, g_o = TO.forward_backward(batch["input_ids"], batch["labels"], sd, cfg)
    tr = make_trainer(cfg, sd)
    loss, acc = tr.forward_backward(dev(batch["input_ids"]), dev(batch["labels"]))
    assert abs(float(loss) - loss_o) < 1e-5 * abs(loss_o)
    assert abs(float(acc) - acc_o) < 1e-7
    for k, g in tr.gradients().items():
        assert rel_err(g.cpu().numpy(), g_o[k]) < GRAD_TOL, k


@pytest.mark.parametrize("H,d,qk_norm", [(2, 64, False), (1, 64, True)])
def test_fused_spatial_backward_vs_oracle(H, d, qk_norm):
    """S = 256 tokens per frame: the fused spatial attention backward (head_dim 32 and 64, with and without qk-norm)."""
    cfg = pkg("config").GenieConfig(num_layers=1, num_heads=H, d_model=d, T=2, S=256, num_factored_vocabs=2,
                                    qk_norm=qk_norm, num_prompt_frames=1)
    syn = pkg("synthetic")
    sd = syn.make_state_dict(cfg, seed=5 + H, law="conditioned")
    ids = syn.make_clips(2, cfg, seed=31)
    batch = TO.maskgit_collate(ids, cfg, TO.NumpyDraws(3))
    loss_o, _, g_o = TO.forward_backward(batch["input_ids"], batch["labels"], sd, cfg)
    tr = make_trainer(cfg, sd)
    loss, _ = tr.forward_backward(dev(batch["input_ids"]), dev(batch["labels"]))
    assert abs(float(loss) - loss_o) < 1e-5 * abs(loss_o)
    for k, g in tr.gradients().items():
        assert rel_err(g.cpu().numpy(), g_o[k]) < GRAD_TOL, k


@pytest.mark.parametrize("precision", ["exact", "f16x3", "bf16"])
def test_bit_reproducible_and_accumulation(golden, precision):
    z, cfg, sd = golden("train_tiny_ln")
    tr = make_trainer(cfg, sd, precision)
    a_ids, a_lab = dev(z["s0_input_ids"]), dev(z["s0_labels"])
    b_ids, b_lab = dev(z["s1_input_ids"]), dev(z["s1_labels"])
    tr.forward_backward(a_ids, a_lab)
    g1 = tr.grads.clone()
    tr.forward_backward(a_ids, a_lab)
    assert torch.equal(g1, tr.grads)  # fixed reduction order everywhere
    tr.forward_backward(b_ids, b_lab)
    g2 = tr.grads.clone()
    tr.forward_backward(a_ids, a_lab)
    tr.forward_backward(b_ids, b_lab, accumulate=True)
    assert float((tr.grads - (g1 + g2)).abs().max()) <= 1e-6 * float(g1.abs().max())


def test_unsupported_configs_fail_loudly(golden):
    z, cfg, sd = golden("train_tiny_ln")
    model = pkg("st_mask_git").STMaskGIT(cfg, precision="exact").load_numpy_state_dict(sd).to("cuda")
    tr = pkg("train").GenieTrainer(model)
    cfg32 = pkg("config").GenieConfig(num_layers=1, num_heads=2, d_model=32, T=4, S=16, num_factored_vocabs=2,
                                      qk_norm=False)
    m32 = pkg("st_mask_git").STMaskGIT(cfg32, precision="bf16").to("cuda")
    with pytest.raises(pkg("_lib").GenieHipError):  # 16-bit training tiles need d_model % 64 == 0
        t32 = pkg("train").GenieTrainer(m32)
        ids = torch.zeros(1, 64, dtype=torch.int64, device="cuda")
        t32.forward_backward(ids, ids)
    with pytest.raises(RuntimeError):
        tr.forward_backward(torch.from_numpy(z["s0_input_ids"]), torch.from_numpy(z["s0_labels"]))


@pytest.mark.parametrize("precision", ["exact", "bf16"])
def test_overfitting_one_batch_reduces_the_loss(golden, precision):
    """End-to-end sanity of the whole chain (forward, backward, clip, AdamW, 16-bit weight refresh): 40 updates on one
    collated batch must drive its masked CE well below the 2 ln 512 = 12.48 of a uniform predictor."""
    z, cfg, _ = golden("train_tiny_ln")
    syn = pkg("synthetic")
    sd = syn.make_state_dict(cfg, seed=1, law="init")
    tr = make_trainer(cfg, sd, precision, lr=3e-3, max_grad_norm=1.0)
    batch = {"input_ids": dev(z["s0_input_ids"]), "labels": dev(z["s0_labels"])}
    first = float(tr.train_step(batch)["loss"])
    for _ in range(39):
        last = float(tr.train_step(batch)["loss"])
    assert 12.0 < first < 13.0
    assert last < first - 2.0, (first, last)
    # the module's inference entry point sees the trained weights (shared flat buffer + refreshed 16-bit copies)
    out = tr.model(batch["input_ids"], batch["labels"])
    assert abs(float(out.loss) - last) < 0.5


def test_train_cli_on_a_dataset_directory(tmp_path):
    """tools/train.py end to end on the on-disk dataset layout (video.bin / segment_ids.bin / metadata.json, a GenieConfig
    json): windows -> collator -> updates -> eval -> `save_pretrained`; the checkpoint reloads through `from_pretrained`."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    data, cfgm = pkg("data"), pkg("config")
    g = np.random.default_rng(0)
    for split in ("train", "val"):
        data.write_token_dataset(tmp_path / split, g.integers(0, 262144, (60, 4, 4)), np.zeros(60, np.int32), hz=30)
    cfg = cfgm.GenieConfig(num_layers=1, num_heads=2, d_model=64, T=4, S=16, num_factored_vocabs=2, qk_norm=False,
                           num_prompt_frames=2)
    cfg.save_pretrained(tmp_path / "cfg.json")
    out = tmp_path / "out"
    cmd = [sys.executable, f"{REPO}/tools/train.py", "--genie_config", str(tmp_path / "cfg.json"), "--train_data_dir",
           str(tmp_path / "train"), "--val_data_dir", str(tmp_path / "val"), "--window_size", "4", "--stride", "2",
           "--output_dir", str(out), "--per_device_train_batch_size", "4", "--per_device_eval_batch_size", "4",
           "--max_train_steps", "3", "--eval_every_n_steps", "3", "--gradient_accumulation_steps", "2", "--seed", "1",
           "--precision", "exact", "--lr_scheduler_type", "custom_cosine", "--num_warmup_steps", "1"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "step 3: train_loss" in res.stdout and "eval_loss" in res.stdout
    ckpt = out / "final_checkpt"
    assert json.load(open(ckpt / "config.json"))["d_model"] == 64
    # resume: two more updates from the saved weights + optimizer state
    cmd2 = [a for a in cmd]
    cmd2[cmd2.index("--max_train_steps") + 1] = "5"
    cmd2[cmd2.index("--output_dir") + 1] = str(tmp_path / "out2")
    res2 = subprocess.run(cmd2 + ["--resume_from_checkpoint", str(ckpt)], capture_output=True, text=True, timeout=600)
    assert res2.returncode == 0, res2.stdout[-2000:] + res2.stderr[-2000:]
    assert "step 4: train_loss" in res2.stdout and "step 5: train_loss" in res2.stdout and "step 3:" not in res2.stdout
    model = pkg("st_mask_git").STMaskGIT.from_pretrained(str(ckpt)).to("cuda")
    ds = data.RawTokenDataset(tmp_path / "val", window_size=4, stride=2, filter_overlaps=True)
    ids = ds.batch(range(2)).cuda()
    x = ids.clone().view(2, 4, 16)
    x[:, 2:] = cfg.image_vocab_size
    loss = float(model(x.view(2, -1), ids).loss)
    assert np.isfinite(loss) and 10.0 < loss < 15.0


def test_optimizer_state_round_trip(golden):
    """trainer.state_dict() / load_state_dict(): a resumed trainer takes bit-identical steps."""
    z, cfg, sd = golden("train_tiny_ln")
    batch = {"input_ids": dev(z["s0_input_ids"]), "labels": dev(z["s0_labels"])}
    a = make_trainer(cfg, sd, lr=1e-3, weight_decay=0.1)
    a.train_step(batch)
    a.train_step(batch)
    ck_model = {k: v.clone() for k, v in a.model.state_dict().items()}
    ck_opt = a.state_dict()
    ref = a.train_step(batch)
    want = {k: v.clone() for k, v in a.model.state_dict().items()}
    model_b = pkg("st_mask_git").STMaskGIT(cfg, precision="exact").to("cuda")
    model_b.load_state_dict(ck_model)
    b = pkg("train").GenieTrainer(model_b, lr=1e-3, weight_decay=0.1)
    b.load_state_dict(ck_opt)
    got = b.train_step(batch)
    assert float(got["loss"]) == float(ref["loss"]) and b.completed_steps == a.completed_steps == 3
    for k, v in b.model.state_dict().items():
        assert torch.equal(v, want[k]), k
