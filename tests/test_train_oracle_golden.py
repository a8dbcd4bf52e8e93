"""Pin the training-step oracle (oracle/genie_train_oracle.py) against the reference's own collator outputs,
autograd gradients, clip_grad_norm_ and AdamW results (tests/golden/train_*.npz, made by
tools/make_goldens_train.py from the imported reference).  CPU only."""
import numpy as np
import pytest

from oracle import genie_train_oracle as TO

FULL = ["train_tiny_ln", "train_tiny_qknorm"]
GRAD_TOL = 2e-5  # max |g - g_ref| / max |g_ref| per tensor, fp32 reference autograd vs fp32 oracle


def replay(z, step):
    kinds = z[f"s{step}_draw_kinds"]
    return TO.ReplayDraws(kinds, [z[f"s{step}_draw_{i}"] for i in range(len(kinds))])


@pytest.mark.parametrize("name", FULL + ["train_shape_dh64"])
@pytest.mark.parametrize("step", [0, 1])
def test_collator_bit_exact(golden, name, step):
    z, cfg, _ = golden(name)
    batch = TO.maskgit_collate(z[f"s{step}_ids"], cfg, replay(z, step))
    assert np.array_equal(batch["input_ids"], z[f"s{step}_input_ids"])
    assert np.array_equal(batch["labels"], z[f"s{step}_labels"])
    assert (batch["input_ids"] == cfg.image_vocab_size).any()


def test_collator_covers_both_branches(golden):
    z, _, _ = golden("train_tiny_ln")
    assert {str(z["s0_branch"]), str(z["s1_branch"])} == {"mlm", "nonmlm"}


@pytest.mark.parametrize("name", FULL)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_loss_and_every_gradient(golden, name, dtype):
    z, cfg, sd = golden(name)
    loss, acc, grads = TO.forward_backward(z["s0_input_ids"], z["s0_labels"], sd, cfg, dtype=dtype)
    assert abs(loss - float(z["s0_loss"])) < 1e-5 * abs(float(z["s0_loss"]))
    assert abs(acc - float(z["s0_acc"])) < 1e-7
    assert set(grads) == set(sd)
    for k in sd:
        ref = z[f"s0_grad/{k}"]
        assert grads[k].shape == ref.shape, k
        assert np.abs(grads[k] - ref).max() <= GRAD_TOL * np.abs(ref).max() + 1e-12, k
    assert abs(TO.grad_norm(grads) - float(z["s0_grad_norm"])) < 1e-5 * float(z["s0_grad_norm"])


def test_gradient_samples_real_geometry(golden):
    z, cfg, sd = golden("train_shape_dh64")
    loss, _, grads = TO.forward_backward(z["s0_input_ids"], z["s0_labels"], sd, cfg)
    assert abs(loss - float(z["s0_loss"])) < 1e-5 * abs(float(z["s0_loss"]))
    for k in sd:
        g = grads[k]
        n_ref = float(z[f"s0_gradnorm/{k}"])
        assert abs(np.sqrt((g.astype(np.float64) ** 2).sum()) - n_ref) <= 1e-4 * n_ref + 1e-12, k
        samp = g.reshape(-1)[:: max(1, g.size // 64)][:64]
        assert np.abs(samp - z[f"s0_gradsample/{k}"]).max() <= 1e-4 * np.abs(g).max() + 1e-12, k


@pytest.mark.parametrize("name", FULL)
def test_two_optimizer_steps(golden, name):
    """collate -> forward/backward -> clip -> AdamW (decay grouping of train.py:426-437) -> scheduler, twice."""
    z, cfg, sd = golden(name)
    params = {k: v.astype(np.float32).copy() for k, v in sd.items()}
    state = {}
    for step in range(2):
        batch = TO.maskgit_collate(z[f"s{step}_ids"], cfg, replay(z, step))
        loss, _, grads = TO.forward_backward(batch["input_ids"], batch["labels"], params, cfg)
        assert abs(loss - float(z[f"s{step}_loss"])) < 2e-5 * abs(float(z[f"s{step}_loss"]))
        tn = TO.grad_norm(grads)
        assert abs(tn - float(z[f"s{step}_grad_norm"])) < 1e-5 * tn
        lr = float(z["lr"]) * TO.lr_factor_custom_cosine(step, 1, 4)
        assert abs(lr - float(z[f"s{step}_lr"])) < 1e-12
        TO.adamw_step(params, grads, state, step + 1, lr, float(z["beta1"]), float(z["beta2"]), float(z["eps"]),
                      float(z["weight_decay"]), grad_scale=TO.clip_coef(tn, float(z["max_grad_norm"])))
    for k in sd:
        # two steps of lr ~1e-3 with |m/sqrt(v)| ~ 1: the update is ~2e-3, parity to 1 % of that
        assert np.abs(params[k] - z[f"final_param/{k}"]).max() < 2.5e-5, k


def test_decay_grouping_quirk():
    # train.py:427: the "layer_norm.weight" pattern matches no GENIE parameter, so norm weights decay; biases do not
    assert TO.decays("decoder.layers.0.norm1.weight")
    assert TO.decays("pos_embed_TSC") and TO.decays("decoder.layers.3.mlp.fc1.weight")
    assert not TO.decays("decoder.layers.0.norm1.bias") and not TO.decays("out_x_proj.bias")


def test_lr_factors():
    assert TO.lr_factor_custom_cosine(0, 2, 10) == 0.5 and TO.lr_factor_custom_cosine(1, 2, 10) == 1.0
    assert abs(TO.lr_factor_custom_cosine(10, 2, 10) - 0.1) < 1e-12
    assert TO.lr_factor_linear(0, 0, 10) == 1.0 and TO.lr_factor_linear(5, 0, 10) == 0.5
