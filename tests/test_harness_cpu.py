"""CPU-only checks of the data formats either side of the hot path against the reference's own outputs
(tests/golden/harness.npz, magvit_small.npz from tools/make_goldens.py): dataset windows and filters, LFQ bit order,
the u8 rescale, and the structure / state-dict mapping of the MAGVIT2 conv stacks."""
import ast
import json

import numpy as np
import pytest
import torch

from conftest import GOLDEN, pkg
from oracle import genie_oracle as O


@pytest.fixture(scope="module")
def harness():
    return np.load(f"{GOLDEN}/harness.npz")


@pytest.fixture(scope="module")
def magvit():
    return np.load(f"{GOLDEN}/magvit_small.npz")


def _write_dataset(tmp_path, z):
    D = pkg("data")
    D.write_token_dataset(tmp_path, z["tokens"], z["segment_ids"])
    return D


@pytest.mark.parametrize("name,kw", [
    ("w4s3", dict(window_size=4, stride=3)),
    ("w4s3_overlap", dict(window_size=4, stride=3, filter_overlaps=True)),
    ("w4s1_nointerrupt", dict(window_size=4, stride=1, filter_interrupts=False)),
    ("w4s2_overlap", dict(window_size=4, stride=2, filter_overlaps=True)),
])
def test_dataset_windows_match_reference(tmp_path, harness, name, kw):
    D = _write_dataset(tmp_path, harness)
    ds = D.RawTokenDataset(tmp_path, **kw)
    assert np.array_equal(np.array(ds.valid_start_inds), harness[f"ds_{name}_starts"])
    assert np.array_equal(ds[0]["input_ids"].numpy(), harness[f"ds_{name}_item0"])
    assert np.array_equal(ds[len(ds) - 1]["input_ids"].numpy(), harness[f"ds_{name}_item_last"])
    item = ds[0]
    assert set(item) == {"input_ids", "labels", "attention_mask"} and item["input_ids"].dtype == torch.int64
    assert item["input_ids"].shape == (4 * 16,) and bool(item["attention_mask"].all())


def test_dataset_edge_cases(tmp_path, harness):
    D = pkg("data")
    D.write_token_dataset(tmp_path / "noseg", harness["tokens"])  # no segment_ids.bin
    with pytest.raises(NotImplementedError):
        D.RawTokenDataset(tmp_path / "noseg", window_size=4)
    ds = D.RawTokenDataset(tmp_path / "noseg", window_size=4, filter_interrupts=False)
    assert len(ds) == 60 - 3
    D.write_token_dataset(tmp_path / "short", harness["tokens"][:3], harness["segment_ids"][:3])
    assert len(D.RawTokenDataset(tmp_path / "short", window_size=4)) == 0  # window longer than the data
    meta = json.load(open(tmp_path / "short" / "metadata.json"))
    assert meta["num_images"] == 3 and meta["s"] == 4 and meta["token_dtype"] == "uint32"


def test_lfq_bit_order(magvit):
    z = O.bits_from_tokens(magvit["bits_ids"])
    assert np.array_equal(z, magvit["bits_z"])  # LSB-first after the reference's .flip(1)
    assert np.array_equal(O.tokens_from_bits(magvit["bits_z"]), magvit["bits_ids"])  # exact inverse


def test_rescale_u8(magvit):
    assert np.array_equal(O.rescale_u8_bf16(magvit["rescale_in_bf16_as_f32"]), magvit["rescale_out"])
    assert np.array_equal(O.rescale_u8_bf16(magvit["dec_out_bf16_as_f32"]), magvit["dec_u8_bf16"])
    assert np.array_equal(O.rescale_u8_f32(magvit["dec_out_f32"]), magvit["dec_u8_f32"])


def test_magvit_structure_and_weights(magvit):
    """Same state-dict keys / shapes as the reference stacks; the torch oracle over those parameters reproduces the
    reference's f32 outputs (pins oracle/magvit2_oracle.py on the small config; mid / full: test_magvit_oracle_mid_and_full)."""
    from oracle import magvit2_oracle as MO
    mv = pkg("magvit2")
    small = ast.literal_eval(str(magvit["cfg"]))
    m = mv.VQModel(mv.VQConfig(**small))
    sd = mv.make_vq_state_dict(m, seed=int(magvit["weight_seed"]))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    z = torch.from_numpy(O.bits_from_tokens(magvit["dec_tokens"]))
    y = MO.decoder_forward(m.decoder, z).numpy()
    assert np.abs(y - magvit["dec_out_f32"]).max() < 1e-4
    x = torch.from_numpy(magvit["enc_frames"]).float() / 127.5 - 1.0
    h = MO.encoder_forward(m.encoder, x).numpy()
    assert np.abs(h - magvit["enc_h"]).max() < 1e-4
    # the full-size config has the conv inventory of SURVEY.md a19/a20
    full = mv.VQModel(mv.VQConfig())
    n_dec = sum(p.numel() for p in full.decoder.parameters())
    n_enc = sum(p.numel() for p in full.encoder.parameters())
    assert round(n_dec / 1e6, 1) == 40.5 and round(n_enc / 1e6, 1) == 25.0
    assert MO.depth_to_space(torch.arange(16.).view(1, 4, 2, 2), 2).shape == (1, 1, 4, 4)
    with pytest.raises(RuntimeError, match="GPU only"):
        mv.bits_from_tokens(torch.zeros(1, 4, 4, dtype=torch.long))
    # the modules hold parameters only: there is no torch / MIOpen execution in the product package
    with pytest.raises(RuntimeError, match="parameter holder"):
        m.decoder(z)
    with pytest.raises(ValueError, match="multiples of 64"):   # and no fallback for an uncovered geometry (checked before any GPU use
        mv._check_widths(m.decoder, "HipDecoder")              # would be needed)


def test_magvit_oracle_mid_and_full():
    """oracle/magvit2_oracle.py against the reference's own outputs: the mid config (256/128 widths) and the SHIPPED
    VQConfig (512..128 channels, one 256x256 frame; improved_model.py:103-182), f32, on the CPU."""
    from conftest import GOLDEN
    from oracle import magvit2_oracle as MO
    mv = pkg("magvit2")
    torch.set_grad_enabled(False)
    for name, cfg_kw in (("magvit_mid", None), ("magvit_full", {})):
        z = np.load(f"{GOLDEN}/{name}.npz")
        kw = cfg_kw if cfg_kw is not None else ast.literal_eval(str(z["cfg"]))
        m = mv.VQModel(mv.VQConfig(**kw))
        m.load_state_dict({k: torch.from_numpy(v) for k, v in mv.make_vq_state_dict(m, int(z["weight_seed"])).items()})
        bits = MO.bits_from_tokens(torch.from_numpy(z["dec_tokens"]))
        y = MO.decoder_forward(m.decoder, bits)
        assert np.abs(y.numpy() - z["dec_out_f32"]).max() < 2e-4 * max(1.0, float(np.abs(z["dec_out_f32"]).max()))
        assert np.array_equal(MO.rescale_u8(y).numpy(), z["dec_u8_f32"]) or \
            (np.abs(MO.rescale_u8(y).numpy().astype(int) - z["dec_u8_f32"].astype(int)).max() <= 1)
        x = torch.from_numpy(z["enc_frames"]).float() / 127.5 - 1.0
        h = MO.encoder_forward(m.encoder, x).numpy()
        href = z["enc_h_f32"] if "enc_h_f32" in z.files else z["enc_h"]
        assert np.abs(h - href).max() < 2e-4 * max(1.0, float(np.abs(href).max()))
