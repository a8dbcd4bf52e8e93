"""GenieConfig -- hyper-parameter surface of the GENIE world model.

Mirrors the reference dataclass field-for-field (reference: genie/config.py:7-55) so that a
``config.json`` written by the reference's ``save_pretrained`` loads here unchanged and vice versa.
``factored_vocab_size`` is derived: the ``num_factored_vocabs``-th integer root of ``image_vocab_size``
(reference: genie/config.py:54-55, genie/factorization_utils.py:103-106).
"""
import json
from dataclasses import dataclass, fields


def nth_root(x: int, n: int) -> int:
    root = round(x ** (1 / n))
    assert root ** n == x, (x, n, root)
    return root


@dataclass
class GenieConfig:
    num_layers: int
    num_heads: int
    d_model: int
    T: int = 16   # frames per clip
    S: int = 256  # tokens per frame (16x16)
    image_vocab_size: int = 262144
    use_mup: bool = False

    num_factored_vocabs: int = 1
    factored_vocab_size: int = None

    # training-only knobs, carried so that reference JSON round-trips
    max_corrupt_rate: float = 0.2
    non_mlm_ratio: float = 0.5
    num_prompt_frames: int = 8

    qkv_bias: bool = False
    proj_bias: bool = True
    attn_drop: float = 0.0
    qk_norm: bool = True

    mlp_ratio: float = 4.0
    mlp_drop: float = 0.0
    mlp_bias: bool = True

    def __post_init__(self):
        self.factored_vocab_size = nth_root(self.image_vocab_size, self.num_factored_vocabs)

    def save_pretrained(self, json_path):
        with open(json_path, "w") as f:
            json.dump(vars(self), f)

    @classmethod
    def from_pretrained(cls, json_path):
        with open(json_path, "r") as f:
            raw = json.load(f)
        known = {f.name for f in fields(cls)}
        return cls(**{k: v for k, v in raw.items() if k in known})

    def shallow_copy(self):
        return GenieConfig(**vars(self))

    # -- derived quantities used by the HIP path ---------------------------------------------
    @property
    def head_dim(self) -> int:
        return self.d_model // self.num_heads

    @property
    def attn_scale(self) -> float:
        # reference: genie/attention.py:26
        return 8.0 / self.head_dim if self.use_mup else self.head_dim ** -0.5

    @property
    def readout_mult(self) -> float:
        # muP readout: Linear(output_mult * x / width_mult), output_mult = 1, width_mult = d/256
        # (reference: genie/st_mask_git.py:298-304, 316-323)
        return 256.0 / self.d_model if self.use_mup else 1.0

    @property
    def mask_token_id(self) -> int:
        return self.image_vocab_size  # reference: genie/st_mask_git.py:51


# Shipped shape (reference: genie/configs/magvit_n32_h8_d256.json) = "GENIE_35M".
def c35() -> GenieConfig:
    return GenieConfig(num_layers=32, num_heads=8, d_model=256, T=16, S=256, num_factored_vocabs=2,
                       qk_norm=False, use_mup=False)


# "GENIE_138M": config.json is only on the HF hub; L=32, d=512 is the only n{L}_h{H}_d{d} shape that
# rounds to 138M parameters (137,545,216).  H=8 (head_dim 64) is INFERRED -- every report says so.
def c138() -> GenieConfig:
    return GenieConfig(num_layers=32, num_heads=8, d_model=512, T=16, S=256, num_factored_vocabs=2,
                       qk_norm=False, use_mup=False)
