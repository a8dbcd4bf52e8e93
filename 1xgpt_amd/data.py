"""RawTokenDataset -- reader of the 1X token dataset layout (counterpart of the reference's data.py:17-106).

On disk: ``metadata.json`` {num_images, s, vocab_size, hz, token_dtype?}, ``video.bin`` = (num_images, s, s) tokens
(uint32 by default), optional ``segment_ids.bin`` = (num_images,) int32.  Windows of ``window_size`` frames spaced
``stride`` apart; ``filter_interrupts`` drops windows whose first and last frame belong to different segments,
``filter_overlaps`` keeps each frame in at most one window.  Same constructor, attributes (``data``,
``metadata``, ``valid_start_inds``) and item dict as the reference.  ``get_maskgit_collator`` is the training
collator of data.py:109-169 (random corruption + MaskGIT masking) on whatever device the clips are on.
"""
import json
import math
import random
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import Dataset as TorchDataset


def _window_starts(n_frames, span, segment_ids=None):
    """First-frame indices of all windows that fit (their last frame is `span` frames later); with `segment_ids`, only windows
    that begin and end in the same recording segment."""
    starts = np.arange(max(n_frames - span, 0))
    if segment_ids is not None and len(starts):
        seg = np.asarray(segment_ids)
        starts = starts[seg[starts] == seg[starts + span]]
    return starts.tolist()


def _drop_shared_frames(starts, window_size, stride):
    """Walk the (increasing) window starts and keep one only if it shares no frame with a window kept before: two windows on
    the same stride grid share a frame exactly when their starts differ by stride, 2 stride, ..., (window_size - 1) stride.
    (Equivalent to the reference's scan over its most recent kept entries, data.py:72-88: kept starts are distinct integers, so
    every one close enough to collide is among them.)"""
    kept, seen = [], set()
    for s0 in starts:
        if not any((s0 - i * stride) in seen for i in range(1, window_size)):
            kept.append(s0)
            seen.add(s0)
    return kept


class RawTokenDataset(TorchDataset):
    def __init__(self, data_dir, window_size, stride=1, filter_interrupts=True, filter_overlaps=False):
        root = Path(data_dir)
        self.metadata = json.loads((root / "metadata.json").read_text())
        n, side = self.metadata["num_images"], self.metadata["s"]
        self.data = np.memmap(root / "video.bin", mode="r", shape=(n, side, side),
                              dtype=np.dtype(self.metadata.get("token_dtype", "uint32")))
        seg_file = root / "segment_ids.bin"
        self.segment_ids = np.memmap(seg_file, dtype=np.int32, mode="r", shape=(n,)) if seg_file.is_file() else None
        if filter_interrupts and self.segment_ids is None:
            raise NotImplementedError("Cannot filter interrupted sequences without segment ids.")
        self.window_size, self.stride = window_size, stride
        self.video_len = (window_size - 1) * stride          # distance from a window's first frame to its last
        self.valid_start_inds = _window_starts(len(self.data), self.video_len, self.segment_ids if filter_interrupts else None)
        if filter_overlaps:
            self.valid_start_inds = _drop_shared_frames(self.valid_start_inds, window_size, stride)

    def __len__(self):
        return len(self.valid_start_inds)

    def __getitem__(self, idx):
        start_ind = self.valid_start_inds[idx]
        x = torch.from_numpy((self.data[start_ind: start_ind + self.video_len + 1: self.stride]).astype(np.int64))
        x = x.flatten()
        return {"input_ids": x, "labels": x, "attention_mask": torch.ones_like(x)}

    def batch(self, idxs):
        """Stack several windows -> (len(idxs), window_size * s * s) int64 (what default_data_collator builds)."""
        return torch.stack([self[i]["input_ids"] for i in idxs])


def write_token_dataset(data_dir, tokens: np.ndarray, segment_ids: np.ndarray = None, hz=30, vocab_size=262144,
                        token_dtype="uint32", extra_metadata=None):
    """Write (num_images, s, s) tokens in the dataset layout (used by generate and by the tests)."""
    data_dir = Path(data_dir)
    data_dir.mkdir(parents=True, exist_ok=True)
    tokens = np.asarray(tokens)
    tokens.astype(np.dtype(token_dtype)).tofile(data_dir / "video.bin")
    if segment_ids is not None:
        np.asarray(segment_ids, dtype=np.int32).tofile(data_dir / "segment_ids.bin")
    meta = {"num_images": int(tokens.shape[0]), "s": int(tokens.shape[1]), "vocab_size": vocab_size, "hz": hz,
            "token_dtype": token_dtype}
    meta.update(extra_metadata or {})
    with open(data_dir / "metadata.json", "w") as f:
        json.dump(meta, f)
    return meta


# ------------------------------------------------------------------ MaskGIT training collator (data.py:109-169)
class TorchDraws:
    """The collator's random draws, in the reference's call order, from torch's / Python's global generators."""

    def __init__(self, device):
        self.device = device

    def rand(self, shape):
        return torch.rand(tuple(shape), device=self.device)

    rand_like = rand

    def randint(self, high, shape):
        return torch.randint(low=0, high=high, size=tuple(shape), dtype=torch.long, device=self.device)

    def py_random(self):
        return random.random()

    def py_randint(self, a, b):
        return random.randint(a, b)

    def py_uniform(self, a, b):
        return random.uniform(a, b)


def maskgit_collate(input_ids, config, draws=None):
    """(B, T*S) int64 clips -> {"input_ids", "labels"} following data.py:112-167 draw for draw.

    `draws` replays captured draws (parity tests); None draws fresh ones on the clips' device."""
    ids = input_ids.to(torch.int64)
    dev = ids.device
    draws = draws or TorchDraws(dev)
    B = ids.shape[0]
    h = w = math.isqrt(config.S)
    nv, Vf = config.num_factored_vocabs, config.factored_vocab_size
    mask_token_id = config.image_vocab_size
    x_THW = ids.reshape(B, config.T, h, w)
    powers = Vf ** torch.arange(nv, device=dev)
    x_THWC = (x_THW.unsqueeze(-1) // powers) % Vf
    labels = x_THW.clone()

    def t(a, dtype):
        return torch.as_tensor(a, device=dev).to(dtype)

    r = t(draws.rand(x_THWC.shape), torch.float32)
    u01 = t(draws.rand(()), torch.float32)
    random_values = t(draws.randint(Vf, x_THWC.shape), torch.long)
    m = r < config.max_corrupt_rate * u01
    x_THWC = torch.where(m, random_values, x_THWC)
    if draws.py_random() < config.non_mlm_ratio:
        first = draws.py_randint(config.num_prompt_frames, config.T - 1)
        correct_rate = draws.py_uniform(0.25, 1.0)
        for i in range(config.T - first):
            correct_rate *= draws.py_uniform(0.9, 1.0)
            r = t(draws.rand((B, h, w, nv)), torch.float32)
            m = r > correct_rate
            x_THWC[:, first + i] = torch.where(m, random_values[:, first + i], x_THWC[:, first + i])
    else:
        first = 1
    while True:
        u = t(draws.rand((B, config.T - first, 1, 1)), torch.float32)
        prob = torch.cos(u * torch.pi / 2)
        r = t(draws.rand_like((B, config.T - first, h, w)), torch.float32)
        mask = r < prob
        if bool(mask.any()):
            break
    x = (x_THWC * powers).sum(-1)
    x[:, first:][mask] = mask_token_id
    return {"input_ids": x.reshape(B, -1), "labels": labels.reshape(B, -1)}


def get_maskgit_collator(config):
    """collate_fn(features: list of {"input_ids": (T*S,) tensor}) -> batch dict, as data.py:109."""
    def collate_fn(features):
        return maskgit_collate(torch.stack([ex["input_ids"] for ex in features]), config)
    return collate_fn
