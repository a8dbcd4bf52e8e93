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
import os
import random
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import Dataset as TorchDataset


class RawTokenDataset(TorchDataset):
    def __init__(self, data_dir, window_size, stride=1, filter_interrupts=True, filter_overlaps=False):
        data_dir = Path(data_dir)
        with open(data_dir / "metadata.json") as f:
            self.metadata = json.load(f)
        shape = (self.metadata["num_images"], self.metadata["s"], self.metadata["s"])
        video_tokens_path, segment_ids_path = data_dir / "video.bin", data_dir / "segment_ids.bin"
        token_dtype = np.dtype(self.metadata.get("token_dtype", "uint32"))
        self.data = np.memmap(video_tokens_path, dtype=token_dtype, mode="r", shape=shape)
        if os.path.isfile(segment_ids_path):
            self.segment_ids = np.memmap(segment_ids_path, dtype=np.int32, mode="r",
                                         shape=(self.metadata["num_images"],))
        else:
            self.segment_ids = None
            if filter_interrupts:
                raise NotImplementedError("Cannot filter interrupted sequences without segment ids.")
        self.window_size, self.stride = window_size, stride
        # frames between the first and last frame of a window (excluding one endpoint)
        self.video_len = (self.window_size - 1) * self.stride

        n_starts = max(len(self.data) - self.video_len, 0)
        starts = np.arange(n_starts)
        if filter_interrupts and n_starts:
            seg = np.asarray(self.segment_ids)
            starts = starts[seg[starts] == seg[starts + self.video_len]]
        self.valid_start_inds = starts.tolist()

        if filter_overlaps:
            # greedy, in order: keep a start unless one of the kept starts of the last window_size*stride
            # entries lies exactly i*stride before it (i = 1..window_size-1), i.e. shares a frame with it
            kept = []
            for start_ind in self.valid_start_inds:
                overlapping = {start_ind - i * self.stride for i in range(1, self.window_size)}
                recent = kept[-self.window_size * self.stride:]
                if not any(k in overlapping for k in recent):
                    kept.append(start_ind)
            self.valid_start_inds = kept

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
