"""Prompted generation harness -- counterpart of the reference's genie/generate.py:62-116.

Prompt ``num_prompt_frames`` frames, MaskGIT-decode the remaining ``window_size - num_prompt_frames`` frames
autoregressively (or teacher-forced in time), and write ``[prompt | generated | ground truth]`` as ``video.bin``
+ ``metadata.json`` in the dataset layout.  Batched: the reference generates one example, this takes (B, T, H, W).
The reference's ``--teacher_force_time`` branch reads a non-existent ``model.image_mask_token`` (generate.py:86);
here it uses ``mask_token_id``, which is what that line means.
"""
import json
from pathlib import Path

import numpy as np
import torch

STRIDE = 15


@torch.no_grad()
def generate_frames(model, example_THW: torch.LongTensor, num_prompt_frames=8, maskgit_steps=2, temperature=0.0,
                    teacher_force_time=False, noise=None):
    """example_THW (B, T, H, W) on the model's device -> outputs (B, T + (T - num_prompt_frames), H, W):
    [prompt frames | predicted frames | ground-truth frames] (generate.py:97-103).
    noise: optional (T - num_prompt_frames, maskgit_steps-1, B, S)."""
    window_size = example_THW.shape[1]
    assert num_prompt_frames <= window_size
    example_THW = example_THW.to(torch.int64).contiguous()
    samples = []
    prompt_THW = example_THW.clone()
    prompt_THW[:, num_prompt_frames:] = model.mask_token_id
    for k, timestep in enumerate(range(num_prompt_frames, window_size)):
        if teacher_force_time:
            prompt_THW = example_THW.clone()
            prompt_THW[:, timestep:] = model.mask_token_id
        samples_HW, _ = model.maskgit_generate(prompt_THW, out_t=timestep, maskgit_steps=maskgit_steps,
                                               temperature=temperature, noise=None if noise is None else noise[k],
                                               return_logits=False)
        samples.append(samples_HW)
        if not teacher_force_time:
            prompt_THW[:, timestep] = samples_HW  # autoregressive (already written in place by maskgit_generate)
    outputs = torch.stack(samples, dim=1)
    outputs = torch.cat([example_THW[:, :num_prompt_frames], outputs], dim=1)
    return torch.cat([outputs, example_THW[:, num_prompt_frames:]], dim=1)


def write_outputs(outputs_THW: torch.LongTensor, output_dir, dataset_metadata: dict, args: dict):
    """video.bin (token_dtype of the source dataset) + metadata.json with the reference's extra keys
    (generate.py:105-116).  outputs for ONE example: (1, n, H, W) or (n, H, W)."""
    output_dir = Path(output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    out = outputs_THW.reshape(-1, *outputs_THW.shape[-2:]).cpu().numpy()
    out.astype(np.dtype(dataset_metadata.get("token_dtype", "uint32"))).tofile(output_dir / "video.bin")
    side = int(out.shape[-1])
    meta = dict(args) | dict(dataset_metadata) | {"num_images": int(out.shape[0]), "h": side, "w": side,
                                                   "t": int(args.get("window_size", 16))}
    with open(output_dir / "metadata.json", "w") as f:
        json.dump(meta, f)
    return meta
