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


@torch.no_grad()
def generate_frames_cached(model, example_THW: torch.LongTensor, num_prompt_frames=8, maskgit_steps=2, temperature=0.0,
                           teacher_force_time=False, noise=None, unmask_mode="random", merge_commit=True, host_loop=False):
    """``generate_frames`` with a temporal KV cache (genie_frame_pass): every pass runs ONE frame through the stack
    against the cached temporal keys/values of the earlier frames instead of the full 16-frame forward --
    one P-frame pass for the prompt + (T-P)*(steps+1) single-frame passes (= 2 full-pass equivalents at P=8, steps=2) instead of (T-P)*steps full
    forwards (16).  Same outputs (per-row arithmetic is unchanged).
    merge_commit: where the library covers it (genie_frames_pass: f16x3, heads of 64 or 32, up to 16,384 rows per pass) the pass that commits frame t's
    final tokens also carries MaskGIT step 0 of frame t+1, so a frame costs `steps` passes instead of `steps + 1`.
    host_loop: False = the whole loop is ONE library call (genie_generate_cached: every pass, sampling and mask step enqueued
    without a host step in between); True = the same loop driven from Python (one C-ABI call per pass / sample / mask step)."""
    import math
    from . import _lib
    lib = _lib.load()
    cfg, w = model._weights()[:2]
    ex = example_THW.to(torch.int64).contiguous()
    B, T = ex.shape[0], ex.shape[1]
    S, V = model.config.S, model.config.factored_vocab_size * model.config.num_factored_vocabs
    P = num_prompt_frames
    assert P <= T and P >= 1
    dev = ex.device
    ids = ex.view(B, T, S)
    ws = model._workspace(B, generate_prompt_frames=P)
    nbytes = lib.genie_prefix_cache_bytes(cfg, B)
    cache = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    if not host_loop and P < T:
        steps = int(maskgit_steps)
        if unmask_mode not in ("random", "greedy"):
            raise NotImplementedError(f"Expected `unmask_mode` to be one of ['greedy', 'random'], got {unmask_mode}")
        nz = None
        if steps > 1 and unmask_mode == "random":   # the draws of torch.rand_like (st_mask_git.py:204-206): the caller's, or fresh ones
            nz = (torch.rand(T - P, steps - 1, B, S, device=dev) if noise is None
                  else noise.to(dev)[:, :steps - 1].reshape(T - P, steps - 1, B, S).float().contiguous())
        uni = torch.rand(T - P, steps, model.config.num_factored_vocabs, B, S, device=dev) if temperature > 1e-8 else None
        gen = torch.empty(B, T - P, S, dtype=torch.int64, device=dev)
        _lib.check(lib.genie_generate_cached(cfg, w, ids.data_ptr(), B, P, T - P, steps, float(temperature),
                                             _lib.UNMASK_GREEDY if unmask_mode == "greedy" else _lib.UNMASK_RANDOM,
                                             0 if nz is None else nz.data_ptr(), 0 if uni is None else uni.data_ptr(),
                                             int(bool(teacher_force_time)), int(bool(merge_commit)), gen.data_ptr(), 0, cache.data_ptr(),
                                             nbytes, ws.data_ptr(), ws.numel(), st), "genie_generate_cached")
        outputs = torch.cat([ex[:, :P], gen.view(B, T - P, model.h, model.w)], dim=1)
        return torch.cat([outputs, ex[:, P:]], dim=1)

    def frame_pass(tokens_BS, t, logits=None):
        _lib.check(lib.genie_frame_pass(cfg, w, tokens_BS.data_ptr(), B, t, cache.data_ptr(), nbytes,
                                        0 if logits is None else logits.data_ptr(), ws.data_ptr(), ws.numel(), st),
                   "genie_frame_pass")

    def commit_and_open(final_BS, mask_BS, t, logits):
        """Commit the final tokens of frame t AND run MaskGIT step 0 of frame t + 1 (all-mask tokens) in ONE two-frame pass
        (genie_frames_pass): frame t + 1 attends the slot the same pass writes.  False = the library does not cover two
        frames per pass for this model / batch (nothing was enqueued): the caller runs the two passes one by one."""
        two = torch.stack([final_BS, mask_BS], dim=1).contiguous()
        rc = lib.genie_frames_pass(cfg, w, two.data_ptr(), B, t, 2, cache.data_ptr(), nbytes, logits.data_ptr(), ws.data_ptr(),
                                   ws.numel(), st)
        if rc == _lib.E_UNSUPPORTED:
            return False
        _lib.check(rc, "genie_frames_pass")
        return True

    # the prompt fills slots 0..P-1 of the cache in ONE P-frame pass: on the fragment-order kernels where they cover the model
    # (genie_frames_pass with nf = P), else genie_clean_pass with the cache's T-frame layout; geometries neither covers fill the
    # slots frame by frame
    rc = _lib.E_UNSUPPORTED
    if P > 1:
        rc = lib.genie_frames_pass(cfg, w, ids[:, :P].contiguous().data_ptr(), B, 0, P, cache.data_ptr(), nbytes, 0, ws.data_ptr(),
                                   ws.numel(), st)
        if rc == _lib.E_UNSUPPORTED:
            rc = lib.genie_clean_pass(cfg, w, ids[:, :P].contiguous().data_ptr(), B, P, T, cache.data_ptr(), nbytes, ws.data_ptr(),
                                      ws.numel(), st)
    if rc == _lib.E_UNSUPPORTED:
        for t in range(P):
            frame_pass(ids[:, t].contiguous(), t)
    else:
        _lib.check(rc, "genie_frames_pass / genie_clean_pass (prompt)")
    logits = torch.empty(B, S, V, dtype=torch.float32, device=dev)
    samples = torch.empty(B, S, dtype=torch.int64, device=dev)
    conf = torch.empty(B, S, dtype=torch.float32, device=dev)
    gen = []
    opened = False   # step 0 of the current frame already ran inside the previous frame's commit pass
    for k, t in enumerate(range(P, T)):
        cur = torch.full((B, S), model.mask_token_id, dtype=torch.int64, device=dev)
        unmasked = torch.zeros(B, S, dtype=torch.uint8, device=dev)
        for step in range(maskgit_steps):
            if not (step == 0 and opened):
                frame_pass(cur, t, logits)
            uni = torch.rand(model.config.num_factored_vocabs, B, S, device=dev) if temperature > 1e-8 else None
            _lib.check(lib.genie_sample(cfg, logits.data_ptr(), _lib.LAYOUT_TOKEN_MAJOR, B, float(temperature),
                                        0 if uni is None else uni.data_ptr(), samples.data_ptr(), conf.data_ptr(), st),
                       "genie_sample")
            last = step == maskgit_steps - 1
            keys, n = None, 0
            if not last:
                n = math.ceil(math.cos((step + 1) / maskgit_steps * math.pi / 2) * S)
                if unmask_mode == "greedy":
                    keys = conf
                elif noise is None:
                    keys = torch.rand(B, S, device=dev)
                else:
                    keys = noise[k][step].to(dev).reshape(B, S).float().contiguous()
            _lib.check(lib.genie_mask_step(0 if keys is None else keys.data_ptr(), n, int(last), model.mask_token_id,
                                           unmasked.data_ptr(), samples.data_ptr(), cur.data_ptr(), S, B, S, st),
                       "genie_mask_step")
        gen.append(cur.view(B, model.h, model.w))
        opened = False
        if t + 1 < T:  # commit frame t (its final tokens, or the ground truth when teacher-forcing in time)
            final = ids[:, t].contiguous() if teacher_force_time else cur
            if merge_commit:
                opened = commit_and_open(final, torch.full_like(cur, model.mask_token_id), t, logits)
                merge_commit = opened   # (unsupported once = unsupported for the whole call)
            if not opened:
                frame_pass(final, t)
    outputs = torch.cat([ex[:, :P], torch.stack(gen, dim=1)], dim=1)
    return torch.cat([outputs, ex[:, P:]], dim=1)


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
