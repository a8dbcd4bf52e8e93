"""Teacher-forced evaluation harness -- counterpart of the reference's genie/evaluate.py.

``GenieEvaluator.predict_zframe_logits`` has the reference's contract (evaluate.py:82-122): for every
prefix [frame_0..frame_{t-1}] mask frames >= t and MaskGIT-decode frame t; return the sampled frames
``(B, T-1, H, W)`` and the step-0 factored logits ``(B, 512, 2, T-1, H, W)``.  Differences: the device
is synchronised before timing (the reference does not, evaluate.py:172-175), tokens never leave HBM, and
``evaluate_clips`` adds a data-parallel entry point (one process per GPU, RCCL all-reduce of the metric
sums) that the reference lacks ("only supports a single GPU", evaluate.py:47).

Run:  python -m 1xgpt_amd.evaluate is not importable by name (leading digit); use
      python tools/evaluate.py --checkpoint_dir DIR [--val_data_dir DIR | --synthetic N]
"""
import time

import torch

from . import _lib
from .eval_utils import AvgMetric, compute_loss
from .st_mask_git import STMaskGIT

# Hardcoded values for the v1.1 dataset (evaluate.py:31-32)
WINDOW_SIZE = 16
STRIDE = 15


class GenieEvaluator:
    def __init__(self, args, decode_latents=None, device="cuda", model: STMaskGIT = None):
        if model is None:
            model = STMaskGIT.from_pretrained(args.checkpoint_dir, precision=getattr(args, "precision", "exact"))
        self.model = model.to(device=device)
        self.model.eval()
        self.decode_latents = decode_latents
        self.device = device
        self.args = args

    def predict_zframe_logits(self, input_ids: torch.LongTensor, noise=None, return_logits=True):
        """input_ids (B, T*H*W) -> (samples (B,T-1,H,W), factored logits (B,512,2,T-1,H,W)).

        Total forward passes = (T-1) * maskgit_steps (evaluate.py:90).
        noise: optional (T-1, maskgit_steps-1, B, S) replay of the "random" unmasking draws."""
        m, a = self.model, self.args
        T = m.config.T
        clips = input_ids.to(self.device).to(torch.int64).view(-1, T, a.latent_h, a.latent_w)
        frames, logits = [], []
        for t in range(1, T):
            # timeline t: frames < t are ground truth, frame t and everything after it all-MASK; decode frame t
            timeline = clips.clone()
            timeline[:, t:] = m.mask_token_id
            frame, lg = m.maskgit_generate(timeline, out_t=t, maskgit_steps=a.maskgit_steps, temperature=a.temperature,
                                           noise=None if noise is None else noise[t - 1], return_logits=return_logits,
                                           check=False)
            frames.append(frame)
            logits.append(lg)
        return torch.stack(frames, dim=1), (torch.stack(logits, dim=3) if return_logits else None)

    def predict_next_frames(self, samples_THW) -> torch.Tensor:
        """Sampled tokens -> RGB frames (B, T-1, 3, 256, 256) uint8 through the on-device MAGVIT2 decoder."""
        from .eval_utils import decode_tokens
        if self.decode_latents is None:
            raise RuntimeError("predict_next_frames needs a decode_latents callable (1xgpt_amd.magvit2)")
        return decode_tokens(samples_THW, self.decode_latents)

    # ------------------------------------------------------------------ teacher-forced prefix reuse
    @torch.no_grad()
    def predict_zframe_logits_reuse(self, input_ids: torch.LongTensor, noise=None, return_logits=True,
                                    unmask_mode="random", step0_hook=None):
        """Same contract and same per-row arithmetic as ``predict_zframe_logits`` in (1 + steps) passes over T-1 frames
        instead of 15 * steps forwards over T: the ground-truth frames < t of every timeline t are identical to one clean
        pass (temporal attention is causal, everything else per-frame), so they are computed once (frames 0..T-2: no
        timeline has the last frame as context) and their temporal keys/values cached; "frame t of timeline t" is then
        decoded for t = 1..T-1 together (genie_clean_pass / genie_masked_frames_logits with frame0 = 1).
        Returns (samples (B,T-1,H,W), logits (B,512,2,T-1,H,W)).
        step0_hook(logits_token_major (B,T-1,S,V)): called right after the step-0 pass is enqueued, while the logits buffer
        still holds the step-0 logits (later steps overwrite it in place); with a hook and return_logits=False no copy of the
        (2 GB at 128 clips) logits is kept."""
        import math
        lib = _lib.load()
        m = self.model
        cfg, w = m._weights()[:2]
        T, S = m.config.T, m.config.S
        n = T - 1                                           # timelines 1..T-1 = frame slots 0..n-1 of the masked passes
        V = m.config.factored_vocab_size * m.config.num_factored_vocabs
        steps, temperature = self.args.maskgit_steps, float(self.args.temperature)
        ids = input_ids.to(self.device).to(torch.int64).view(-1, T, S)
        B = ids.shape[0]
        dev = ids.device
        ws = m._workspace(B)
        nbytes = lib.genie_prefix_cache_bytes(cfg, B)
        if getattr(self, "_cache", None) is None or self._cache.numel() < nbytes or self._cache.device != dev:
            self._cache = None
            self._cache = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        cache = self._cache
        st = torch.cuda.current_stream().cuda_stream
        ctx = ids[:, :n].contiguous()                       # ground-truth context frames 0..T-2
        _lib.check(lib.genie_clean_pass(cfg, w, ctx.data_ptr(), B, n, n, cache.data_ptr(), nbytes, ws.data_ptr(), ws.numel(),
                                        st), "genie_clean_pass")
        cur = torch.full((B, n, S), m.mask_token_id, dtype=torch.int64, device=dev)
        unmasked = torch.zeros(B * n, S, dtype=torch.uint8, device=dev)
        samples = torch.empty(B * n, S, dtype=torch.int64, device=dev)
        conf = torch.empty(B * n, S, dtype=torch.float32, device=dev)
        logits = torch.empty(B, n, S, V, dtype=torch.float32, device=dev)
        logits0 = None
        for step in range(steps):
            _lib.check(lib.genie_masked_frames_logits(cfg, w, cur.data_ptr(), B, 1, n, cache.data_ptr(), nbytes,
                                                      logits.data_ptr(), ws.data_ptr(), ws.numel(), st),
                       "genie_masked_frames_logits")
            if step == 0:
                if step0_hook is not None:
                    step0_hook(logits)
                if return_logits or step0_hook is None:
                    logits0 = logits.clone() if steps > 1 else logits
            uni = None
            if temperature > 1e-8:
                uni = torch.rand(m.config.num_factored_vocabs, B * n, S, device=dev)
            _lib.check(lib.genie_sample(cfg, logits.data_ptr(), _lib.LAYOUT_TOKEN_MAJOR, B * n, temperature,
                                        0 if uni is None else uni.data_ptr(), samples.data_ptr(), conf.data_ptr(), st),
                       "genie_sample")
            last = step == steps - 1
            keys, k_unmask = None, 0
            if not last:
                k_unmask = math.ceil(math.cos((step + 1) / steps * math.pi / 2) * S)
                if unmask_mode == "greedy":
                    keys = conf
                elif noise is None:
                    keys = torch.rand(B, n, S, device=dev)
                else:  # reference draw order: one (B,S) tensor per timeline t and step
                    keys = noise[:, step].to(dev).reshape(n, B, S).permute(1, 0, 2).contiguous()
            _lib.check(lib.genie_mask_step(0 if keys is None else keys.data_ptr(), k_unmask, int(last), m.mask_token_id,
                                           unmasked.data_ptr(), samples.data_ptr(), cur.data_ptr(), S, B * n, S, st),
                       "genie_mask_step")
        samples_THW = samples.view(B, n, m.h, m.w)
        self._last_token_major_logits0 = logits0
        if not return_logits:
            return samples_THW, None
        nv, vf = m.config.num_factored_vocabs, m.config.factored_vocab_size
        fl = logits0.reshape(B, n, m.h, m.w, nv, vf).permute(0, 5, 4, 1, 2, 3)  # B Vf nv T-1 H W
        return samples_THW, fl

    @torch.no_grad()
    def evaluate_metric_sums_reuse(self, input_ids, labels=None, noise=None):
        """``evaluate_metric_sums`` on the prefix-reuse path (same six sums)."""
        lib = _lib.load()
        m = self.model
        T, S = m.config.T, m.config.S
        ids = input_ids.to(self.device).to(torch.int64).view(-1, T, S)
        lab = ids if labels is None else labels.to(self.device).to(torch.int64).view(-1, T, S)
        B = ids.shape[0]
        lab = lab.contiguous()
        ids = ids.contiguous()
        st = torch.cuda.current_stream().cuda_stream
        ce = torch.zeros(3, dtype=torch.float64, device=ids.device)
        sums = torch.zeros(6, dtype=torch.float64, device=ids.device)
        cfg = m._weights()[0]

        def ce_of_step0(lg0):  # (B,T-1,S,V) token-major, clip frames 1..T-1; targets are indexed in the full (B,T,S) clip
            _lib.check(lib.genie_factored_ce(cfg, lg0.data_ptr(), _lib.LAYOUT_TOKEN_MAJOR, lab.data_ptr(), 0, B, 1, T,
                                             ce.data_ptr(), st), "genie_factored_ce")

        # the CE is taken from the step-0 logits on the stream BEFORE the next MaskGIT step overwrites them: no 2 GB copy
        samples, _ = self.predict_zframe_logits_reuse(ids, noise=noise, return_logits=False, step0_hook=ce_of_step0)
        # (ground truth frames 1..T-1 == samples).sum() and the vector's sizes, on the device (no torch arithmetic)
        _lib.check(lib.genie_metric_hits(ids.data_ptr() + S * 8, T * S, samples.data_ptr(), (T - 1) * S, B, (T - 1) * S,
                                         ce.data_ptr(), float(B * (T - 1) * S), float(B * (T - 1)), float(B), sums.data_ptr(), st),
                   "genie_metric_hits")
        return sums

    @torch.no_grad()
    def evaluate_metric_sums(self, input_ids, labels=None, noise=None):
        """One batch of the metric loop (evaluate.py:167-179) as device-side sums, no logits materialised
        for the caller: returns float64 tensor [sum CE, n CE tokens, sum (gt == sample), n sampled tokens,
        n frames, n clips]."""
        lib = _lib.load()
        m = self.model
        T, S = m.config.T, m.config.S
        cfg = m._weights()[0]
        ids = input_ids.to(self.device).to(torch.int64).view(-1, T, m.h, m.w).contiguous()
        lab = ids if labels is None else labels.to(self.device).to(torch.int64).view(-1, T, m.h, m.w).contiguous()
        B = ids.shape[0]
        ce = torch.zeros(3, dtype=torch.float64, device=ids.device)
        sums = torch.zeros(6, dtype=torch.float64, device=ids.device)
        stream = torch.cuda.current_stream().cuda_stream
        sizes = (float(B * (T - 1) * S), float(B * (T - 1)), float(B))
        for k, t in enumerate(range(1, T)):
            p = ids.clone()
            p[:, t:] = m.mask_token_id
            s, fl = m.maskgit_generate(p, out_t=t, maskgit_steps=self.args.maskgit_steps,
                                       temperature=self.args.temperature,
                                       noise=None if noise is None else noise[k], check=False)
            # fl is a permuted view of the contiguous (B, V, H, W) step-0 logits of frame t
            lg = fl.permute(0, 2, 1, 3, 4)
            assert lg.is_contiguous()
            _lib.check(lib.genie_factored_ce(cfg, lg.data_ptr(), _lib.LAYOUT_BCTHW, lab.data_ptr(), 0, B, t, t + 1,
                                             ce.data_ptr(), stream), "genie_factored_ce")
            s = s.contiguous()
            # (ids[:, t] == s).sum() accumulated on the device; the last call also files the CE pair and the sizes
            _lib.check(lib.genie_metric_hits(ids.data_ptr() + t * S * 8, T * S, s.data_ptr(), S, B, S,
                                             ce.data_ptr() if t == T - 1 else 0, *sizes, sums.data_ptr(), stream),
                       "genie_metric_hits")
        return sums


@torch.no_grad()
def evaluate_clips(evaluator: GenieEvaluator, clips: torch.LongTensor, batch_size=16, noise_seed=None,
                   distributed=False, reuse=True, clip_offset=0):
    """Metric loop over ``clips`` (N, T*H*W) with the reference's AvgMetric weighting (eval_utils.py:16-25).

    With ``distributed=True`` every rank passes ITS shard of the clips; the six sums are all-reduced (SUM)
    once at the end -- a <= 48-byte message, latency-bound on xGMI -- so the returned means are whole-job
    means, identical on every rank.  Returns dict(loss, acc, frames, clips, seconds, frames_per_sec).
    clip_offset: index of this shard's first clip in the whole job; the "random" unmasking draws of a batch are keyed by
    (noise_seed, global index of its first clip), so a job gives the same draws however it is sharded over ranks (given
    shard boundaries that are multiples of batch_size)."""
    dev = evaluator.device
    total = torch.zeros(6, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m = evaluator.model
    for i in range(0, clips.shape[0], batch_size):
        batch = clips[i:i + batch_size]
        noise = None
        if noise_seed is not None and evaluator.args.maskgit_steps > 1:
            g = torch.Generator(device="cpu").manual_seed(noise_seed + clip_offset + i)
            noise = torch.rand(m.config.T - 1, evaluator.args.maskgit_steps - 1, batch.shape[0], m.config.S,
                               generator=g).to(dev)
        fn = evaluator.evaluate_metric_sums_reuse if reuse else evaluator.evaluate_metric_sums
        total += fn(batch, noise=noise)
    torch.cuda.synchronize()
    seconds = time.perf_counter() - t0
    if distributed:
        import torch.distributed as dist
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        seconds = float(t.item())
    tot = total.tolist()
    return dict(loss=tot[0] / tot[1], acc=tot[2] / tot[3], frames=int(tot[4]), clips=int(tot[5]), seconds=seconds,
                frames_per_sec=tot[4] / seconds)


def run_reference_style_loop(evaluator: GenieEvaluator, batches, verbose=True):
    """The reference's per-batch loop (evaluate.py:167-191) minus LPIPS: gen_time / loss / acc AvgMetrics."""
    from collections import defaultdict
    metrics = defaultdict(AvgMetric)
    T = evaluator.model.config.T
    for batch in batches:
        ids = batch["input_ids"]
        bs = ids.size(0)
        torch.cuda.synchronize()
        start = time.time()
        samples, factored_logits = evaluator.predict_zframe_logits(ids)
        torch.cuda.synchronize()
        frames_per_batch = (T - 1) * bs
        metrics["gen_time"].update((time.time() - start) / frames_per_batch, bs)
        loss = compute_loss(batch["labels"], factored_logits)
        gt = ids.to(samples.device).view(bs, T, *samples.shape[-2:])
        acc = (gt[:, 1:] == samples).float().mean().item()
        metrics["loss"].update(loss, bs)
        metrics["acc"].update(acc, bs)
        if verbose:
            print({key: f"{val.mean():.4f}" for key, val in metrics.items()})
    return metrics
