"""Training step on the HIP path -- counterpart of the reference's train.py:426-441 (optimizer + grouping),
:468-492 (learning-rate schedules) and :600-633 (forward, backward, clip, step), with the data-parallel gradient
exchange accelerate/DDP does for it.

    trainer = GenieTrainer(model, lr=1e-4, weight_decay=0.0, max_grad_norm=1.0)
    out = trainer.train_step(collate_fn(features))        # dict(loss, acc, grad_norm, lr)

Design (MI355X-first):
  * parameters, gradients and both Adam moments live in four flat f32 buffers laid out in the order gradients become
    READY during the backward (readout, layers L-1..0, embeddings); the model's nn.Parameters are re-pointed at views
    of the parameter buffer, so `model.state_dict()` / `save_pretrained` keep working;
  * every activation the backward needs is kept (288 GB of HBM; nothing is recomputed except the spatial softmax);
  * multi-GPU: one process per GPU, clips sharded, gradients summed with one RCCL all-reduce per bucket of layers,
    launched as soon as that bucket's backward has been enqueued so the exchange overlaps the remaining backward;
    1/world_size and the clip coefficient are folded into the AdamW kernel (no host sync in the step);
  * precisions: the model's (`exact` f32 MFMA; `f16x3` split-f16 MFMA with f32-class gradients; `bf16` = what
    accelerate's bf16 autocast computes); parameters, gradients and Adam moments are f32 in all of them;
  * GPU only: there is no CPU fallback for the model math (the collator in data.py is device-agnostic data prep).
MuAdamW (--mu_transfer, train.py:439; third-party mup fork) is not built.
"""
import ctypes as C
import math

import torch
import torch.distributed as dist

from . import _lib


# ------------------------------------------------------------------ schedules (train.py:468-492)
def lr_factor_custom_cosine(warmup_steps, max_steps, end_ratio=0.1):
    """train.py:468-477: linear warm-up to 1, cosine decay to `end_ratio`."""
    def f(step):
        if step < warmup_steps:
            return (step + 1) / warmup_steps
        remaining = max_steps - warmup_steps
        return ((1 + math.cos(math.pi * (step - warmup_steps) / remaining)) / 2) * (1 - end_ratio) + end_ratio
    return f


def lr_factor_linear(warmup_steps, max_steps):
    """transformers.get_scheduler("linear", ...) -- the default --lr_scheduler_type (train.py:483-492)."""
    def f(step):
        if step < warmup_steps:
            return step / max(1, warmup_steps)
        return max(0.0, (max_steps - step) / max(1, max_steps - warmup_steps))
    return f


def decays(name: str) -> bool:
    """train.py:427-437: names containing "bias" or "layer_norm.weight" are excluded from weight decay.  No GENIE
    parameter is called layer_norm.* (they are norm1/norm2/norm), so LayerNorm weights DO decay, as in the reference."""
    return not ("bias" in name or "layer_norm.weight" in name)


# ------------------------------------------------------------------ pointer tables
def _ptr(t):
    return None if t is None else t.data_ptr()


LINEAR_WEIGHTS = ("spatial_attn.qkv", "spatial_attn.proj", "temporal_attn.qkv", "temporal_attn.proj", "mlp.fc1", "mlp.fc2")


def linear_weight_names(config):
    """The nn.Linear weights that have 16-bit operand copies in the bf16 / f16x3 precisions."""
    out = [f"decoder.layers.{i}.{n}.weight" for i in range(config.num_layers) for n in LINEAR_WEIGHTS]
    return out + ["out_x_proj.weight"]


def weights_table(config, tensors, w16=None):
    """genie_weights table over state-dict-named tensors (parameters or gradients); `w16` (optional) maps Linear weight
    names to their 16-bit copies (*_w16 members).  Returns (table, keepalive)."""
    L = config.num_layers
    layers = (_lib.LayerWeights * L)()
    g = tensors.get
    h = (w16 or {}).get
    for i in range(L):
        p = f"decoder.layers.{i}."
        lw = layers[i]
        lw.norm1_w, lw.norm1_b = _ptr(g(p + "norm1.weight")), _ptr(g(p + "norm1.bias"))
        lw.norm2_w, lw.norm2_b = _ptr(g(p + "norm2.weight")), _ptr(g(p + "norm2.bias"))
        for name, aw in (("spatial_attn.", lw.spatial), ("temporal_attn.", lw.temporal)):
            aw.qkv_w, aw.qkv_b = _ptr(g(p + name + "qkv.weight")), _ptr(g(p + name + "qkv.bias"))
            aw.proj_w, aw.proj_b = _ptr(g(p + name + "proj.weight")), _ptr(g(p + name + "proj.bias"))
            aw.norm_w, aw.norm_b = _ptr(g(p + name + "norm.weight")), _ptr(g(p + name + "norm.bias"))
            aw.qkv_w16, aw.proj_w16 = _ptr(h(p + name + "qkv.weight")), _ptr(h(p + name + "proj.weight"))
        lw.fc1_w, lw.fc1_b = _ptr(g(p + "mlp.fc1.weight")), _ptr(g(p + "mlp.fc1.bias"))
        lw.fc2_w, lw.fc2_b = _ptr(g(p + "mlp.fc2.weight")), _ptr(g(p + "mlp.fc2.bias"))
        lw.fc1_w16, lw.fc2_w16 = _ptr(h(p + "mlp.fc1.weight")), _ptr(h(p + "mlp.fc2.weight"))
    w = _lib.Weights()
    w.pos_embed = _ptr(g("pos_embed_TSC"))
    w.mask_embed = _ptr(g("token_embed.mask_token_embed"))
    for j in range(config.num_factored_vocabs):
        w.embed[j] = _ptr(g(f"token_embed.factored_embeds.{j}.weight"))
    w.out_w, w.out_b = _ptr(g("out_x_proj.weight")), _ptr(g("out_x_proj.bias"))
    w.out_w16 = _ptr(h("out_x_proj.weight"))
    w.layers_host = layers
    return w, layers


def ready_order(config, names):
    """Parameter names in the order their gradients are final during the backward: readout, layers L-1..0 (within a
    layer: decaying tensors first, so each layer is two AdamW ranges), then the embedding side."""
    names = list(names)
    head = [n for n in names if n.startswith("out_x_proj.")]
    emb = [n for n in names if n.startswith("token_embed.") or n == "pos_embed_TSC"]
    out = sorted(head, key=lambda n: not decays(n))
    for i in reversed(range(config.num_layers)):
        p = f"decoder.layers.{i}."
        lay = [n for n in names if n.startswith(p)]
        out += sorted(lay, key=lambda n: not decays(n))
    out += emb
    assert sorted(out) == sorted(names), "unexpected parameter names"
    return out


class BucketReducer:
    """Sum-all-reduce of contiguous slices of one flat gradient buffer, one slice per bucket, asynchronously.

    `ready(hi)` says "gradients [0, hi) are final (their kernels are enqueued on the current stream)"; every bucket
    that is now complete is handed to the process group at once (RCCL orders itself after the current stream), so
    the exchange of late layers runs under the backward of early ones.  `finish()` makes the current stream wait for
    all of them.  Works on any backend (gloo in the CPU tests)."""

    def __init__(self, flat, bounds, group=None, always=False):
        # always: hand the buckets to the process group even at world size 1 (a one-GPU box exercising RCCL itself)
        self.flat, self.bounds, self.group, self.always = flat, list(bounds), group, always
        self.reset()

    def reset(self):
        self.next, self.works = 0, []

    @property
    def world(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def ready(self, hi):
        if self.world == 1 and not (self.always and dist.is_available() and dist.is_initialized()):
            return
        while self.next < len(self.bounds) and self.bounds[self.next][1] <= hi:
            lo, up = self.bounds[self.next]
            self.works.append(dist.all_reduce(self.flat[lo:up], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.next += 1

    def finish(self):
        self.ready(self.flat.numel())
        for w in self.works:
            w.wait()
        self.reset()


def bucket_bounds(segments, target_elems):
    """Greedy contiguous buckets over [(lo, hi)] segments (in ready order) of at least `target_elems` elements."""
    out, start, end = [], None, None
    for lo, hi in segments:
        if start is None:
            start = lo
        end = hi
        if end - start >= target_elems:
            out.append((start, end))
            start = None
    if start is not None:
        out.append((start, end))
    return out


class GenieTrainer:
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0,
                 gradient_accumulation_steps=1, lr_lambda=None, bucket_mb=64, group=None):
        self.model, self.config = model, model.config
        dev = model._device()  # raises on CPU: no fallback
        self.lib = _lib.load()
        self.precision = model.precision
        self.cfg = _lib.make_cfg(self.config, model._prec)
        self.base_lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.accum = gradient_accumulation_steps
        self.lr_lambda = lr_lambda or (lambda step: 1.0)
        self.completed_steps, self._micro = 0, 0

        named = dict(model.named_parameters())
        self.order = ready_order(self.config, named.keys())
        offs, o = {}, 0
        for n in self.order:
            offs[n] = (o, o + named[n].numel())
            o += (named[n].numel() + 63) // 64 * 64  # 256-byte aligned tensors (float4 / MFMA staging loads)
        self.n_flat = o
        self.params = torch.zeros(o, dtype=torch.float32, device=dev)
        self.grads = torch.zeros_like(self.params)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.offsets = offs
        with torch.no_grad():
            for n in self.order:
                lo, hi = offs[n]
                view = self.params[lo:hi].view(named[n].shape)
                view.copy_(named[n].data)
                named[n].data = view  # the module now reads (and checkpoints) the flat buffer
        # every replica starts from rank 0's parameters, as DistributedDataParallel does when it wraps the module
        # (the reference relies on that broadcast: train.py:441 `accelerator.prepare`); the trainer itself only
        # all-reduces gradients, so replicas that start apart would stay apart
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.broadcast(self.params, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        model.refresh_weights()
        self.p_views = {n: self.params[offs[n][0]:offs[n][1]] for n in self.order}
        self.g_views = {n: self.grads[offs[n][0]:offs[n][1]].view(named[n].shape) for n in self.order}
        # 16-bit operand copies of the Linear weights, both orientations (bf16: 1 plane, f16x3: [hi | lo])
        self.w16 = self.w16T = None
        self.wT_table = None
        if self.precision != "exact":
            npl = 1 if self.precision == "bf16" else 2
            lin = linear_weight_names(self.config)
            self.w16 = {n: torch.empty((npl,) + tuple(named[n].shape), dtype=torch.int16, device=dev) for n in lin}
            self.w16T = {n: torch.empty((npl,) + tuple(named[n].shape)[::-1], dtype=torch.int16, device=dev) for n in lin}
            self.wT_table, self._wTk = weights_table(self.config, {}, self.w16T)
        self.w_table, self._wk = weights_table(self.config, {n: named[n].data for n in self.order}, self.w16)
        self.g_table, self._gk = weights_table(self.config, self.g_views)
        self.pack_weights()
        # AdamW ranges: maximal runs of consecutive tensors with the same decay flag (padding between tensors is zero
        # and stays zero: zero gradient, zero moments)
        runs = []
        for n in self.order:
            lo, hi = offs[n]
            hi_pad = lo + (hi - lo + 63) // 64 * 64
            if runs and runs[-1][2] == decays(n):
                runs[-1][1] = hi_pad
            else:
                runs.append([lo, hi_pad, decays(n)])
        self.adam_runs = [(lo, hi, dk) for lo, hi, dk in runs]
        # all-reduce buckets: per-layer segments merged up to bucket_mb
        segs, L = [], self.config.num_layers
        groups = [[n for n in self.order if n.startswith("out_x_proj.")]]
        groups += [[n for n in self.order if n.startswith(f"decoder.layers.{i}.")] for i in reversed(range(L))]
        groups += [[n for n in self.order if n.startswith("token_embed.") or n == "pos_embed_TSC"]]
        for gnames in groups:
            lo = offs[gnames[0]][0]
            hi = offs[gnames[-1]][0] + (named[gnames[-1]].numel() + 63) // 64 * 64
            segs.append((lo, hi))
        self.segments = segs  # [head, layer L-1, ..., layer 0, embeddings]
        self.reducer = BucketReducer(self.grads, bucket_bounds(segs, bucket_mb * (1 << 20) // 4), group)
        self.sums = torch.zeros(3, dtype=torch.float64, device=dev)
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
        self.scratch = torch.zeros(1024, dtype=torch.float64, device=dev)
        self._acts = self._ws = None
        self._B = 0

    def pack_weights(self):
        """Refresh the 16-bit weight copies from the f32 parameters (no-op in the exact precision).  Call this (or
        `sync_external_weights`) after ANY weight mutation the trainer did not make itself -- `model.load_state_dict`,
        `load_numpy_state_dict` -- or the 16-bit GEMM operands keep the old weights until the next optimizer step."""
        if self.precision != "exact":
            _lib.check(self.lib.genie_train_pack_weights(self.cfg, self.w_table, self.w_table, self.wT_table,
                                                         self._stream()), "genie_train_pack_weights")

    def sync_external_weights(self):
        """After the caller changed the model's parameters in place (they are views of the flat buffer): repack the
        16-bit operand copies here and the inference-side copies of the module."""
        self.pack_weights()
        self.model.refresh_weights()

    # ------------------------------------------------------------------ buffers
    def _buffers(self, B):
        if B != self._B:
            self._acts = self._ws = None
            na = self.lib.genie_train_activation_bytes(self.cfg, B)
            nw = self.lib.genie_train_workspace_bytes(self.cfg, B)
            if na == 0 or nw == 0:
                _lib.check(self.lib.genie_check_config(self.cfg), "genie_check_config")
            self._acts = torch.empty(na, dtype=torch.uint8, device=self.params.device)
            self._ws = torch.empty(nw, dtype=torch.uint8, device=self.params.device)
            self._B = B
        return self._acts, self._ws

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    # ------------------------------------------------------------------ forward / backward (train.py:611-617)
    def forward_backward(self, input_ids, labels, accumulate=False, reduce=True):
        """One micro-batch: loss/acc of STMaskGIT.forward and gradients into the flat buffer (added when
        `accumulate`).  Returns (loss, acc) as 0-dim float64 CUDA tensors (no host sync)."""
        if not (input_ids.is_cuda and labels.is_cuda):
            raise RuntimeError("1xgpt_amd runs on the GPU only (no CPU fallback): move the batch to cuda")
        ids = input_ids.to(torch.int64).contiguous()
        lab = labels.to(torch.int64).contiguous()
        B = ids.shape[0]
        assert ids.shape == lab.shape == (B, self.config.T * self.config.S), ids.shape
        acts, ws = self._buffers(B)
        lib, cfg, st = self.lib, self.cfg, self._stream()
        acc_flag = 1 if accumulate else 0
        _lib.check(lib.genie_train_forward(cfg, self.w_table, ids.data_ptr(), lab.data_ptr(), B, acts.data_ptr(),
                                           acts.numel(), self.sums.data_ptr(), st), "genie_train_forward")
        sums = self.sums.clone()
        self.reducer.reset()
        _lib.check(lib.genie_train_backward_head(cfg, self.w_table, self.wT_table, self.g_table, B, acts.data_ptr(),
                                                 ws.data_ptr(), ws.numel(), acc_flag, st), "genie_train_backward_head")
        if reduce:
            self.reducer.ready(self.segments[0][1])
        L = self.config.num_layers
        for k, layer in enumerate(reversed(range(L))):
            _lib.check(lib.genie_train_backward_layer(cfg, self.w_table, self.wT_table, self.g_table, layer, B,
                                                      acts.data_ptr(), ws.data_ptr(), ws.numel(), acc_flag, st),
                       "genie_train_backward_layer")
            if reduce:
                self.reducer.ready(self.segments[1 + k][1])
        _lib.check(lib.genie_train_backward_embed(cfg, self.g_table, ids.data_ptr(), B, ws.data_ptr(), ws.numel(),
                                                  acc_flag, st), "genie_train_backward_embed")
        if reduce:
            self.reducer.finish()
        return sums[0] / sums[2], sums[1] / sums[2]

    # ------------------------------------------------------------------ clip + AdamW + scheduler (train.py:628-633)
    def grad_sumsq(self):
        self.sumsq.zero_()
        _lib.check(self.lib.genie_sumsq(self.grads.data_ptr(), self.n_flat, self.sumsq.data_ptr(),
                                        self.scratch.data_ptr(), self._stream()), "genie_sumsq")
        return self.sumsq

    def current_lr(self):
        return self.base_lr * self.lr_lambda(self.completed_steps)

    def optimizer_step(self):
        world = self.reducer.world
        mult = 1.0 / (world * self.accum)
        ss = self.grad_sumsq()
        lr = self.current_lr()
        clip = self.max_grad_norm if self.max_grad_norm is not None else 0.0
        step = self.completed_steps + 1
        st = self._stream()
        for lo, hi, dk in self.adam_runs:
            off = lo * 4
            _lib.check(self.lib.genie_adamw_step(
                self.params.data_ptr() + off, self.grads.data_ptr() + off, self.exp_avg.data_ptr() + off,
                self.exp_avg_sq.data_ptr() + off, hi - lo, lr, self.betas[0], self.betas[1], self.eps,
                self.weight_decay if dk else 0.0, step, mult, ss.data_ptr() if clip > 0 else None, clip, st),
                "genie_adamw_step")
        self.completed_steps += 1
        self.pack_weights()
        self.model.refresh_weights()  # the module's own 16-bit copies (inference entry points) are stale
        return torch.sqrt(ss[0]) * mult, lr

    def train_step(self, batch):
        """One micro-batch of the reference loop (train.py:604-633); the optimizer runs every
        `gradient_accumulation_steps` calls.  Returns device scalars (no host sync)."""
        is_update = (self._micro + 1) % self.accum == 0
        loss, acc = self.forward_backward(batch["input_ids"], batch["labels"], accumulate=self._micro % self.accum != 0,
                                          reduce=is_update and self.accum == 1)
        self._micro += 1
        out = {"loss": loss, "acc": acc}
        if is_update:
            if self.accum > 1 and self.reducer.world > 1:  # no_sync() micro-batches: reduce once, after the last
                self.reducer.reset()
                self.reducer.finish()
            out["grad_norm"], out["lr"] = self.optimizer_step()
        return out

    def gradients(self):
        """{state-dict name: gradient view} (shapes of the parameters)."""
        return self.g_views

    # ------------------------------------------------------------------ optimizer checkpoint (train.py:560-590 resume)
    def state_dict(self):
        """Optimizer state in torch.optim.AdamW's vocabulary, keyed by parameter NAME (the model's own weights are
        saved by `model.save_pretrained` / `state_dict`)."""
        named = dict(self.model.named_parameters())
        st = {}
        for n in self.order:
            lo, hi = self.offsets[n]
            st[n] = {"exp_avg": self.exp_avg[lo:hi].view(named[n].shape).clone(),
                     "exp_avg_sq": self.exp_avg_sq[lo:hi].view(named[n].shape).clone()}
        return {"state": st, "completed_steps": self.completed_steps, "micro_step": self._micro,
                "hyper": {"lr": self.base_lr, "betas": tuple(self.betas), "eps": self.eps,
                          "weight_decay": self.weight_decay, "max_grad_norm": self.max_grad_norm,
                          "gradient_accumulation_steps": self.accum}}

    def load_state_dict(self, sd):
        """Inverse of `state_dict` (moments, step counters); hyper-parameters stay those of this trainer (a warning
        names every saved value that differs)."""
        mine = {"lr": self.base_lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                "max_grad_norm": self.max_grad_norm, "gradient_accumulation_steps": self.accum}
        diff = {k: (v, mine[k]) for k, v in (sd.get("hyper") or {}).items()
                if k in mine and (tuple(v) if isinstance(v, (list, tuple)) else v) != mine[k]}
        if diff:
            import warnings
            warnings.warn("GenieTrainer.load_state_dict: resuming with different hyper-parameters (saved, current): "
                          + ", ".join(f"{k}={a}->{b}" for k, (a, b) in diff.items()))
        for n in self.order:
            lo, hi = self.offsets[n]
            self.exp_avg[lo:hi].copy_(sd["state"][n]["exp_avg"].reshape(-1))
            self.exp_avg_sq[lo:hi].copy_(sd["state"][n]["exp_avg_sq"].reshape(-1))
        self.completed_steps = int(sd["completed_steps"])
        self._micro = int(sd.get("micro_step", 0))
        self.pack_weights()
