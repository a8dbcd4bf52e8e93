"""Data-parallel plumbing for the evaluation path: one process per GPU, clips sharded, one all-reduce.

Clips are independent (no cross-clip op anywhere on the path), so the only
exchange is the metric reduction the reference's trainer does with ``accelerator.reduce`` (train.py:635):
an all-reduce(SUM) of a handful of float64 partial sums, then mean = sum / count exactly as
``AvgMetric`` weights batches (eval_utils.py:16-25).  On ROCm the "nccl" backend is RCCL over xGMI; the
message is <= 64 bytes, i.e. latency-bound.  The same code runs on "gloo" for the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment; returns (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("GENIE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kwargs = {}
        if backend == "nccl":
            dev_index = local_device_index(local_rank)
            torch.cuda.set_device(dev_index)
            kwargs["device_id"] = torch.device("cuda", dev_index)
        if backend == "gloo":
            # gloo announces its connections on STDOUT; callers (bench.py) print exactly one JSON line there
            import sys
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank


def local_device_index(local_rank: int) -> int:
    """GPU of this rank: LOCAL_RANK, unless GENIE_FORCE_DEVICE pins every rank to one device (single-GPU smoke runs
    of the multi-process path with the gloo backend)."""
    forced = os.environ.get("GENIE_FORCE_DEVICE")
    return int(forced) if forced is not None else local_rank


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of n_items for `rank`; sizes differ by at most one, earlier ranks larger."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def reduce_metric_sums(sums: torch.Tensor, seconds: float = None):
    """all-reduce(SUM) a vector of partial sums in place; optionally all-reduce(MAX) the wall time.
    No-op for a single process."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        if seconds is not None:
            t = torch.tensor([seconds], dtype=torch.float64, device=sums.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            seconds = float(t.item())
    return sums, seconds


def means_from_sums(sums):
    """[sum CE, n CE, sum hits, n tokens, n frames, n clips] -> dict(loss, acc, frames, clips)."""
    s = [float(v) for v in sums]
    return dict(loss=s[0] / s[1], acc=s[2] / s[3], frames=int(round(s[4])), clips=int(round(s[5])))


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
