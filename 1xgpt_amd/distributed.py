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


HIP_INIT_STALL_RC = 17   # exit code of a rank whose FIRST GPU touch did not return in time (see init_device)
RDZV_TIMEOUT_RC = 18     # exit code of a rank whose rendezvous did not complete in time (a peer never arrived)


def _stall_exit(what, code):
    """Runs on a timer thread while the main thread is stuck: dump every thread's stack and leave with `code`.  The process
    has not completed a GPU call yet (or is waiting in the rendezvous), so ending it is safe; its supervisor -- bench.py's
    per-rank parent, which never touches the GPU -- starts a FRESH process once.  Nothing is ever re-exec'ed in place."""
    import faulthandler
    import sys
    print(f"1xgpt_amd.distributed: rank {os.environ.get('RANK', '0')}: {what} -- giving up this process (exit {code})",
          file=sys.stderr, flush=True)
    try:
        faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
    except Exception:
        pass
    os._exit(code)


def init_device(local_rank: int, world: int = 1):
    """First GPU touch of this process, made robust for multi-rank starts (one of six 2-rank starts on a
    fresh box hung before the second rank's HIP initialisation returned):
      * staggered: rank r waits r * GENIE_HIP_INIT_STAGGER seconds (default 0.5) so that the ranks of a node do not open
        the driver at the same instant;
      * watched: if the touch (device query, set_device, a 1-element allocation, a synchronise) has not returned after
        GENIE_HIP_INIT_TIMEOUT seconds (default 180; the minutes-long first `import torch` on a fresh box happens before the
        clock starts) the process dumps its stacks and exits with HIP_INIT_STALL_RC instead of hanging its peers until a
        collective timeout.
    Returns the device index, or None when no GPU is visible (CPU / gloo test runs)."""
    import threading
    import time
    if world > 1:
        time.sleep(float(os.environ.get("GENIE_HIP_INIT_STAGGER", "0.5")) * local_rank)
    limit = float(os.environ.get("GENIE_HIP_INIT_TIMEOUT", "180"))
    timer = threading.Timer(limit, _stall_exit, (f"GPU initialisation did not return within {limit:.0f} s", HIP_INIT_STALL_RC))
    timer.daemon = True
    if limit > 0:
        timer.start()
    try:
        if not torch.cuda.is_available():
            return None
        dev_index = local_device_index(local_rank)
        torch.cuda.set_device(dev_index)
        torch.empty(1, device=torch.device("cuda", dev_index))
        torch.cuda.synchronize(dev_index)
        return dev_index
    finally:
        timer.cancel()


def _is_timeout(e: BaseException) -> bool:
    """A rendezvous that ran out of time: c10d raises DistStoreError / DistNetworkError / TimeoutError / RuntimeError with
    'timed out' or 'timeout' somewhere in the text, depending on which layer noticed."""
    if isinstance(e, TimeoutError):
        return True
    text = f"{type(e).__name__} {e}".lower()
    return "timeout" in text or "timed out" in text


def init_distributed(backend=None, force_group=False):
    """Initialise torch.distributed from the torchrun environment; returns (rank, world_size, local_rank).
    Order: this rank's GPU first (init_device: staggered, watched), then the rendezvous.  Only the RENDEZVOUS is bounded by
    GENIE_RDZV_TIMEOUT seconds (default 600; on expiry the rank exits with RDZV_TIMEOUT_RC rather than waiting for the
    backend's 30-minute default): once the group exists its timeout is set back to GENIE_COLLECTIVE_TIMEOUT seconds (default
    1800), so ranks may reach a later all-reduce or barrier far apart (uneven evaluation shards, a leg that runs on rank 0 only).
    force_group (or GENIE_DIST_FORCE_GROUP=1): create the process group even at world size 1 -- a one-GPU box can then push
    the path's collectives through RCCL itself (tests/test_hip_multirank.py)."""
    import datetime
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_group = force_group or os.environ.get("GENIE_DIST_FORCE_GROUP", "0") == "1"
    if backend != "gloo" or os.environ.get("GENIE_FORCE_DEVICE") is not None:
        init_device(local_rank, world)   # (a pure-CPU gloo run has nothing to initialise)
    if (world > 1 or force_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("GENIE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        limit = float(os.environ.get("GENIE_RDZV_TIMEOUT", "600"))
        kwargs = {}
        if backend == "nccl":
            dev_index = local_device_index(local_rank)
            torch.cuda.set_device(dev_index)
            kwargs["device_id"] = torch.device("cuda", dev_index)

        def rendezvous():
            try:
                dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                        timeout=datetime.timedelta(seconds=limit), **kwargs)
            except Exception as e:  # store / rendezvous timeout: a peer never arrived
                if _is_timeout(e):
                    _stall_exit(f"rendezvous did not complete within {limit:.0f} s ({type(e).__name__}: {e})", RDZV_TIMEOUT_RC)
                raise
            # the bound above was for the rendezvous only: give the group back the backend's collective timeout (30 minutes)
            try:
                from torch.distributed import distributed_c10d as c10d
                c10d._set_pg_timeout(datetime.timedelta(seconds=float(os.environ.get("GENIE_COLLECTIVE_TIMEOUT", "1800"))))
            except Exception as e:   # (private helper: if a torch release drops it the short timeout stays -- say so)
                import sys
                print(f"1xgpt_amd.distributed: collectives keep the {limit:.0f} s rendezvous timeout ({type(e).__name__}: {e})",
                      file=sys.stderr)

        if backend == "gloo":
            # gloo announces its connections on STDOUT; callers (bench.py) print exactly one JSON line there
            import sys
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                rendezvous()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
        else:
            rendezvous()
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


def reduce_metric_sums(sums: torch.Tensor, seconds: float = None, always: bool = False):
    """all-reduce(SUM) a vector of partial sums in place; optionally all-reduce(MAX) the wall time.
    No-op for a single process unless `always` (then a world-size-1 group still runs the collective: the RCCL smoke test)."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or always):
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
