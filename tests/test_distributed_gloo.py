"""world_size-2 gloo test (CPU) of the data-parallel metric reduction: sharded partial sums all-reduced to the
same whole-job means on every rank, equal to the single-process AvgMetric result (eval_utils.py:16-25)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO, pkg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from conftest import pkg as _pkg
    D = _pkg("distributed")
    r, w, _ = D.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    # per-clip metric contributions of a fixed synthetic job of 10 clips
    g = np.random.default_rng(0)
    n_clips, tok = 10, 15 * 256
    ce = g.random(n_clips) * 3 + 12
    hits = g.integers(0, tok, n_clips)
    lo, hi = D.shard_range(n_clips, rank, world)
    sums = torch.tensor([ce[lo:hi].sum() * tok, (hi - lo) * tok, hits[lo:hi].sum(), (hi - lo) * tok,
                         (hi - lo) * 15, hi - lo], dtype=torch.float64)
    sums, secs = D.reduce_metric_sums(sums, seconds=1.0 + rank)
    D.barrier()
    q.put((rank, D.means_from_sums(sums.tolist()), secs))
    torch.distributed.destroy_process_group()


def test_gloo_world2_metric_reduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = np.random.default_rng(0)
    ce = g.random(10) * 3 + 12
    hits = g.integers(0, 15 * 256, 10)
    AvgMetric = pkg("eval_utils").AvgMetric
    am = AvgMetric()
    for c in ce:
        am.update(float(c), 1)
    for rank, m, secs in res:
        assert abs(m["loss"] - am.mean()) < 1e-12
        assert abs(m["acc"] - hits.sum() / (10 * 15 * 256)) < 1e-12
        assert m["frames"] == 150 and m["clips"] == 10
        assert secs == 2.0  # MAX over ranks
