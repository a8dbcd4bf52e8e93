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


# ---------------------------------------------------------------------------------------------------------------
# gradient exchange of the training step: bucketed, asynchronous SUM all-reduce over one flat buffer
# ---------------------------------------------------------------------------------------------------------------
def _train_cfg():
    return pkg("config").GenieConfig(num_layers=2, num_heads=2, d_model=32, T=4, S=16, num_factored_vocabs=2,
                                     qk_norm=False, num_prompt_frames=2)


def _shard_grads(rank, world):
    """Per-rank gradients of the oracle on this rank's clips, flattened in the trainer's ready order."""
    from oracle import genie_train_oracle as TO
    cfg = _train_cfg()
    syn, T = pkg("synthetic"), pkg("train")
    sd = syn.make_state_dict(cfg, seed=3, law="conditioned")
    ids = syn.make_clips(4, cfg, seed=21)
    batch = TO.maskgit_collate(ids, cfg, TO.NumpyDraws(8))
    lo, hi = pkg("distributed").shard_range(4, rank, world)
    _, _, g = TO.forward_backward(batch["input_ids"][lo:hi], batch["labels"][lo:hi], sd, cfg)
    order = T.ready_order(cfg, sd.keys())
    return cfg, order, g


def _flatten(order, g):
    offs, o = {}, 0
    for n in order:
        offs[n] = (o, o + g[n].size)
        o += (g[n].size + 63) // 64 * 64
    flat = torch.zeros(o, dtype=torch.float32)
    for n in order:
        flat[offs[n][0]:offs[n][1]] = torch.from_numpy(g[n].reshape(-1))
    return flat, offs


def _grad_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from conftest import pkg as _pkg
    D, T = _pkg("distributed"), _pkg("train")
    D.init_distributed(backend="gloo")
    cfg, order, g = _shard_grads(rank, world)
    flat, offs = _flatten(order, g)
    # segments in ready order: head, layer 1, layer 0, embeddings; tiny buckets so that several are in flight
    groups = [[n for n in order if n.startswith("out_x_proj.")]]
    groups += [[n for n in order if n.startswith(f"decoder.layers.{i}.")] for i in (1, 0)]
    groups += [[n for n in order if n.startswith("token_embed.") or n == "pos_embed_TSC"]]
    segs = [(offs[gn[0]][0], (offs[gn[-1]][1] + 63) // 64 * 64) for gn in groups]
    red = T.BucketReducer(flat, T.bucket_bounds(segs, 4096))
    assert red.world == world
    for lo, hi in segs:  # "this segment's backward has been enqueued"
        red.ready(hi)
    red.finish()
    q.put((rank, flat.numpy().copy()))
    torch.distributed.destroy_process_group()


def test_gloo_world2_gradient_buckets():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    flats = []
    for r in range(world):
        _, order, g = _shard_grads(r, world)
        flats.append(_flatten(order, g)[0].numpy())
    want = flats[0] + flats[1]
    assert np.abs(want).max() > 0
    for r in range(world):
        assert np.array_equal(res[r], want)  # identical on every rank, every element reduced exactly once


def test_ready_order_and_buckets():
    T = pkg("train")
    cfg = _train_cfg()
    names = [k for k, *_ in pkg("synthetic").state_dict_spec(cfg)]
    order = T.ready_order(cfg, names)
    assert sorted(order) == sorted(names)
    assert order[0].startswith("out_x_proj.") and order[-1] in ("pos_embed_TSC", "token_embed.factored_embeds.1.weight")
    first_l0 = min(i for i, n in enumerate(order) if n.startswith("decoder.layers.0."))
    last_l1 = max(i for i, n in enumerate(order) if n.startswith("decoder.layers.1."))
    assert last_l1 < first_l0  # layer 1's gradients are final before layer 0's
    lay = [n for n in order if n.startswith("decoder.layers.1.")]
    flags = [T.decays(n) for n in lay]
    assert flags == sorted(flags, reverse=True)  # decaying tensors first: two AdamW ranges per layer
    assert T.decays("decoder.layers.0.norm1.weight") and not T.decays("decoder.layers.0.mlp.fc1.bias")
    b = T.bucket_bounds([(0, 100), (100, 250), (250, 260), (260, 1000)], 200)
    assert b == [(0, 250), (250, 1000)]
    assert T.bucket_bounds([(0, 10)], 1 << 20) == [(0, 10)]
