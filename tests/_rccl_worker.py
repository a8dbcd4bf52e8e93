"""Child process of tests/test_hip_multirank.py::test_rccl_world1 (not collected by pytest): a world-size-1 process group on the
"nccl" backend (= RCCL on ROCm) on the box's one GPU, and the path's two collectives pushed through it on device tensors --
the six-sum f64 metric all-reduce of evaluate (reference: accelerator.reduce, train.py:635) and the trainer's bucketed gradient
all-reduce.  Prints one JSON line."""
import importlib
import json
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    D = importlib.import_module("1xgpt_amd.distributed")
    T = importlib.import_module("1xgpt_amd.train")
    rank, world, local_rank = D.init_distributed(backend="nccl", force_group=True)
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    dev = torch.device("cuda", D.local_device_index(local_rank))
    sums = torch.tensor([14.5, 3840.0, 3000.0, 3840.0, 15.0, 1.0], dtype=torch.float64, device=dev)
    before = sums.clone()
    out, secs = D.reduce_metric_sums(sums, seconds=1.25, always=True)
    torch.cuda.synchronize()
    flat = torch.arange(3 * 1024 * 1024, dtype=torch.float32, device=dev)
    ref = flat.clone()
    red = T.BucketReducer(flat, T.bucket_bounds([(0, 1 << 20), (1 << 20, 2 << 20), (2 << 20, 3 << 20)], 1 << 20), always=True)
    red.ready(1 << 20)          # first bucket goes out while "later layers" would still be running
    n_early = len(red.works)
    red.finish()
    torch.cuda.synchronize()
    D.barrier()
    maps = open("/proc/self/maps").read()
    res = {"backend": dist.get_backend(), "world": dist.get_world_size(), "sums_equal": bool(torch.equal(out, before)),
           "seconds": secs, "buckets_early": n_early, "grads_equal": bool(torch.equal(flat, ref)),
           "librccl_mapped": "librccl" in maps, "libgenie": os.path.realpath(importlib.import_module("1xgpt_amd._lib").LIB_PATH)}
    dist.destroy_process_group()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
