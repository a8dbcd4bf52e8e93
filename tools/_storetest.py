import importlib, os, sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tools")
import bench_gemm
for (M, N, K) in [(4096, 1024, 512), (2048, 1024, 512), (16384, 1024, 512), (65536, 1024, 512)]:
    for prec in ("f16x3", "bf16"):
        ms, tf, err = bench_gemm.run(prec, M, N, K, iters=5)
        print(prec, M, N, K, f"{ms*1e3:.1f} us {tf:.1f} TF", flush=True)
