"""Per-shape timing of the MAGVIT2 3x3 implicit-GEMM conv (genie_conv3x3_gn_bf16 / genie_conv3x3_bf16) on the layer
geometries of the shipped VQConfig (base 128, ch_mult (1,1,2,2,4)): TFLOP/s per shape, random bf16 operands."""
import argparse, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import torch
_lib = importlib.import_module("1xgpt_amd._lib")

SHAPES = [  # (H, W, Cin, Cout, d2s)
    (256, 256, 128, 128, 0), (128, 128, 128, 128, 0), (128, 128, 128, 512, 1), (64, 64, 256, 256, 0),
    (64, 64, 256, 128, 0), (64, 64, 128, 512, 1), (32, 32, 256, 256, 0), (32, 32, 256, 1024, 1), (16, 16, 512, 512, 0),
    (16, 16, 512, 256, 0), (16, 16, 256, 1024, 1), (256, 256, 128, 8, 0),
]

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--gn", type=int, default=1)
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    zero = torch.zeros(4096, dtype=torch.bfloat16, device=dev)
    out = []
    for H, W, ci, co, d2s in SHAPES:
        n = a.frames
        x = torch.randn(n, H, W, ci, device=dev).to(torch.bfloat16)
        w = (torch.randn(co, 9 * ci, device=dev) * (9 * ci) ** -0.5).to(torch.bfloat16)
        b = torch.randn(co, device=dev)
        y = torch.empty(n * H * W * co, dtype=torch.bfloat16, device=dev)
        part = torch.empty(lib.genie_conv_gn_part_floats(n, H, W, co), dtype=torch.float32, device=dev)
        def run():
            if a.gn:
                rc = lib.genie_conv3x3_gn_bf16(x.data_ptr(), w.data_ptr(), b.data_ptr(), 0, y.data_ptr(), zero.data_ptr(), n, H, W,
                                               ci, co, d2s, 1, part.data_ptr(), 32, st)
                if rc == 0:
                    return "gn"
            _lib.check(lib.genie_conv3x3_bf16(x.data_ptr(), w.data_ptr(), b.data_ptr(), 0, y.data_ptr(), zero.data_ptr(), n, H, W,
                                              ci, co, d2s, st), "conv")
            return "plain"
        kind = run(); run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        fl = 2.0 * n * H * W * co * 9 * ci
        r = {"H": H, "W": W, "Cin": ci, "Cout": co, "d2s": d2s, "kind": kind, "us": round(us, 1), "tflops": round(fl / us / 1e6, 1)}
        print(r, flush=True)
        out.append(r)
    print(json.dumps({"frames": a.frames, "results": out}))

if __name__ == "__main__":
    main()
