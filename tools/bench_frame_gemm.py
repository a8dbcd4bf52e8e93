"""Linear micro-benchmark at frame-pass sizes (M = 256 * clips * frames rows): the fragment-order kernels of csrc/kernels_frame.hip
(register-direct `fr`, LDS-tiled `frm`) beside the row-major dispatcher (genie_linear_lowp: gemm16_sm / nt / v2 / pp), f16x3.
Weights are rotated over enough copies that every call streams them from HBM, as a pass over 32 layers does."""
import argparse, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_lib = importlib.import_module("1xgpt_amd._lib")
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, nargs="+", default=[256, 512, 1024, 2048, 4096, 8192])
ap.add_argument("--copies", type=int, default=24)
a = ap.parse_args()
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
for M in a.rows:
    for name, N, K in (("qkv", 1536, 512), ("proj", 512, 512), ("fc1", 2048, 512), ("fc2", 512, 2048)):
        x = torch.randn(M, K, device="cuda")
        b = torch.randn(N, device="cuda"); y = torch.zeros(M, N, device="cuda")
        Ws = [torch.randn(N, K, device="cuda") / K ** 0.5 for _ in range(a.copies)]
        x_rm = torch.empty(2, M, K, dtype=torch.float16, device="cuda"); x_fr = torch.empty(2 * M * K, dtype=torch.float16, device="cuda")
        _lib.check(lib.genie_pack_split_f16(x.data_ptr(), x_rm.data_ptr(), x.numel(), st), "pack")
        _lib.check(lib.genie_pack_frame_w16(x.data_ptr(), x_fr.data_ptr(), M, K, st), "pack")
        W_rm, W_fr = [], []
        for W in Ws:
            r = torch.empty(2, N, K, dtype=torch.float16, device="cuda"); f = torch.empty(2 * N * K, dtype=torch.float16, device="cuda")
            _lib.check(lib.genie_pack_split_f16(W.data_ptr(), r.data_ptr(), W.numel(), st), "pack")
            _lib.check(lib.genie_pack_frame_w16(W.data_ptr(), f.data_ptr(), N, K, st), "pack")
            W_rm.append(r); W_fr.append(f)
        def timed(call, reps=96):
            for i in range(8): call(i % a.copies)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(reps): call(i % a.copies)
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / reps
        res = {}
        res["row-major"] = timed(lambda i: _lib.check(lib.genie_linear_lowp(_lib.PREC_F16X3, x_rm.data_ptr(), W_rm[i].data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, 0, 0, st), "lin"))
        ref = y.clone()
        for mode, tag in ((1, "fr"), (2, "frm")):
            if (mode == 1 and K > 512) or (mode == 2 and M % 128):
                continue
            rc = lib.genie_frame_linear(x_fr.data_ptr(), W_fr[0].data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, mode, st)
            if rc != 0:
                continue
            res[tag] = timed(lambda i: _lib.check(lib.genie_frame_linear(x_fr.data_ptr(), W_fr[i].data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, mode, st), "frl"))
            _lib.check(lib.genie_linear_lowp(_lib.PREC_F16X3, x_rm.data_ptr(), W_rm[(96 - 1) % a.copies].data_ptr(), b.data_ptr(), ref.data_ptr(), M, N, K, 0, 0, st), "lin")
            res[tag + "_maxdiff"] = (y - ref).abs().max().item()
        print(f"M={M:5d} {name:5s} N={N:4d} K={K:4d} " + " ".join(f"{k} {v:8.1f}us" if not k.endswith("diff") else f"{k} {v:.1e}" for k, v in res.items()), flush=True)
