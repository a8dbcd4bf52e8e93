"""Keeps the 4096^3 f16x3 GEMM (gemm16_pp) running for ~12 s on zero-filled or random operands and prints its rate: the load
under which tools/gpu_power_probe.sh samples board power and shader clock."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_lib = importlib.import_module("1xgpt_amd._lib")
zero = "--zero" in sys.argv
shape = [int(a) for a in sys.argv[1:] if a.isdigit()]   # optional: M N K (default 4096 4096 4096)
lib = _lib.load()
M, N, K = shape if len(shape) == 3 else (4096, 4096, 4096)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(M, K, device="cuda", generator=g); W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
if zero:
    x.zero_(); W.zero_()
b = torch.zeros(N, device="cuda"); y = torch.empty(M, N, device="cuda")
st = torch.cuda.current_stream().cuda_stream
x16 = torch.empty(2, M, K, dtype=torch.float16, device="cuda"); W16 = torch.empty(2, N, K, dtype=torch.float16, device="cuda")
_lib.check(lib.genie_pack_split_f16(x.data_ptr(), x16.data_ptr(), x.numel(), st), "pack")
_lib.check(lib.genie_pack_split_f16(W.data_ptr(), W16.data_ptr(), W.numel(), st), "pack")
def call():
    _lib.check(lib.genie_linear_lowp(_lib.PREC_F16X3, x16.data_ptr(), W16.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, 0, 0, st), "lin")
call(); torch.cuda.synchronize()
t0 = time.time(); n = 0
while time.time() - t0 < 12.0:
    for _ in range(200):
        call()
    torch.cuda.synchronize(); n += 200
dt = time.time() - t0
print(f"M={M} N={N} K={K}", ("zero" if zero else "random"), f"operands: {2.0 * M * N * K * n / dt / 1e12:.1f} TFLOP/s algorithmic (x3 MFMA issue) over {dt:.1f} s")
