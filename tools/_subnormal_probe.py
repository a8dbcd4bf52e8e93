import importlib, sys, torch
sys.path.insert(0, "/root/repo")
_lib = importlib.import_module("1xgpt_amd._lib"); lib = _lib.load()
M, N, K = 16384, 1536, 512
st = torch.cuda.current_stream().cuda_stream
for val in (2e-5, 3e-6, 6e-8):
    x16 = torch.zeros(2, M, K, dtype=torch.float16, device="cuda"); x16[0] = val          # hi plane subnormal, lo = 0
    W16 = torch.zeros(2, N, K, dtype=torch.float16, device="cuda"); W16[0] = 1.0
    y = torch.empty(M, N, device="cuda")
    _lib.check(lib.genie_linear_lowp(_lib.PREC_F16X3, x16.data_ptr(), W16.data_ptr(), 0, y.data_ptr(), M, N, K, 0, 0, st), "lin")
    exp = float(torch.tensor(val).half().float()) * K
    print(f"A subnormal {val:g}: y = {y[0,0].item():.6e}  expected {exp:.6e}")
    # subnormal on the W side (hi plane of W gets multiplied by 2048 in registers: becomes normal)
    x16[0] = 1.0; W16[0] = val
    _lib.check(lib.genie_linear_lowp(_lib.PREC_F16X3, x16.data_ptr(), W16.data_ptr(), 0, y.data_ptr(), M, N, K, 0, 0, st), "lin")
    print(f"W subnormal {val:g}: y = {y[0,0].item():.6e}  expected {exp:.6e}")
