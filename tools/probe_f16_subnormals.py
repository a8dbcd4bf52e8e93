#!/usr/bin/env python3
"""Probe (GPU): do gfx950's f16 matrix instructions take SUBNORMAL f16 inputs exactly?  The packed operand split of the GEMM
epilogues (csrc/common.hpp split_f16_x4) does not flush a subnormal hi plane, which is only correct if they do.  Prints the f16x3
Linear of a constant subnormal operand (A side, then W side) next to the exact value; measured on MI355X: exact down to 2^-24."""
import importlib, sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
