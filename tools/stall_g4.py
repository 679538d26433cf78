"""Cycle accounting of gemm_g4 (diagnostics library): s_memtime stamps of workgroup 100, wave 0 -- prologue, K loop per K-tile, epilogue --
and the shader clock from s_memtime / s_memrealtime (100 MHz).  Usage: python tools/stall_g4.py"""
import ctypes
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
L = s2v._lib
DEV = "cuda:0"
M = 38400
lib = L.diag_lib()
lib.s2v_g4_debug_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
lib.s2v_set_gemm_g4t(0)  # this tool reads gemm_g4's own stamps: keep the persistent gemm_g4t out of the routing
for name, N, K, epi in (("qkv", 9216, 3072, 0), ("out", 3072, 3072, 0), ("ff1+gelu", 12288, 3072, 1), ("ff2", 3072, 12288, 0)):
    A = (torch.randn(M, K, device=DEV) * 0.5).bfloat16()
    W = (torch.randn(N, K, device=DEV) * 0.02).bfloat16()
    b = torch.zeros(N, device=DEV, dtype=torch.bfloat16)
    C = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    lib.s2v_set_gemm_impl(9)
    for _ in range(2):
        L.check(lib.s2v_op_linear(L.ptr(A), L.ptr(W), L.ptr(b), L.ptr(C), M, N, K, epi, 1, 0, L.stream_ptr()))
    torch.cuda.synchronize()
    d = (ctypes.c_ulonglong * 8)()
    lib.s2v_g4_debug_read(d, 1)
    L.check(lib.s2v_op_linear(L.ptr(A), L.ptr(W), L.ptr(b), L.ptr(C), M, N, K, epi, 1, 0, L.stream_ptr()))
    torch.cuda.synchronize()
    lib.s2v_g4_debug_read(d, 1)
    nT, n = K // 64, max(1, d[3])
    pro, loop, epi_c = d[0] / n, d[1] / n, d[2] / n
    tot = pro + loop + epi_c
    clk = d[4] / max(1, d[5]) * 100.0
    print(f"{name:9s} {n:5d} workgroups, K-tiles {nT:4d}: mean {tot:9.0f} cycles = prologue {pro:6.0f} + K loop {loop:9.0f} ({loop / nT:7.1f} per K-tile; 2065 of MFMA) + "
          f"epilogue {epi_c:7.0f} ({100.0 * epi_c / tot:4.1f} %); shader clock {clk:6.0f} MHz", flush=True)
    lib.s2v_set_gemm_impl(7)
