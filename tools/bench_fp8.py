"""fp8 (W8A8, e4m3) vs bf16 GEMM at the C3 shapes: TFLOP/s of the GEMM alone and with the activation quantisation pass."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
L = s2v._lib
DEV = "cuda:0"


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


M = 38400
for name, N, K, epi in (("qkv", 9216, 3072, 0), ("out", 3072, 3072, 0), ("ff1+gelu", 12288, 3072, 1), ("ff2", 3072, 12288, 0)):
    A = (torch.randn(M, K, device=DEV) * 0.5).bfloat16()
    W = (torch.randn(N, K, device=DEV) * 0.02).bfloat16()
    b = torch.zeros(N, device=DEV, dtype=torch.bfloat16)
    C = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    need = M * K + N * K + 4 * (M + N)
    scratch = torch.empty(need, dtype=torch.uint8, device=DEV)
    f8 = lambda: L.check(L.lib().s2v_op_linear_fp8(L.ptr(A), L.ptr(W), L.ptr(b), L.ptr(C), M, N, K, epi, L.ptr(scratch), need, L.stream_ptr()))
    f16 = lambda: L.check(L.lib().s2v_op_linear(L.ptr(A), L.ptr(W), L.ptr(b), L.ptr(C), M, N, K, epi, 1, 0, L.stream_ptr()))
    t8, t16 = timeit(f8), timeit(f16)
    fl = 2 * M * N * K
    print(f"{name:9s} M={M} N={N} K={K}: fp8 incl. quantising A and W {t8:7.3f} ms ({fl/t8/1e9:7.1f} TFLOP/s)   bf16 {t16:7.3f} ms ({fl/t16/1e9:7.1f} TFLOP/s)", flush=True)
