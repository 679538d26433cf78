"""GEMM ablation (diagnostics): kernels with the LDS-DMA, the ds_read+MFMA part or the DMA wait removed."""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
L = s2v._lib
from microbench import timeit  # noqa
DEV = "cuda:0"
M, N, K = 38400, 12288, 3072
A = (torch.randn(M, K, device=DEV) * 0.5).bfloat16()
W = (torch.randn(N, K, device=DEV) * 0.02).bfloat16()
b = torch.zeros(N, device=DEV, dtype=torch.bfloat16)
C = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
f = lambda: L.check(L.diag_lib().s2v_op_linear(L.ptr(A), L.ptr(W), L.ptr(b), L.ptr(C), M, N, K, 0, 1, 0, L.stream_ptr()))
cases = [("pp64 full", 7), ("pp64 no-DMA", 7 | (1 << 8)), ("pp64 no-MFMA", 7 | (3 << 8)), ("pp64 no-setprio", 7 | (7 << 8))]
for rep in range(2):
    for name, code in cases:
        L.diag_lib().s2v_set_gemm_impl(code)
        ms = timeit(f, iters=8)
        print(f"{name:22s}: {ms:7.3f} ms  ({2*M*N*K/ms/1e9:7.1f} TFLOP/s equivalent)", flush=True)
