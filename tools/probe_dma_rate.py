"""Per-CU LDS-DMA rate probe: the DMA-only ablation of gemm_bf16_pp64 on grids of 64 / 128 / 256 / 1024 tiles."""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
L = s2v._lib
from microbench import timeit  # noqa
DEV = "cuda:0"
for (M, N, K) in ((4096, 4096, 1024), (4096, 4096, 3072), (4096, 4096, 12288), (8192, 8192, 1024), (8192, 8192, 4096), (2048, 2048, 12288)):
    A = (torch.randn(M + 256, K, device=DEV) * 0.5).bfloat16()
    W = (torch.randn(N, K, device=DEV) * 0.02).bfloat16()
    b = torch.zeros(N, device=DEV, dtype=torch.bfloat16)
    C = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    f = lambda: L.check(L.diag_lib().s2v_op_linear(L.ptr(A), L.ptr(W), L.ptr(b), L.ptr(C), M, N, K, 0, 1, 0, L.stream_ptr()))
    tiles = (M // 256) * (N // 256)
    for name, code in (("full", 7), ("no-MFMA", 7 | (3 << 8)), ("no-DMA", 7 | (1 << 8))):
        L.diag_lib().s2v_set_gemm_impl(code)
        ms = timeit(f, iters=20)
        by = tiles * 2 * 256 * K * 2
        print(f"tiles={tiles:5d} K={K:5d} {name:8s}: {ms:7.3f} ms  DMA {by/ms/1e9:7.2f} TB/s = {by/ms/1e6/min(tiles,256)/2.1:6.1f} B/clk/CU@2.1GHz  ({2*M*N*K/ms/1e9:7.1f} TF)", flush=True)
L.diag_lib().s2v_set_gemm_impl(7)
