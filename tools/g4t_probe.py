"""gemm_g4t (persistent, trickled epilogue) against gemm_g4 on the same box: bit-identity and time, on the C3 shapes that qualify.
Usage: python tools/g4t_probe.py [quick]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
L = s2v._lib
DEV = "cuda:0"
D = L.diag_lib()
D.s2v_set_gemm_g4t.argtypes = [__import__("ctypes").c_int]


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


shapes = [("small-gelu", 8192, 4096, 1280, 1), ("small-bias", 8192, 4096, 1536, 0), ("ff1+gelu", 38144, 12288, 3072, 1), ("qkv-plain", 38144, 9216, 3072, 0)]
if "quick" in sys.argv:
    shapes = shapes[:2]
if "ff1" in sys.argv:
    shapes = shapes[2:3]
for name, M, N, K, epi in shapes:
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    A = (torch.randn(M, K, device=DEV, generator=g) * 0.5).bfloat16()
    W = (torch.randn(N, K, device=DEV, generator=g) * 0.05).bfloat16()
    b = (torch.randn(N, device=DEV, generator=g) * 0.2).bfloat16()
    C = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    f = lambda: L.check(D.s2v_op_linear(L.ptr(A), L.ptr(W), L.ptr(b), L.ptr(C), M, N, K, epi, 1, 0, L.stream_ptr()))
    D.s2v_set_gemm_impl(9)
    res = {}
    for on in (0, 1, 1, 0):
        D.s2v_set_gemm_g4t(on)
        C.fill_(float("nan"))
        f()
        torch.cuda.synchronize()
        key = "g4t" if on else "g4"
        if key in res:
            same = torch.equal(res[key], C)
            print(f"   {key} repeat identical: {same}", flush=True)
        res[key] = C.clone()
    same = torch.equal(res["g4"], res["g4t"])
    nbad = (res["g4"] != res["g4t"]).sum().item() if not same else 0
    print(f"{name}: M={M} N={N} K={K} epi={epi}  g4t == g4 bitwise: {same}  (differing elements {nbad}, nan in g4t {torch.isnan(res['g4t'].float()).sum().item()})", flush=True)
    if not same:
        d = (res["g4"].float() - res["g4t"].float())
        bad = d.nonzero()
        print("   first differing (row, col):", bad[:6].tolist(), " max abs", d.abs().max().item(), flush=True)
        rows = torch.unique(bad[:, 0])
        cols = torch.unique(bad[:, 1])
        print(f"   rows touched {rows.numel()} (min {rows.min().item()} max {rows.max().item()}), cols touched {cols.numel()} (min {cols.min().item()} max {cols.max().item()})", flush=True)
    for on in (0, 1):
        D.s2v_set_gemm_g4t(on)
        ms = timeit(f)
        print(f"   {'g4t' if on else 'g4 '}: {ms:8.3f} ms  {2*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)
    t = timeit(lambda: torch.matmul(A, W.T), iters=5)
    print(f"   (hipBLASLt via torch.matmul, no epilogue: {t:8.3f} ms  {2*M*N*K/t/1e9:8.1f} TFLOP/s)", flush=True)
    del A, W, C, res
D.s2v_set_gemm_g4t(1)
