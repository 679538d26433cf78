import ctypes, importlib, os, sys, torch
sys.path.insert(0, "/root/repo")
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
L = s2v._lib
Dg = L.diag_lib()
Dg.s2v_set_gemm_g4t.argtypes = [ctypes.c_int]
DEV = "cuda:0"
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n
M = 38144
for name, N, K in (("out", 3072, 3072), ("ff2", 3072, 12288)):
    g = torch.Generator().manual_seed(1)
    A = (torch.randn(M, K, generator=g) * 0.5).bfloat16().to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(DEV)
    b = (torch.randn(N, generator=g) * 0.2).bfloat16().to(DEV)
    C = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    def run():
        Dg.s2v_set_gemm_impl(9)
        rc = Dg.s2v_op_linear(L.ptr(A), L.ptr(W), L.ptr(b), L.ptr(C), M, N, K, 0, 1, 0, L.stream_ptr())
        assert rc == 0
    for rnd in range(2):
        for on in (0, 1):
            Dg.s2v_set_gemm_g4t(on)
            ms = timed(run)
            print(f"{name} M {M} N {N} K {K} bias epilogue, g4t={on}: {ms:.3f} ms  {2.0*M*N*K/ms/1e9:.0f} TF", flush=True)
Dg.s2v_set_gemm_g4t(1)
