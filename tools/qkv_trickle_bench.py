"""Same-box A/B of the fused QKV projection (bias + q/k LayerNorm + rotary, EPI_BIAS_QKNORM) at the C3 shape: gemm_g4 with the C++ epilogue
against gemm_g4t with the epilogue trickled through the next tile's K loop, and the plain-bias projection on gemm_g4t for reference.
    python tools/qkv_trickle_bench.py            (diagnostics build)"""
import ctypes, importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
L = s2v._lib
Dg = L.diag_lib()
Dg.s2v_set_gemm_g4t.argtypes = [ctypes.c_int]
DEV = "cuda:0"
M, D, K, tok, text = 38144, 3072, 3072, 19126, 226
g = torch.Generator().manual_seed(1)
A = (torch.randn(M, K, generator=g) * 0.5).bfloat16().to(DEV)
W = (torch.randn(3 * D, K, generator=g) * 0.05).bfloat16().to(DEV)
b = (torch.randn(3 * D, generator=g) * 0.2).bfloat16().to(DEV)
ln = [(1.0 + 0.3 * torch.randn(64, generator=g)).bfloat16().to(DEV), (0.2 * torch.randn(64, generator=g)).bfloat16().to(DEV),
      (1.0 + 0.3 * torch.randn(64, generator=g)).bfloat16().to(DEV), (0.2 * torch.randn(64, generator=g)).bfloat16().to(DEV)]
ang = torch.rand(tok - text, 32, generator=g) * 6.28
cs = torch.cat([ang.cos(), ang.sin()], dim=1).float().contiguous().to(DEV)
C = torch.empty(M, 3 * D, device=DEV, dtype=torch.bfloat16)


def fused():
    rc = Dg.s2v_diag_qkv_qknorm(L.ptr(A), L.ptr(W), L.ptr(b), L.ptr(ln[0]), L.ptr(ln[1]), L.ptr(ln[2]), L.ptr(ln[3]), L.ptr(cs), L.ptr(C), M, D, K, tok, text,
                                ctypes.c_float(1e-6), L.stream_ptr())
    assert rc == 0, Dg.s2v_last_error()


def plain():
    Dg.s2v_set_gemm_impl(9)
    rc = Dg.s2v_op_linear(L.ptr(A), L.ptr(W), L.ptr(b), L.ptr(C), M, 3 * D, K, 0, 1, 0, L.stream_ptr())
    assert rc == 0, Dg.s2v_last_error()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / n


flops = 2.0 * M * 3 * D * K
outs = {}
for rnd in range(2):
    for name, on, fn in (("g4 + C++ q/k-norm epilogue", 0, fused), ("g4t trickled q/k-norm", 1, fused), ("g4t plain bias", 1, plain), ("g4 plain bias", 0, plain)):
        Dg.s2v_set_gemm_g4t(on)
        ms = timed(fn)
        if fn is fused:
            outs[on] = C.clone()
        print(f"round {rnd}: {name:30s} {ms:7.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s", flush=True)
Dg.s2v_set_gemm_g4t(1)
print("bit-identical:", torch.equal(outs[0], outs[1]), " differing elements:", (outs[0] != outs[1]).sum().item())
