"""Rates of the fp32 matrix-pipe kernels (gemm_f32m, attn_f32m) at the C3 shapes, beside the VALU kernels they replace.
    python tools/f32m_bench.py [valu]      (valu: also time gemm_simple_k / attn_simple_k on reduced shapes)"""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
L = s2v._lib
DEV = "cuda:0"


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


if "c1" in sys.argv:  # configs[0] (2B, 9 x 256 x 256: M = 2500 rows, D = 1920): every tile shape the launcher can pick, and its pick (impl 3)
    M = 2500
    for name, N, K, epi in (("qkv", 5760, 1920, 0), ("out", 1920, 1920, 0), ("ff1+gelu", 7680, 1920, 1), ("ff2", 1920, 7680, 0)):
        A = torch.randn(M, K, device=DEV)
        W = torch.randn(N, K, device=DEV) * 0.02
        b = torch.randn(N, device=DEV)
        C = torch.empty(M, N, device=DEV)
        best = {}
        for rnd in range(3):  # three interleaved rounds, the minimum per tile: the first launches of a process run at a lower clock
            for impl, tag in ((30, "128x128"), (31, "128x64"), (32, "64x128"), (33, "64x64"), (3, "launcher")):
                ms = timed(lambda: L.check(L.lib().s2v_op_linear(L.ptr(A), L.ptr(W), L.ptr(b), L.ptr(C), M, N, K, epi, L.DTYPE_F32, impl, L.stream_ptr())), n=20)
                best[tag] = min(best.get(tag, 1e9), ms)
        for tag, ms in best.items():
            print(f"gemm_c1 {name:9s} {tag:9s}: M {M} N {N} K {K}: {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s", flush=True)
    for B, H, N in ((2, 30, 1250),):  # configs[0]'s attention (a two-wave workgroup form was tried in round 6 and dropped: profiles/r06_f32m_c1_tiles.txt)
        D = H * 64
        qkv = torch.randn(B * N + 64, 3 * D, device=DEV)
        out = torch.empty(B * N, D, device=DEV)
        best = {}
        for rnd in range(3):
            for impl, tag in ((5, "attn_f32m"),):
                ms = timed(lambda: L.check(L.lib().s2v_op_attention(L.ptr(qkv), None, L.ptr(out), B, H, N, L.DTYPE_F32, impl, L.stream_ptr())), n=20)
                best[tag] = min(best.get(tag, 1e9), ms)
        for tag, ms in best.items():
            print(f"attention_c1 {tag:9s}: B {B} H {H} N {N}: {ms * 1e3:8.1f} us  {4.0 * B * H * N * N * 64 / ms / 1e9:7.1f} TFLOP/s", flush=True)
    sys.exit(0)
M = 38252
for name, N, K, epi in (("qkv", 9216, 3072, 0), ("out", 3072, 3072, 0), ("ff1+gelu", 12288, 3072, 1), ("ff2", 3072, 12288, 0)):
    A = torch.randn(M, K, device=DEV)
    W = torch.randn(N, K, device=DEV) * 0.02
    b = torch.randn(N, device=DEV)
    C = torch.empty(M, N, device=DEV)
    for impl, tag in ((3, "f32m"),) + (((1, "valu"),) if "valu" in sys.argv else ()):
        Mi = M if impl == 3 else 4096
        ms = timed(lambda: L.check(L.lib().s2v_op_linear(L.ptr(A), L.ptr(W), L.ptr(b), L.ptr(C), Mi, N, K, epi, L.DTYPE_F32, impl, L.stream_ptr())))
        print(f"gemm {name:9s} {tag}: M {Mi} N {N} K {K}: {ms:8.2f} ms  {2.0 * Mi * N * K / ms / 1e9:7.1f} TFLOP/s", flush=True)
    del A, W, C
for B, H, N in ((2, 48, 19126),):
    D = H * 64
    qkv = torch.randn(B * N + 64, 3 * D, device=DEV)
    out = torch.empty(B * N, D, device=DEV)
    for impl, tag in ((5, "f32m"),) + (((1, "valu"),) if "valu" in sys.argv else ()):
        Hi = H if impl == 5 else 2
        ms = timed(lambda: L.check(L.lib().s2v_op_attention(L.ptr(qkv), None, L.ptr(out), B, Hi, N, L.DTYPE_F32, impl, L.stream_ptr())), n=2)
        print(f"attention {tag}: B {B} H {Hi} N {N}: {ms:8.2f} ms  {4.0 * B * Hi * N * N * 64 / ms / 1e9:7.1f} TFLOP/s", flush=True)
