"""Kernel micro-benchmarks at the C3 (CogVideoX-5B, 49x480x720, CFG pair) shapes; prints TFLOP/s per kernel.
Usage: python tools/microbench.py [gemm] [attn] [ew]"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
L = s2v._lib
DEV = "cuda:0"
LABELS = {0: "tile128x128", 2: "stag256x128", 5: "w8-lockstep", 7: "pp64-pingpong", 8: "q4-fourwave", 9: "g4-asm"}
IMPLS = [int(x) for x in os.environ.get("S2V_IMPLS", "7,5").split(",")]
if os.environ.get("S2V_NO_G4T") == "1":
    L.diag_lib().s2v_set_gemm_g4t(0)  # impl 9 then means gemm_g4 itself, not the persistent gemm_g4t where that qualifies


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


def bench_gemm():
    M = 38400
    for name, N, K, epi in (("qkv", 9216, 3072, 0), ("out", 3072, 3072, 0), ("ff1+gelu", 12288, 3072, 1), ("ff2", 3072, 12288, 0)):
        A = (torch.randn(M, K, device=DEV) * 0.5).bfloat16()
        W = (torch.randn(N, K, device=DEV) * 0.02).bfloat16()
        b = torch.zeros(N, device=DEV, dtype=torch.bfloat16)
        C = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        f = lambda: L.check(L.diag_lib().s2v_op_linear(L.ptr(A), L.ptr(W), L.ptr(b), L.ptr(C), M, N, K, epi, 1, 0, L.stream_ptr()))
        ref = None
        for impl in IMPLS:
            L.diag_lib().s2v_set_gemm_impl(impl)
            C.zero_()
            f()
            torch.cuda.synchronize()
            if ref is None:
                ref = C.clone()
            elif not torch.equal(ref, C):
                print(f"   !! impl {impl} differs from impl {IMPLS[0]}: max abs {(ref.float() - C.float()).abs().max().item():.4g}", flush=True)
            ms = timeit(f)
            print(f"gemm[{LABELS[impl]}] {name:9s} M={M} N={N} K={K}: {ms:8.3f} ms  {2*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)
        t = timeit(lambda: torch.matmul(A, W.T), iters=5)
        print(f"   (hipBLASLt via torch.matmul: {t:8.3f} ms  {2*M*N*K/t/1e9:8.1f} TFLOP/s)", flush=True)
        L.diag_lib().s2v_set_gemm_impl(7)
        del A, W, C


def bench_gemm_c1():
    """the four linears of a C1 block (2B, 9 x 256 x 256: M = 2500 rows, padded to 2560 here) on each tiling: which one fills 256 CUs best"""
    M = 2560
    for name, N, K, epi in (("qkv", 5760, 1920, 0), ("out", 1920, 1920, 0), ("ff1+gelu", 7680, 1920, 1), ("ff2", 1920, 7680, 0)):
        A = (torch.randn(M, K, device=DEV) * 0.5).bfloat16()
        W = (torch.randn(N, K, device=DEV) * 0.02).bfloat16()
        b = torch.zeros(N, device=DEV, dtype=torch.bfloat16)
        C = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        f = lambda: L.check(L.diag_lib().s2v_op_linear(L.ptr(A), L.ptr(W), L.ptr(b), L.ptr(C), M, N, K, epi, 1, 0, L.stream_ptr()))
        for impl in (0, 2, 7, 9):
            if impl in (7, 9) and N % 256:
                continue
            L.diag_lib().s2v_set_gemm_impl(impl)
            f()
            torch.cuda.synchronize()
            ms = timeit(f, iters=50, warm=10)
            print(f"gemm_c1[{LABELS[impl]}] {name:9s} M={M} N={N} K={K}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)
        L.diag_lib().s2v_set_gemm_impl(9)


def bench_attn():
    for (B, H, N) in ((2, 48, 19126), (2, 30, 19126), (2, 30, 1250)):
        D = H * 64
        qkv = torch.randn(B * N + 64, 3 * D, device=DEV).bfloat16()
        out = torch.empty(B * N, D, device=DEV, dtype=torch.bfloat16)
        vt = torch.zeros(B * H * 64 * ((N + 63) // 64 * 64), dtype=torch.bfloat16, device=DEV)
        f = lambda: L.check(L.lib().s2v_op_attention(L.ptr(qkv), L.ptr(vt), L.ptr(out), B, H, N, 1, 0, L.stream_ptr()))
        ms = timeit(f, iters=5, warm=2)
        fl = 4 * B * H * N * N * 64
        print(f"attn B={B} H={H} N={N}: {ms:8.3f} ms  {fl/ms/1e9:8.1f} TFLOP/s (incl. V^T transpose)", flush=True)
        q, k, v = (qkv[: B * N, i * D : (i + 1) * D].reshape(B, N, H, 64).transpose(1, 2) for i in range(3))
        t = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v), iters=3, warm=1)
        print(f"   (torch SDPA: {t:8.3f} ms  {fl/t/1e9:8.1f} TFLOP/s)", flush=True)
        del qkv, out, vt


if __name__ == "__main__":
    what = sys.argv[1:] or ["gemm", "attn"]
    print(torch.cuda.get_device_name(0), flush=True)
    if "gemm_c1" in what:
        bench_gemm_c1()
    if "gemm" in what:
        bench_gemm()
    if "attn" in what:
        bench_attn()


def bench_vae():
    import time
    cfg = s2v.VAEConfig(scaling_factor=0.7)
    sd = s2v.weights.synthetic_vae_state_dict(cfg, seed=1, device=DEV)
    vae = s2v.HipAutoencoderKLCogVideoX(cfg, torch.bfloat16, DEV)
    vae.load_state_dict(sd)
    del sd
    lat = torch.randn(1, 13, 16, 60, 90, device=DEV).bfloat16()
    for tiling in (False, True):
        vae.use_tiling = tiling
        y = vae.decode_latents(lat)
        torch.cuda.synchronize()
        t0 = time.time()
        y = vae.decode_latents(lat)
        t_enq = time.time() - t0  # host time to enqueue the decode (launch-bound if close to the total)
        torch.cuda.synchronize()
        dt = time.time() - t0
        fl = 441.04e12 if tiling else 315.03e12
        print(f"vae decode 13x60x90 -> {tuple(y.shape)} tiling={tiling}: {dt*1e3:8.1f} ms  {fl/dt/1e12:7.1f} TFLOP/s  "
              f"enqueue={t_enq*1e3:.0f} ms finite={bool(torch.isfinite(y.float()).all())} mem={torch.cuda.mem_get_info()[0]/2**30:.0f} GiB free", flush=True)


if "vae" in sys.argv[1:]:
    bench_vae()

if "t5" in sys.argv[1:]:
    # T5-v1.1-XXL encode of the CFG pair's prompts (2 x 226 tokens): per-kernel times come from `rocprofv3 --kernel-trace --stats`
    tcfg = s2v.T5Config()
    t5 = s2v.HipT5EncoderModel(tcfg, torch.bfloat16, DEV)
    t5.load_state_dict(s2v.weights.synthetic_t5_state_dict(tcfg, seed=10, device=DEV, dtype=torch.bfloat16, gain=0.5))
    ids = torch.randint(1, tcfg.vocab_size, (2, 226), device=DEV)
    t5(ids)
    torch.cuda.synchronize()
    for _ in range(3):
        t0 = time.perf_counter()
        emb = t5(ids)[0]
        torch.cuda.synchronize()
        print(f"t5 xxl encode 2 x 226 tokens: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
