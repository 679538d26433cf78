"""Same-box A/B of the joint attention at the C3 / C5 geometries: s2v_op_attention (bf16 attn_q4 / attn_pp) against s2v_op_attention_fp8qk
(attn_q4f: MX e4m3 q / k, QK^T on v_mfma_scale_f32_32x32x64_f8f6f4).  Both ops include their V^T (and quantisation) passes; run under
`rocprofv3 --kernel-trace --stats` for the kernels alone.  Also prints the rel-L2 of the two outputs against each other.
    python tools/attn_fp8qk_probe.py [N ...]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import importlib

s2v = importlib.import_module("disentangled-subject-to-vid_amd")
L = s2v._lib
DEV = "cuda:0"
B, H = 2, 48
D = H * 64
for N in [int(x) for x in sys.argv[1:]] or [19126, 50626]:
    g = torch.Generator(device=DEV).manual_seed(N)
    qkv = torch.randn(B * N + 64, 3 * D, generator=g, device=DEV, dtype=torch.float32).bfloat16()
    npad = (N + 63) // 64 * 64
    vt = torch.zeros(B * H * 64 * npad, dtype=torch.bfloat16, device=DEV)
    need = B * H * (66 * N + 68 * npad) + 1024
    scratch = torch.zeros(need, dtype=torch.uint8, device=DEV)
    o16 = torch.empty(B * N, D, dtype=torch.bfloat16, device=DEV)
    o8 = torch.empty_like(o16)
    st = L.stream_ptr()

    def run16():
        L.check(L.lib().s2v_op_attention(L.ptr(qkv), L.ptr(vt), L.ptr(o16), B, H, N, L.DTYPE_BF16, 0, st))

    def run8():
        L.check(L.lib().s2v_op_attention_fp8qk(L.ptr(qkv), L.ptr(vt), L.ptr(scratch), need, L.ptr(o8), B, H, N, st))

    oh = torch.empty_like(o16)

    def runh():   # attn_p_format 1: fp16 P / V^T, packed fp16 row sums (attn_q4h)
        L.check(L.lib().s2v_op_attention(L.ptr(qkv), L.ptr(vt), L.ptr(oh), B, H, N, L.DTYPE_BF16, 3, st))

    for name, fn in (("bf16  ", run16), ("f16-P ", runh), ("fp8-qk", run8)) * int(__import__("os").environ.get("ROUNDS", "2")):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        print(f"N={N} {name}: {ms:8.3f} ms per op (V^T / quantisation passes included)  {4 * B * H * N * N * 64 / ms / 1e9:7.1f} TFLOP/s", flush=True)
    relh = ((oh.float() - o16.float()).norm() / o16.float().norm()).item()
    print(f"N={N} rel-L2 f16-P vs bf16 output: {relh:.3e}  finite {bool(torch.isfinite(oh.float()).all())}", flush=True)
    rel = ((o8.float() - o16.float()).norm() / o16.float().norm()).item()
    print(f"N={N} rel-L2 fp8-qk vs bf16 output (unit-variance q, k): {rel:.3e}  finite {bool(torch.isfinite(o8.float()).all())}", flush=True)
