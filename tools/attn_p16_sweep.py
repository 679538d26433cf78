"""What attn_p_format 1 (fp16 P, deferred maximum 2^14) costs on spiky score distributions: the attention launch at the C3 shape with q scaled so
that the scores have a spread of 1 ... 12 natural units, bf16 P (impl 0, threshold 2^64) against fp16 P (impl 3), interleaved, op level."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
L = s2v._lib
DEV = "cuda:0"
B, H, N = 2, 48, 19126
D = H * 64
g = torch.Generator(device=DEV).manual_seed(1)
base = torch.randn(B * N + 64, 3 * D, generator=g, device=DEV, dtype=torch.float32)
vt = torch.zeros(B * H * 64 * ((N + 63) // 64 * 64), dtype=torch.bfloat16, device=DEV)
out = torch.empty(B * N, D, dtype=torch.bfloat16, device=DEV)
st = L.stream_ptr()
for spread in (1.0, 2.0, 3.0, 4.0, 6.0, 8.0, 12.0):
    x = base.clone()
    x[:, :D] *= spread
    qkv = x.bfloat16()
    res = {}
    for rnd in range(2):
        for impl in (0, 3):
            L.check(L.lib().s2v_op_attention(L.ptr(qkv), L.ptr(vt), L.ptr(out), B, H, N, L.DTYPE_BF16, impl, st))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                L.check(L.lib().s2v_op_attention(L.ptr(qkv), L.ptr(vt), L.ptr(out), B, H, N, L.DTYPE_BF16, impl, st))
            torch.cuda.synchronize()
            res.setdefault(impl, []).append((time.perf_counter() - t0) / 3 * 1e3)
    a, b = min(res[0]), min(res[3])
    print(f"score spread {spread:4.1f}: bf16 P {a:7.3f} ms   fp16 P {b:7.3f} ms   ({(b / a - 1) * 100:+5.1f} %)", flush=True)
