"""attn_q4 (variants 6 / 7: per item / persistent) against the round-2 ping-pong kernel attn_pp (10 / 11) over the sequence length (diagnostics
library): where launch_attn_bf16's ATTN_PP_MAX_TOKENS belongs."""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
L = s2v._lib
DEV = "cuda:0"
def timeit(fn, iters=50, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (B, H, N) in ((2, 30, 1250), (2, 48, 1250), (2, 30, 4000), (2, 48, 6000), (2, 48, 8192), (2, 48, 12000), (2, 30, 19126)):
    D = H * 64
    qkv = torch.randn(B * N + 64, 3 * D, device=DEV).bfloat16()
    out = torch.empty(B * N, D, device=DEV, dtype=torch.bfloat16)
    vt = torch.zeros(B * H * 64 * ((N + 63) // 64 * 64), dtype=torch.bfloat16, device=DEV)
    f = lambda: L.check(L.diag_lib().s2v_op_attention(L.ptr(qkv), L.ptr(vt), L.ptr(out), B, H, N, 1, 0, L.stream_ptr()))
    ref = None
    for v in (6, 10, 7, 11):
        L.diag_lib().s2v_set_attn_variant(v)
        out.zero_(); f(); torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        same = torch.equal(ref, out)
        print(f"B={B} H={H} N={N} variant {v:2d}: {timeit(f)*1e3:7.1f} us  same_as_6={same}", flush=True)
    L.diag_lib().s2v_set_attn_variant(0)
