"""Timing-only ablations of attn_bf16_k's softmax VALU work at the C3 shape (results of variants != 0 are wrong on purpose).
Usage: python tools/ablate_attn.py [variant ...]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
L = s2v._lib
DEV = "cuda:0"
sys.path.insert(0, os.path.join(ROOT, "tools"))
from microbench import timeit  # noqa: E402

B, H, N = 2, 48, 19126
D = H * 64
qkv = torch.randn(B * N + 64, 3 * D, device=DEV).bfloat16()
out = torch.empty(B * N, D, device=DEV, dtype=torch.bfloat16)
vt = torch.zeros(B * H * 64 * ((N + 63) // 64 * 64), dtype=torch.bfloat16, device=DEV)
f = lambda: L.check(L.lib().s2v_op_attention(L.ptr(qkv), L.ptr(vt), L.ptr(out), B, H, N, 1, 0, L.stream_ptr()))
fl = 4 * B * H * N * N * 64
names = {0: "default (8 waves)", 1: "no exp2", 2: "no row-sum adds", 3: "no row max", 4: "4-wave blocks"}
ref = None
for v in [int(x) for x in sys.argv[1:]] or [0, 1, 2, 3]:
    L.lib().s2v_set_attn_variant(v)
    out.zero_()
    f()
    torch.cuda.synchronize()
    if v == 0 and ref is None:
        ref = out.float().clone()
    elif ref is not None:
        d = out.float() - ref
        print(f"   variant {v} vs 0: max abs {d.abs().max().item():.3e}  rel L2 {(d.norm() / ref.norm()).item():.3e}", flush=True)
    ms = timeit(f, iters=5, warm=2)
    print(f"attn variant {v} ({names.get(v, '?')}): {ms:8.3f} ms  {fl/ms/1e9:8.1f} TFLOP/s", flush=True)
L.lib().s2v_set_attn_variant(0)
