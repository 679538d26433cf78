"""gemm_q4 (diagnostics build, impl 8) against an fp32 reference: error map per 32-row x 64-column unit of the first output tiles."""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
L = s2v._lib
DEV = "cuda:0"
M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (2048, 512, 3072)))
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
torch.manual_seed(0)
A = (torch.randn(M, K, device=DEV) * 0.5).bfloat16()
W = (torch.randn(N, K, device=DEV) * 0.02).bfloat16()
b = torch.randn(N, device=DEV).bfloat16()
C = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
L.diag_lib().s2v_set_gemm_impl(8)
L.check(L.diag_lib().s2v_op_linear(L.ptr(A), L.ptr(W), L.ptr(b), L.ptr(C), M, N, K, epi, 1, 0, L.stream_ptr()))
torch.cuda.synchronize()
ref = A.float() @ W.float().T + b.float()
if epi == 1:
    ref = torch.nn.functional.gelu(ref.bfloat16().float(), approximate="tanh")
err = (C.float() - ref).abs()
print(f"M={M} N={N} K={K} epi={epi}: max err {err.max().item():.4g} (ref max {ref.abs().max().item():.3g}); bad elements {(err > 0.05).sum().item()} of {err.numel()}")
um = err[: min(M, 512), : min(N, 512)].reshape(-1, 32, min(N, 512) // 64, 64).amax(dim=(1, 3))
for r in range(um.shape[0]):
    print(" ".join("X" if v > 0.05 else "." for v in um[r].tolist()))
tiles = err.reshape(M // 256, 256, N // 256, 256).amax(dim=(1, 3))
print("bad tiles:", (tiles > 0.05).sum().item(), "of", tiles.numel())
L.diag_lib().s2v_set_gemm_impl(7)
