"""Per-wave stall accounting of gemm_bf16_pp64 (ABL 4, s_memtime) on the FF1 shape."""
import ctypes, importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
L = s2v._lib
DEV = "cuda:0"
M, N, K = 38400, 12288, 3072
A = (torch.randn(M, K, device=DEV) * 0.5).bfloat16()
W = (torch.randn(N, K, device=DEV) * 0.02).bfloat16()
b = torch.zeros(N, device=DEV, dtype=torch.bfloat16)
C = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
L.diag_lib().s2v_set_gemm_impl(int(sys.argv[1]) if len(sys.argv) > 1 else 7 | (4 << 8))
for _ in range(3):
    L.check(L.diag_lib().s2v_op_linear(L.ptr(A), L.ptr(W), L.ptr(b), L.ptr(C), M, N, K, 0, 1, 0, L.stream_ptr()))
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 64)()
L.diag_lib().s2v_debug_read.argtypes = [ctypes.c_void_p]
assert L.diag_lib().s2v_debug_read(buf) == 0
names = ["bar(load)", "PROLOGUE/nh", "mfma issue", "EPILOGUE/nh", "bar(comp)", "loop total", "ds_read issue", "dma issue"]
nh = K // 32
print("cycles per half-step (s_memtime ticks / %d half-steps); %s" % (nh, ", ".join(names)))
for w in range(8):
    print(f"wave {w}: " + "  ".join(f"{buf[w*8+e]/nh:8.1f}" for e in range(8)))
L.diag_lib().s2v_set_gemm_impl(7)
