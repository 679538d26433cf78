"""Per-wave stall accounting of gemm_q4 (ACCT, s_memtime) on the FF1 shape: vmcnt wait, barrier wait, MFMA-step issue, loop."""
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
ABL = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L.diag_lib().s2v_set_gemm_impl(8 | (ABL << 8))
f = lambda: L.check(L.diag_lib().s2v_op_linear(L.ptr(A), L.ptr(W), L.ptr(b), L.ptr(C), M, N, K, 0, 1, 0, L.stream_ptr()))
for _ in range(3):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    f()
e1.record()
torch.cuda.synchronize()
print(f"ablate {ABL}: {e0.elapsed_time(e1) / 10:.3f} ms per launch")
buf = (ctypes.c_longlong * 64)()
L.diag_lib().s2v_debug_read.argtypes = [ctypes.c_void_p]
assert L.diag_lib().s2v_debug_read(buf) == 0
nt = K // 64
ntile = max(buf[7], 1)
print("per output tile (block 100, %d tiles): conversion, K loop [per K-tile] | final flush, prologue, whole WG, WG in 100 MHz ticks" % ntile)
for w in range(4):
    print(f"wave {w}: conv {buf[w*8+0]/ntile:9.0f}  loop {buf[w*8+3]/ntile:9.0f} [{buf[w*8+3]/ntile/nt:7.1f}] | {buf[w*8+1]:7d} {buf[w*8+4]:6d} {buf[w*8+5]:9d} {buf[w*8+6]:8d}")
L.diag_lib().s2v_set_gemm_impl(7)

blk = (ctypes.c_longlong * 2048)()
L.diag_lib().s2v_debug_read_blocks.argtypes = [ctypes.c_void_p]
if ABL in (4, 5) and L.diag_lib().s2v_debug_read_blocks(blk) == 0:
    import numpy as np
    bb = np.array(list(blk), dtype=np.int64).reshape(2, 256, 4)
    order = [0, 1] if bb[0, :, 0].min() < bb[1, :, 0].min() else [1, 0]
    b, b2 = bb[order[0]], bb[order[1]]  # the last two launches, older first
    t0 = b[:, 0].min()
    st, en = (b[:, 0] - t0) / 100.0, (b[:, 1] - t0) / 100.0  # microseconds
    print(f"workgroup start: min {st.min():.1f} max {st.max():.1f} us; end: min {en.min():.1f} max {en.max():.1f} us; "
          f"duration mean {np.mean(en - st):.1f} us; tiles {b[:, 3].min()}..{b[:, 3].max()}; clock {np.mean(b[:, 2] / (en - st)) / 1e3:.3f} GHz")
    print(f"next launch: first workgroup starts {(b2[:, 0].min() - t0) / 100.0:.1f} us, i.e. {(b2[:, 0].min() - b[:, 1].max()) / 100.0:.1f} us after the last one of this launch ended")
    for x in range(8):
        sel = b[x::8]
        print(f"  xcd {x}: start {((sel[:, 0] - t0) / 100.0).mean():7.1f}  end {((sel[:, 1] - t0) / 100.0).mean():8.1f}  us/tile {np.mean((sel[:, 1] - sel[:, 0]) / 100.0 / sel[:, 3]):.2f}  clock {np.mean(sel[:, 2] / ((sel[:, 1] - sel[:, 0]) / 100.0)) / 1e3:.3f} GHz")
