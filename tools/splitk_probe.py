"""gemm_g4 with and without split K on the C1 FF2 shape (80 tiles, K = 7680) and the out-projection shape (K = 1920, which
choose_splitk leaves alone).  Run under `rocprofv3 --kernel-trace`: the two forms differ in grid size (tiles x 256 vs tiles x S x 256
work-items); tools/splitk_probe.py --summ <trace.csv> prints the mean duration per (kernel, grid)."""
import csv
import importlib
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--summ":
    acc = defaultdict(list)
    for r in csv.DictReader(open(sys.argv[2])):
        if "gemm" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:40], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"))].append(
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(acc.items()):
        v = v[len(v) // 4:]
        print(f"{k[0]:42s} grid {k[1]:>8s}: {len(v):3d} launches, mean {sum(v) / len(v):8.1f} us, min {min(v):8.1f} us")
    sys.exit(0)

import torch

s2v = importlib.import_module("disentangled-subject-to-vid_amd")
L = s2v._lib
DEV = "cuda:0"
for M, N, K in ((2560, 2048, 7680), (2560, 2048, 1920), (2560, 2048, 3840)):
    A = (torch.randn(M, K, device=DEV) * 0.5).bfloat16()
    W = (torch.randn(N, K, device=DEV) * 0.02).bfloat16()
    b = torch.zeros(N, device=DEV, dtype=torch.bfloat16)
    C = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    for impl in (0, 2):
        for _ in range(12):
            rc = L.lib().s2v_op_linear(L.ptr(A), L.ptr(W), L.ptr(b), L.ptr(C), M, N, K, 0, 1, impl, L.stream_ptr())
            if rc != 0:
                print(M, N, K, "impl", impl, "refused:", L.lib().s2v_last_error().decode())
                break
        torch.cuda.synchronize()
