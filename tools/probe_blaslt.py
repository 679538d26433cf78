"""What does the vendor GEMM (hipBLASLt through torch.matmul) run for the FF1 shape?  Reference point only."""
import torch
DEV = "cuda:0"
M, N, K = 38400, 12288, 3072
A = (torch.randn(M, K, device=DEV) * 0.5).bfloat16()
W = (torch.randn(N, K, device=DEV) * 0.02).bfloat16()
for _ in range(6):
    C = torch.matmul(A, W.T)
torch.cuda.synchronize()
