import sys, torch, importlib
sys.path.insert(0, ".")
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
sys.path.insert(0, "tests")
import importlib.util
spec = importlib.util.spec_from_file_location("t", "tests/test_gpu_fp8.py"); T = importlib.util.module_from_spec(spec); spec.loader.exec_module(T)
B, H, N = 1, 1, 64
D = 64
def run(qkv):
    got, *_ = T._attention_fp8qk(s2v, qkv.bfloat16(), B, H, N)
    return got
# 1: uniform
qkv = torch.zeros(N, 3 * D); qkv[:, 2 * D:] = torch.arange(N).float()[:, None] / 8
print("uniform: out[0,:4]", run(qkv)[0, :4].tolist(), "expect", (torch.arange(N).float() / 8).mean().item())
# 2: one-hot matching: q_i = 40 e_{i}, k_n = e_n  (64 dims, 64 keys) -> row i selects key i: out[i] = i / 8
qkv = torch.zeros(N, 3 * D)
for i in range(N):
    qkv[i, i] = 64.0
    qkv[i, D + i] = 8.0
qkv[:, 2 * D:] = torch.arange(N).float()[:, None] / 8
o = run(qkv)
print("one-hot: selected key per query row (expect 0..63):", (o[:, 0] * 8).round().int().tolist())
# 3: scales: k rows with different magnitudes
qkv = torch.zeros(N, 3 * D)
for i in range(N):
    qkv[i, i] = 64.0
    qkv[i, D + i] = 8.0 * (2.0 ** ((i % 5) - 2))
    qkv[i, D + (i + 32) % 64] = 0.01
qkv[:, 2 * D:] = torch.arange(N).float()[:, None] / 8
o = run(qkv)
print("scaled one-hot: selected key per query row (expect 0..63):", (o[:, 0] * 8).round().int().tolist())
