"""Race screen for the MFMA GEMM schedules (pytest -m gpu): the ping-pong kernel (default) must give BIT-IDENTICAL results to
the lock-step ring kernel and to itself across repeated launches, on shapes that exercise one K-tile, odd K-tile counts,
ragged M, the padded last column tile, and both epilogues of s2v_op_linear; plus fp32-reference closeness.  The ring kernel and
the schedule knob exist only in libs2v_hip_diag.so (build.py --diag, built by __graft_entry__.build()); the product library's
result for the same call must equal both."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def run(L, A, W, b, M, N, K, epi, impl):
    """impl None: the product library; 5 / 7: the diagnostics library with that schedule selected"""
    C = torch.full((A.shape[0], N), float("nan"), device=DEV, dtype=torch.bfloat16)
    lib = L.lib() if impl is None else L.diag_lib()
    if impl is not None:
        lib.s2v_set_gemm_impl(impl)
    rc = lib.s2v_op_linear(L.ptr(A), L.ptr(W), L.ptr(b), L.ptr(C), M, N, K, epi, 1, 0, L.stream_ptr())
    assert rc == 0, lib.s2v_last_error()
    torch.cuda.synchronize()
    if impl is not None:
        lib.s2v_set_gemm_impl(7)
    return C[:M]


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (256, 512, 128), (512, 256, 192), (1024, 768, 3072), (4096, 3072, 1024),
                                   (2560, 1024, 12288)])
@pytest.mark.parametrize("epi", [0, 1])
def test_pingpong_matches_ring_bitwise_and_is_repeatable(s2v, M, N, K, epi):
    L = s2v._lib
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * 0.5).bfloat16().to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(DEV)
    b = (torch.randn(N, generator=g) * 0.1).bfloat16().to(DEV)
    ref = run(L, A, W, b, M, N, K, epi, 5)          # lock-step 8-wave ring kernel
    assert torch.isfinite(ref.float()).all()
    for rep in range(6):
        out = run(L, A, W, b, M, N, K, epi, 7 if rep % 2 else None)  # ping-pong K64 kernel: diagnostics build / product build
        assert torch.equal(out, ref), f"rep {rep}: max diff {(out.float() - ref.float()).abs().max().item()}"
    y = A.float() @ W.float().T + b.float()
    if epi == 1:
        y = torch.nn.functional.gelu(y.bfloat16().float(), approximate="tanh")
    rel = ((ref.float() - y).norm() / y.norm()).item()
    assert rel <= 1e-2, rel

