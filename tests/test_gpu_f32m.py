"""The fp32 model dtype on the fp32 matrix pipe (csrc/gemm_f32m.hip, csrc/attention_f32m.hip: v_mfma_f32_32x32x2_f32).

fp32 is the CPU-reference-parity mode (north_star: <= 1e-3 max-abs on the latents).  Until round 5 it ran on VALU kernels
(gemm_simple_k, attn_simple_k) -- exact but minutes per step at the headline geometry.  The MFMA kernels have to be the SAME
arithmetic, only faster:
  * GEMM: bit-identical to the VALU kernel (the instruction is a k-ordered fmaf chain), every epilogue, ragged shapes, the
    implicit-GEMM convolution addressing (through the fp32 VAE);
  * attention: <= 2e-5 against fp64 SDPA like the VALU kernel, ragged tails, spiked keys;
  * engine: fp32 engine on the matrix pipe == fp32 engine with force_simple (VALU) to 1e-5 on a whole forward, and (elsewhere)
    every fp32 golden / oracle test of the suite now runs through these kernels.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _linear(s2v, A, W, b, epi, impl):
    L = s2v._lib
    M, K = A.shape
    N = W.shape[0]
    C = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
    L.check(L.lib().s2v_op_linear(L.ptr(A), L.ptr(W), L.ptr(b) if b is not None else None, L.ptr(C), M, N, K, epi, L.DTYPE_F32, impl, L.stream_ptr()))
    torch.cuda.synchronize()
    return C


@pytest.mark.parametrize("M,N,K,epi", [(128, 128, 16, 0), (77, 52, 40, 0), (300, 200, 100, 1), (1, 1, 4, 0), (513, 129, 3072, 0),
                                        (2500, 1920, 64, 1), (1000, 64, 1920, 0), (129, 7680, 256, 1)])
def test_gemm_f32m_bit_identical_to_valu(s2v, M, N, K, epi):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.3).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    got = _linear(s2v, A, W, b, epi, 3)
    ref = _linear(s2v, A, W, b, epi, 1)
    assert torch.isfinite(got).all()
    assert torch.equal(got, ref), (got - ref).abs().max().item()
    for impl in (30, 31, 32, 33):  # round 6: every tile shape the launcher may pick (128 x 128, 128 x 64, 64 x 128, 64 x 64) returns the same bits
        assert torch.equal(_linear(s2v, A, W, b, epi, impl), ref), impl
    exact = A.double() @ W.double().T + b.double()
    if epi == 1:
        exact = torch.nn.functional.gelu(exact, approximate="tanh")
    assert (got.double() - exact).abs().max().item() <= 2e-6 * max(1.0, exact.abs().max().item()) * max(1.0, (K / 64) ** 0.5)


def test_gemm_f32m_no_bias_and_asymmetric_identity(s2v):
    """A = I against an asymmetric W: a transposed or permuted fragment map cannot pass"""
    n = 160
    A = torch.eye(n, device=DEV)
    W = (torch.arange(n * n, dtype=torch.float32, device=DEV).reshape(n, n) * 0.37).sin()
    got = _linear(s2v, A, W, None, 0, 3)
    assert torch.equal(got, W.T.contiguous())


@pytest.mark.parametrize("B,H,N", [(1, 2, 129), (2, 2, 300), (1, 1, 32), (1, 3, 1000), (2, 1, 31)])
def test_attention_f32m_vs_fp64_and_valu(s2v, B, H, N):
    g = torch.Generator().manual_seed(N)
    D = H * 64
    qkv = torch.randn(B * N, 3 * D, generator=g)
    qkv[5, D:D + 64] *= 6.0  # a spiked key row: forces large online-softmax rescales
    q, k, v = (qkv[:, i * D:(i + 1) * D].reshape(B, N, H, 64).transpose(1, 2) for i in range(3))
    ref = torch.nn.functional.scaled_dot_product_attention(q.double(), k.double(), v.double()).transpose(1, 2).reshape(B * N, D)
    qd = torch.cat([qkv, torch.zeros(64, 3 * D)]).to(DEV)
    L = s2v._lib
    outs = {}
    for impl in (5, 1):
        out = torch.full((B * N, D), float("nan"), dtype=torch.float32, device=DEV)
        L.check(L.lib().s2v_op_attention(L.ptr(qd), None, L.ptr(out), B, H, N, L.DTYPE_F32, impl, L.stream_ptr()))
        torch.cuda.synchronize()
        outs[impl] = out.cpu().double()
        assert torch.isfinite(outs[impl]).all()
    scale = max(1.0, ref.abs().max().item())
    assert (outs[5] - ref).abs().max().item() <= 2e-5 * scale
    assert (outs[5] - outs[1]).abs().max().item() <= 1e-5 * scale


@pytest.mark.parametrize("variant", ["rope", "sincos"])
def test_engine_f32_matrix_pipe_vs_valu(s2v, variant):
    """one forward of a 4-layer, 6-head model at a ragged token count: matrix-pipe engine against the VALU engine"""
    cfg = s2v.tiny(use_rope=variant == "rope", heads=6, layers=4, text_dim=128, temb=128)
    sd = s2v.weights.synthetic_state_dict(cfg, seed=5, parity=True)
    g = torch.Generator().manual_seed(6)
    F, H, W, T = 3, 10, 14, 7
    lat = torch.randn(2, F, 16, H, W, generator=g)
    text = torch.randn(2, T, 128, generator=g)
    ref = torch.randn(1, 1, 16, H, W, generator=g) * 0.7
    outs = []
    for simple in (False, True):
        m = s2v.HipCogVideoXTransformer3DModel(cfg, torch.float32, DEV, simple)
        m.load_state_dict(sd)
        eng = m.engine
        eng.set_geometry(2, T, F, H, W)
        eng.prepare_tables(H * 8, W * 8)
        eng.set_conditioning(text, ref)
        outs.append(eng.forward(lat, torch.tensor([500.0, 500.0])).cpu())
        torch.cuda.synchronize()
    assert torch.isfinite(outs[0]).all()
    err = (outs[0] - outs[1]).abs().max().item()
    assert err <= 1e-5 * max(1.0, outs[1].abs().max().item()), err
