"""W8A8 fp8 path (BASELINE configs[4]: "CogVideoX-5B fp8 (CDNA4 fp8 MFMA) weights").  The reference has no fp8 implementation
(diffusers/src/diffusers/quantizers is bitsandbytes-only), so parity is UNPINNED by construction; the contract is stated
against this build's own arithmetic:
  * the fp8 GEMM must equal, to fp32 accumulation order, a torch emulation of the same quantisation (per-row amax / 448
    scales, round-to-nearest-even e4m3, fp32 products) -- this pins the operand layout of v_mfma_scale_f32_32x32x64_f8f6f4,
    the scale handling and the epilogues;
  * against the un-quantised bf16 GEMM the relative L2 error stays below 4e-2 (e4m3 has 3 mantissa bits).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
FP8_ENGINE_BAR = 1.4e-2   # 2 x measured (round 6, gpurun r06a): fp8 / fp8-qk engines against the bf16 engine 6.3e-3 ... 6.9e-3 rel-L2 over the five sites (until round 5: 5e-2); parity unpinned
DEV = "cuda:0"


def quant_rows(x):
    """per-row dynamic e4m3 quantisation as quant_rows_fp8_k does it"""
    amax = x.float().abs().amax(dim=1, keepdim=True)
    scale = torch.where(amax > 0, amax * (1.0 / 448.0), torch.ones_like(amax))
    q = (x.float() * (1.0 / scale)).to(torch.float8_e4m3fn)
    return q, scale


@pytest.mark.parametrize("M,N,K,epi", [(256, 256, 128, 0), (512, 768, 3072, 0), (1024, 512, 1024, 1), (256, 1024, 12288, 0)])
def test_op_linear_fp8_matches_emulated_quantisation(s2v, M, N, K, epi):
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * torch.rand(M, 1, generator=g) * 2).bfloat16()   # rows of different magnitude
    A[3, 17] = 30.0                                                                    # an outlier sets that row's scale
    W = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    b = (torch.randn(N, generator=g) * 0.1).bfloat16()
    Ad, Wd, bd = A.to(DEV), W.to(DEV), b.to(DEV)
    C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    need = M * K + N * K + 4 * (M + N)
    scratch = torch.empty(need, dtype=torch.uint8, device=DEV)
    L = s2v._lib
    L.check(L.lib().s2v_op_linear_fp8(L.ptr(Ad), L.ptr(Wd), L.ptr(bd), L.ptr(C), M, N, K, epi, L.ptr(scratch), need, L.stream_ptr()))
    torch.cuda.synchronize()
    got = C.float().cpu()
    assert torch.isfinite(got).all()
    qa, sa = quant_rows(Ad)
    qw, sw = quant_rows(Wd)
    emu = (qa.float() @ qw.float().T) * sa * sw.T + bd.float()
    full = Ad.float() @ Wd.float().T + bd.float()
    if epi == 1:
        emu = torch.nn.functional.gelu(emu.bfloat16().float(), approximate="tanh")
        full = torch.nn.functional.gelu(full.bfloat16().float(), approximate="tanh")
    emu, full = emu.cpu(), full.cpu()
    rel_emu = ((got - emu).norm() / emu.norm()).item()
    rel_full = ((got - full).norm() / full.norm()).item()
    assert rel_emu <= 4e-3, rel_emu      # bf16 output rounding + accumulation order only
    assert rel_full <= 4e-2, rel_full    # the e4m3 quantisation error itself
    # the scratch holds what the emulation computed: quantised bytes and scales, bit for bit
    aq = scratch[: M * K].view(torch.float8_e4m3fn).view(M, K)
    assert torch.equal(aq.view(torch.uint8), qa.view(torch.uint8))
    as_ = scratch[M * K + N * K: M * K + N * K + 4 * M].view(torch.float32)
    assert torch.allclose(as_, sa.flatten(), rtol=1e-6, atol=0)


def _run_engine(s2v, cfg, sd, lat, text, ref, t, graph=False):
    m = s2v.HipCogVideoXTransformer3DModel(cfg, torch.bfloat16, DEV)
    m.load_state_dict(sd)
    eng = m.engine
    B, F, C, H, W = 2, lat.shape[1], lat.shape[2], lat.shape[3], lat.shape[4]
    eng.set_geometry(2, text.shape[1], F, H, W)
    eng.prepare_tables(H * 8, W * 8)
    eng.set_conditioning(text, ref)
    y = eng.forward(lat, torch.tensor([t, t]), shared_latent=True)
    torch.cuda.synchronize()
    return m, y


def test_fp8_engine_medium_model_vs_bf16_engine(s2v):
    """weight_format = "fp8" against the same model in bf16: per-step noise_pred relative L2 <= 5e-2 (the stated tolerance of the
    fp8 configuration; the reference has nothing to compare with), LoRA merged before the quantisation"""
    import copy

    cfg = s2v.tiny(use_rope=True, heads=4, layers=2, text_dim=128, temb=64)  # D = 256: whole fp8 tiles
    cfg.max_text_seq_length = 7
    sd = s2v.weights.synthetic_state_dict(cfg, seed=5, parity=True)
    g = torch.Generator().manual_seed(17)
    lat = torch.randn(1, 3, 16, 16, 24, generator=g).bfloat16()
    text = torch.randn(2, 7, 128, generator=g).bfloat16()
    ref = (torch.randn(1, 1, 16, 16, 24, generator=g) * 0.7).bfloat16()
    _, y16 = _run_engine(s2v, cfg, sd, lat, text, ref, 500.0)
    cfg8 = copy.copy(cfg)
    cfg8.weight_format = "fp8"
    m8, y8 = _run_engine(s2v, cfg8, sd, lat, text, ref, 500.0)
    assert torch.isfinite(y8.float()).all()
    rel = ((y8.float() - y16.float()).norm() / y16.float().norm()).item()
    print(f"MEASURED fp8 engine vs bf16 (site A): rel-l2 {rel:.3e}")
    assert 0 < rel <= FP8_ENGINE_BAR, rel  # > 0: the fp8 path really ran
    # a replica that receives the arena (fp8 copies and scales included) reproduces the result bit for bit
    m2 = s2v.HipCogVideoXTransformer3DModel(cfg8, torch.bfloat16, DEV)
    m2.engine.weight_arena().copy_(m8.engine.weight_arena())
    m2.engine.mark_weights_loaded()
    eng = m2.engine
    eng.set_geometry(2, 7, 3, 16, 24)
    eng.prepare_tables(128, 192)
    eng.set_conditioning(text, ref)
    y2 = eng.forward(lat, torch.tensor([500.0, 500.0]), shared_latent=True)
    torch.cuda.synchronize()
    assert torch.equal(y2, y8)
    with pytest.raises(s2v.S2VError):  # D = 192 does not tile for the fp8 kernel: refused at creation, not silently run in bf16
        bad = s2v.tiny(use_rope=True, heads=3, layers=1)
        bad.weight_format = "fp8"
        s2v.S2VEngine(bad, torch.bfloat16, DEV)


@pytest.mark.parametrize("lat_hw,frames", [((8, 12), 2), ((8, 8), 1), ((16, 24), 2)])
def test_fp8_engine_short_sequences_vs_bf16_engine(s2v, lat_hw, frames):
    """the fp8 engine routes attention to attn_q4 at ANY length (its epilogue writes the MX e4m3 hand-over of the out-projection), so
    below 320 tokens (five 64-key tiles) every KV tile of attn_q4 goes through its rare-path handler -- 79 and 39 tokens here, and 295
    (the last length below it); ADVICE r3.  Same tolerance as the medium model: rel-L2 <= 5e-2 against the bf16 engine (parity unpinned)."""
    import copy

    cfg = s2v.tiny(use_rope=True, heads=4, layers=2, text_dim=128, temb=64)
    cfg.max_text_seq_length = 7
    sd = s2v.weights.synthetic_state_dict(cfg, seed=5, parity=True)
    g = torch.Generator().manual_seed(19)
    H, W = lat_hw
    lat = torch.randn(1, frames, 16, H, W, generator=g).bfloat16()
    text = torch.randn(2, 7, 128, generator=g).bfloat16()
    ref = (torch.randn(1, 1, 16, H, W, generator=g) * 0.7).bfloat16()
    ntok = 7 + (frames + 1) * (H // 2) * (W // 2)
    assert ntok < 320
    _, y16 = _run_engine(s2v, cfg, sd, lat, text, ref, 500.0)
    cfg8 = copy.copy(cfg)
    cfg8.weight_format = "fp8"
    _, y8 = _run_engine(s2v, cfg8, sd, lat, text, ref, 500.0)
    assert torch.isfinite(y8.float()).all()
    rel = ((y8.float() - y16.float()).norm() / y16.float().norm()).item()
    print(f"MEASURED fp8 engine vs bf16 (site B, ntok {ntok}): rel-l2 {rel:.3e}")
    assert 0 < rel <= FP8_ENGINE_BAR, (ntok, rel)
    u, c = y8.chunk(2)
    assert not torch.equal(u, c)  # the CFG pair saw different text


def test_fp8_engine_full_tokens_properties_and_closeness(s2v):
    """5B width, 2 layers, N = 19126 tokens: finite, CFG-symmetric, deterministic, graph replay == eager, and within 5e-2
    relative L2 of the bf16 engine on the same weights"""
    import copy

    cfg = s2v.cogvideox_5b()
    cfg.num_layers = 2
    sd = s2v.weights.synthetic_state_dict(cfg, seed=3, device=DEV, parity=True)
    g = torch.Generator(device=DEV).manual_seed(4)
    F, H, W, T = 13, 60, 90, 226
    t1 = torch.randn(1, T, 4096, generator=g, device=DEV)
    text = torch.cat([t1, t1])
    ref = torch.randn(1, 1, 16, H, W, generator=g, device=DEV) * 0.7
    lat = torch.randn(1, F, 16, H, W, generator=g, device=DEV).bfloat16()
    _, y16 = _run_engine(s2v, cfg, sd, lat, text, ref, 500.0)
    cfg8 = copy.copy(cfg)
    cfg8.weight_format = "fp8"
    m8, y8 = _run_engine(s2v, cfg8, sd, lat, text, ref, 500.0)
    assert torch.isfinite(y8.float()).all()
    assert torch.equal(y8[0], y8[1])
    rel = ((y8.float() - y16.float()).norm() / y16.float().norm()).item()
    print(f"MEASURED fp8 engine vs bf16 (site C): rel-l2 {rel:.3e}")
    assert 0 < rel <= FP8_ENGINE_BAR, rel
    eng = m8.engine
    sch = s2v.CogVideoXDDIMScheduler(snr_shift_scale=1.0)
    sch.set_timesteps(50)
    a, b = lat.clone(), lat.clone()
    for x, graph in ((a, False), (b, True)):
        for i in range(2):
            t = sch.timesteps[i]
            eng.denoise_step(x, float(t), sch.coef(t, torch.bfloat16, 6.0), use_graph=graph)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


def test_fp8_quantisation_fused_into_layernorm_is_bit_identical_to_the_separate_pass(s2v):
    """under weight_format = "fp8" ln_modulate_k writes the e4m3 image + row scales of its output itself (the operand of the QKV /
    FF1 projection) instead of bf16 rows that quant_rows_fp8_k would re-read; the diagnostics build can switch that off: same bytes,
    so the same forward, bit for bit (ragged token count: the masked lanes of a row must not reach the amax)"""
    import copy

    L = s2v._lib
    diag = L.diag_lib()
    prev = L._lib
    L.lib()
    try:
        L._lib = diag
        cfg = s2v.tiny(use_rope=True, heads=4, layers=2, text_dim=128, temb=64)
        cfg.max_text_seq_length = 7
        cfg.weight_format = "fp8"
        sd = s2v.weights.synthetic_state_dict(cfg, seed=5, parity=True)
        g = torch.Generator().manual_seed(18)
        lat = torch.randn(1, 3, 16, 18, 22, generator=g).bfloat16()
        text = torch.randn(2, 7, 128, generator=g).bfloat16()
        ref = (torch.randn(1, 1, 16, 18, 22, generator=g) * 0.7).bfloat16()
        outs = []
        for fused in (1, 0):
            diag.s2v_set_fused_q8(fused)
            outs.append(_run_engine(s2v, copy.copy(cfg), sd, lat, text, ref, 300.0)[1].clone())
        assert torch.isfinite(outs[0].float()).all()
        assert torch.equal(outs[0], outs[1]), (outs[0].float() - outs[1].float()).abs().max().item()
    finally:
        diag.s2v_set_fused_q8(1)
        L._lib = prev


def quant_mx(h):
    """MX e4m3 as the FF1 epilogue produces it (gemm_epi.h): blocks of 32 columns, scale = the smallest power of two s with
    amax / s <= 448 (the E8M0 byte is the biased exponent of amax / 448 rounded up), elements = rne_e4m3(x / s)"""
    M, F = h.shape
    b = h.float().view(M, F // 32, 32)
    amax = b.abs().amax(dim=2, keepdim=True)
    t = (amax * (1.0 / 448.0)).contiguous().view(torch.int32)
    eb = ((t + 0x7FFFFF) >> 23).clamp(1, 254)
    s = ((eb << 23).view(torch.float32))
    q = (b * (((254 - eb) << 23).view(torch.float32))).to(torch.float8_e4m3fn)
    return (q.float() * s).view(M, F)


@pytest.mark.parametrize("M,D,F", [(256, 256, 1024), (512, 768, 3072)])
def test_op_ff_fp8_mx_hand_over_matches_emulation(s2v, M, D, F):
    """the fp8 FeedForward (attention.py:1237-1243) with GELU(h) handed from the FF1 epilogue to the FF2 as MX e4m3 -- block scales taken
    per lane by v_mfma_scale_f32_32x32x64_f8f6f4 -- against a torch emulation of the same quantisation (rel-L2 <= 6e-3: bf16 roundings
    of h can flip an e4m3 rounding), against the per-row-quantised hand-over (mx = 0) and against the un-quantised bf16 arithmetic"""
    L = s2v._lib
    g = torch.Generator().manual_seed(M + D + F)
    x = (torch.randn(M, D, generator=g) * (0.5 + torch.rand(M, 1, generator=g))).bfloat16().to(DEV)
    w1 = (torch.randn(F, D, generator=g) / D ** 0.5).bfloat16().to(DEV)
    b1 = (torch.randn(F, generator=g) * 0.1).bfloat16().to(DEV)
    w2 = (torch.randn(D, F, generator=g) / F ** 0.5).bfloat16().to(DEV)
    b2 = (torch.randn(D, generator=g) * 0.1).bfloat16().to(DEV)
    outs = {}
    for mx in (1, 0):
        out = torch.full((M, D), float("nan"), dtype=torch.bfloat16, device=DEV)
        L.check(L.lib().s2v_op_ff_fp8(L.ptr(x), L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(b2), L.ptr(out), M, D, F, mx, L.stream_ptr()))
        torch.cuda.synchronize()
        outs[mx] = out.float()
        assert torch.isfinite(outs[mx]).all()
    qx, sx = quant_rows(x)
    q1, s1 = quant_rows(w1)
    q2, s2 = quant_rows(w2)
    h = torch.nn.functional.gelu(((qx.float() @ q1.float().T) * sx * s1.T + b1.float()).bfloat16().float(), approximate="tanh").bfloat16().float()
    emu_mx = (quant_mx(h) @ q2.float().T) * s2.T + b2.float()
    qh, sh = quant_rows(h.bfloat16())
    emu_row = (qh.float() @ q2.float().T) * sh * s2.T + b2.float()
    full = torch.nn.functional.gelu((x.float() @ w1.float().T + b1.float()).bfloat16().float(), approximate="tanh").bfloat16().float() @ w2.float().T + b2.float()

    def rel(a, b):
        return ((a - b).norm() / b.norm()).item()

    assert rel(outs[1], emu_mx) <= 6e-3, rel(outs[1], emu_mx)
    assert rel(outs[0], emu_row) <= 6e-3, rel(outs[0], emu_row)
    assert rel(outs[1], full) <= 7e-2 and rel(outs[0], full) <= 7e-2   # two chained e4m3 GEMMs: 4e-2 each (test above) in quadrature
    assert rel(outs[1], full) <= 1.1 * rel(outs[0], full) + 1e-3, "block scales must not be worse than one scale per row"


def test_fp8_mx_hand_overs_against_the_row_quantised_engine(s2v):
    """the two MX hand-overs of the fp8 engine -- attention output -> out-projection (attention_q4.hip epilogue) and GELU(FF1) -> FF2
    (gemm_epi.h) -- against the same engine with the per-row quantisation passes (diagnostics switch s2v_set_fp8_mx): both within the
    fp8 tolerance of the bf16 engine, the block-scaled one not worse, the two close to each other; ragged token count (rows past Ntok
    must not be written, the last 256-row item is partial)"""
    import copy

    L = s2v._lib
    diag = L.diag_lib()
    prev = L._lib
    L.lib()
    try:
        L._lib = diag
        cfg = s2v.tiny(use_rope=True, heads=4, layers=2, text_dim=128, temb=64)
        cfg.max_text_seq_length = 7
        sd = s2v.weights.synthetic_state_dict(cfg, seed=5, parity=True)
        g = torch.Generator().manual_seed(19)
        lat = torch.randn(1, 3, 16, 18, 22, generator=g).bfloat16()
        text = torch.randn(2, 7, 128, generator=g).bfloat16()
        ref = (torch.randn(1, 1, 16, 18, 22, generator=g) * 0.7).bfloat16()
        y16 = _run_engine(s2v, copy.copy(cfg), sd, lat, text, ref, 300.0)[1].float()
        cfg8 = copy.copy(cfg)
        cfg8.weight_format = "fp8"
        outs = {}
        for mx in (1, 0):
            diag.s2v_set_fp8_mx(mx)
            outs[mx] = _run_engine(s2v, copy.copy(cfg8), sd, lat, text, ref, 300.0)[1].float()
            assert torch.isfinite(outs[mx]).all()

        def rel(a, b):
            return ((a - b).norm() / b.norm()).item()

        assert not torch.equal(outs[0], outs[1]), "the switch must change the arithmetic"
        assert rel(outs[1], y16) <= 5e-2 and rel(outs[0], y16) <= 5e-2
        assert rel(outs[1], y16) <= 1.15 * rel(outs[0], y16) + 1e-3
        assert rel(outs[1], outs[0]) <= 5e-2
    finally:
        diag.s2v_set_fp8_mx(1)
        L._lib = prev


# ------------------------------------------------------------------------------------------------ fp8 QK^T (weight_format 2 / "fp8-qk")
def mx_quant_blocks(x):
    """qk_quant_mx_k's quantisation of [..., 64] fp32 rows: 32-element blocks, E8M0 scale = the smallest power of two with
    amax / scale <= 448, bytes = rne_e4m3(x / scale).  Returns (e4m3 bytes as uint8 [..., 64], E8M0 exponents uint8 [..., 2])"""
    xb = x.float().reshape(*x.shape[:-1], 2, 32)
    amax = xb.abs().amax(dim=-1)
    bits = (amax * (1.0 / 448.0)).view(torch.int32).to(torch.int64)
    eb = ((bits + 0x7FFFFF) >> 23).clamp(1, 253)
    inv = ((254 - eb) << 23).to(torch.int32).view(torch.float32)
    q = (xb * inv.unsqueeze(-1)).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).reshape(*x.shape), eb.to(torch.uint8)


def mx_dequant_blocks(q_u8, eb):
    xb = q_u8.view(torch.float8_e4m3fn).float().reshape(*q_u8.shape[:-1], 2, 32)
    scale = (eb.to(torch.int32) << 23).view(torch.float32)
    return (xb * scale.unsqueeze(-1)).reshape(*q_u8.shape)


def _attention_fp8qk(s2v, qkv, B, H, N):
    L = s2v._lib
    D = H * 64
    npad = (N + 63) // 64 * 64
    qd = torch.cat([qkv, torch.zeros(64, 3 * D, dtype=torch.bfloat16)]).to(DEV)
    out = torch.full((B * N, D), float("nan"), dtype=torch.bfloat16, device=DEV)
    vt = torch.zeros(B * H * 64 * npad, dtype=torch.bfloat16, device=DEV)
    r256 = lambda v: (v + 255) // 256 * 256
    offs = [0, r256(B * H * N * 64)]
    offs.append(offs[1] + r256(B * H * N * 2))
    offs.append(offs[2] + r256(B * H * npad * 64))
    need = offs[3] + B * H * npad * 4
    scratch = torch.zeros(need, dtype=torch.uint8, device=DEV)
    L.check(L.lib().s2v_op_attention_fp8qk(L.ptr(qd), L.ptr(vt), L.ptr(scratch), need, L.ptr(out), B, H, N, L.stream_ptr()))
    torch.cuda.synchronize()
    sc = scratch.cpu()
    q8 = sc[offs[0]: offs[0] + B * H * N * 64].reshape(B, H, N, 64)
    q8s = sc[offs[1]: offs[1] + B * H * N * 2].reshape(B, H, N, 2)
    k8 = sc[offs[2]: offs[2] + B * H * npad * 64].reshape(B, H, npad, 64)
    k8s = sc[offs[3]: offs[3] + B * H * npad * 4].reshape(B, H, npad // 64, 2, 32, 4)   # [tile][block hi][r][byte kb (+ 2 unused)]
    return out.float().cpu(), q8, q8s, k8, k8s


@pytest.mark.parametrize("B,H,N", [(1, 2, 64), (1, 2, 200), (2, 3, 449), (1, 2, 1300), (1, 1, 5000)])
def test_op_attention_fp8qk_matches_emulated_quantisation(s2v, B, H, N):
    """attn_q4f (QK^T on v_mfma_scale_f32_32x32x64_f8f6f4 with MX e4m3 q / k): (1) the quantised bytes and block scales are, bit for bit, the
    torch emulation of qk_quant_mx_k; (2) the output equals fp64 softmax attention on the DEQUANTISED operands within the bf16 kernel's own
    tolerance (P is rounded to bf16 as in attn_q4) -- this pins the operand / scale layout of the scaled MFMA; (3) against attention on the
    un-quantised operands the error is that of e4m3 scores: on unit-variance q, k (score spread 1) rel-L2 <= 6e-2 (measured 4.05e-2 at
    N = 19126 and 50626, tools/attn_fp8qk_probe.py), on the magnified rows / blocks of (1) and (2) (score spread ~4: a peaked softmax) <= 1.2e-1
    (measured 6.2e-2 .. 8.3e-2).  Parity unpinned: the reference has no fp8 path."""
    g = torch.Generator().manual_seed(N + H)
    D = H * 64
    qkv = torch.randn(B * N, 3 * D, generator=g)
    qkv[:, :D] *= torch.rand(B * N, 1, generator=g) * 1.5 + 0.25          # rows and blocks of different magnitude
    qkv[:, D + 32: D + 64] *= 3.0
    qkv[5 % N, D: D + 64] *= 6.0                                            # a spiked key row
    qkv = qkv.bfloat16()
    got, q8, q8s, k8, k8s = _attention_fp8qk(s2v, qkv, B, H, N)
    assert torch.isfinite(got).all()
    c0 = 0.125 * 1.4426950408889634
    q, k, v = (qkv.float()[:, i * D:(i + 1) * D].reshape(B, N, H, 64).transpose(1, 2) for i in range(3))
    eq, es = mx_quant_blocks(q * c0)
    ek, eks = mx_quant_blocks(k)
    assert torch.equal(q8, eq) and torch.equal(q8s, es)
    assert torch.equal(k8[:, :, :N], ek) and (k8[:, :, N:] == 0).all()
    npad = k8.shape[2]
    eks_pad = torch.full((B, H, npad, 2), 127, dtype=torch.uint8)
    eks_pad[:, :, :N] = eks
    # k8s[b][h][tile][hi][r][kb] = scale of (key 64 tile + 32 kb + r, block hi)
    want = eks_pad.reshape(B, H, npad // 64, 2, 32, 2).permute(0, 1, 2, 5, 4, 3)
    assert torch.equal(k8s[..., :2], want)
    qd, kd = mx_dequant_blocks(eq, es).double(), mx_dequant_blocks(ek, eks).double()
    s = qd @ kd.transpose(-1, -2)                                           # exp2-domain scores of the dequantised operands
    p = torch.exp2(s - s.amax(dim=-1, keepdim=True))
    ref = ((p @ v.double()) / p.sum(dim=-1, keepdim=True)).transpose(1, 2).reshape(B * N, D)
    err = (got.double() - ref).abs().max().item()
    assert err <= 2e-2 * max(1.0, ref.abs().max().item()), err
    full = torch.nn.functional.scaled_dot_product_attention(q.double(), k.double(), v.double()).transpose(1, 2).reshape(B * N, D)
    rel = ((got.double() - full).norm() / full.norm()).item()
    assert rel <= 1.2e-1, rel
    plain = torch.randn(B * N, 3 * D, generator=g).bfloat16()
    got1 = _attention_fp8qk(s2v, plain, B, H, N)[0]
    q, k, v = (plain.float()[:, i * D:(i + 1) * D].reshape(B, N, H, 64).transpose(1, 2) for i in range(3))
    full1 = torch.nn.functional.scaled_dot_product_attention(q.double(), k.double(), v.double()).transpose(1, 2).reshape(B * N, D)
    rel1 = ((got1.double() - full1).norm() / full1.norm()).item()
    assert rel1 <= 6e-2, rel1


@pytest.mark.parametrize("use_rope", [True, False], ids=["rope", "sincos"])
def test_fp8_qk_norm_folded_into_the_quantisation_pass_is_bit_identical(s2v, use_rope):
    """under fp8 QK^T the attention kernel reads q and k only as MX e4m3 images, so the product folds their per-head LayerNorm + rotary embedding
    into the pass that makes the images (qk_norm_quant_mx_k) and leaves the QKV projection its plain bias epilogue.  The diagnostics build can put
    the normalisation back into the projection's epilogue (3) or into the stand-alone kernel (0): the same arithmetic at the same rounding points
    in all three places -> the same forward, bit for bit (ragged token count, text rows inside a tile, with and without rotary tables)."""
    import copy

    L = s2v._lib
    diag = L.diag_lib()
    prev = L._lib
    L.lib()
    try:
        L._lib = diag
        cfg = s2v.tiny(use_rope=use_rope, heads=4, layers=2, text_dim=128, temb=64)
        cfg.max_text_seq_length = 7
        cfg.weight_format = "fp8-qk"
        sd = s2v.weights.synthetic_state_dict(cfg, seed=5, parity=True)
        g = torch.Generator().manual_seed(19)
        lat = torch.randn(1, 3, 16, 18, 22, generator=g).bfloat16()
        text = torch.randn(2, 7, 128, generator=g).bfloat16()
        ref = (torch.randn(1, 1, 16, 18, 22, generator=g) * 0.7).bfloat16()
        outs = []
        for mode in (1, 3, 0):
            diag.s2v_set_fused_qk(mode)
            outs.append(_run_engine(s2v, copy.copy(cfg), sd, lat, text, ref, 300.0)[1].clone())
        assert torch.isfinite(outs[0].float()).all()
        assert torch.equal(outs[0], outs[1]), ("quantisation pass vs projection epilogue", (outs[0].float() - outs[1].float()).abs().max().item())
        assert torch.equal(outs[0], outs[2]), ("quantisation pass vs stand-alone kernel", (outs[0].float() - outs[2].float()).abs().max().item())
    finally:
        diag.s2v_set_fused_qk(1)
        L._lib = prev


@pytest.mark.parametrize("lat_hw,frames", [((16, 24), 3), ((8, 12), 2), ((32, 48), 2)])
def test_fp8_qk_engine_vs_fp8_and_bf16_engines(s2v, lat_hw, frames):
    """weight_format = "fp8-qk" (fp8 linears AND MX e4m3 QK^T, an option beyond configs[4]'s "fp8 weights"): against the bf16 engine within the
    stated tolerance of the fp8 configurations (rel-L2 <= 5e-2), different from the plain fp8 engine (the fp8 QK^T really ran), deterministic,
    hipGraph replay == eager; 79 tokens (every tile through the rare-path handler), 391 and 775 tokens.  Parity unpinned."""
    import copy

    cfg = s2v.tiny(use_rope=True, heads=4, layers=2, text_dim=128, temb=64)
    cfg.max_text_seq_length = 7
    sd = s2v.weights.synthetic_state_dict(cfg, seed=5, parity=True)
    g = torch.Generator().manual_seed(23)
    H, W = lat_hw
    lat = torch.randn(1, frames, 16, H, W, generator=g).bfloat16()
    text = torch.randn(2, 7, 128, generator=g).bfloat16()
    ref = (torch.randn(1, 1, 16, H, W, generator=g) * 0.7).bfloat16()
    _, y16 = _run_engine(s2v, cfg, sd, lat, text, ref, 500.0)
    cfg8 = copy.copy(cfg)
    cfg8.weight_format = "fp8"
    _, y8 = _run_engine(s2v, cfg8, sd, lat, text, ref, 500.0)
    cfgq = copy.copy(cfg)
    cfgq.weight_format = "fp8-qk"
    mq, yq = _run_engine(s2v, cfgq, sd, lat, text, ref, 500.0)
    assert torch.isfinite(yq.float()).all()
    rel16 = ((yq.float() - y16.float()).norm() / y16.float().norm()).item()
    rel8 = ((yq.float() - y8.float()).norm() / y8.float().norm()).item()
    print(f"MEASURED fp8-qk engine vs bf16: rel-l2 {rel16:.3e}; vs fp8 engine: {rel8:.3e}")
    assert 0 < rel16 <= FP8_ENGINE_BAR, rel16
    assert 0 < rel8 <= FP8_ENGINE_BAR, rel8
    eng = mq.engine
    y2 = eng.forward(lat, torch.tensor([500.0, 500.0]), shared_latent=True)
    torch.cuda.synchronize()
    assert torch.equal(y2, yq)
    sch = s2v.CogVideoXDDIMScheduler(snr_shift_scale=1.0)
    sch.set_timesteps(50)
    a, b = lat.clone().to(DEV), lat.clone().to(DEV)
    for x, graph in ((a, False), (b, True)):
        for i in range(2):
            t = sch.timesteps[i]
            eng.denoise_step(x, float(t), sch.coef(t, torch.bfloat16, 6.0), use_graph=graph)
    torch.cuda.synchronize()
    assert torch.isfinite(a.float()).all() and torch.equal(a, b)
