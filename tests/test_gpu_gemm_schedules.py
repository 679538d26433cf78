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
        lib.s2v_set_gemm_impl(9)  # the diagnostics library's default: the product's choice
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



@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (768, 512, 384), (1024, 768, 3072), (4096, 3072, 1024), (2560, 1024, 12288), (1152, 768, 1024)])
@pytest.mark.parametrize("epi", [0, 1])
def test_g4_generated_asm_kernel_matches_ring_bitwise_and_is_repeatable(s2v, M, N, K, epi):
    """gemm_g4 (gemm_g4.hip: four waves, generated-asm K loop, the product's kernel for plain operands with an even number of K-tiles
    >= 4) against the lock-step ring kernel: same products, same fp32 accumulation order per output element (k ascending in steps of
    16), same epilogue code -- so BIT-IDENTICAL, launch after launch; shapes: the minimum of four K-tiles, six, the C3 depths, a padded
    last column tile, an M of 4.5 row tiles (the op-level entry then takes the 128-row kernel for all of it: gemm_g4_ok refuses)"""
    L = s2v._lib
    g = torch.Generator().manual_seed(M + N + K + 1)
    A = (torch.randn(M, K, generator=g) * 0.5).bfloat16().to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(DEV)
    b = (torch.randn(N, generator=g) * 0.1).bfloat16().to(DEV)
    ref = run(L, A, W, b, M, N, K, epi, 5)
    assert torch.isfinite(ref.float()).all()
    for rep in range(6):
        out = run(L, A, W, b, M, N, K, epi, 9 if rep % 2 else None)  # diagnostics build with g4 selected / product build
        assert torch.equal(out, ref), f"rep {rep}: max diff {(out.float() - ref.float()).abs().max().item()}"


@pytest.mark.parametrize("M,N,K", [(8192, 4096, 1280), (9472, 3840, 1920), (8192, 4096, 3072), (6144, 7680, 12288)])
@pytest.mark.parametrize("epi", [0, 1])
def test_g4t_trickled_epilogue_kernel_is_bit_identical_to_g4(s2v, M, N, K, epi):
    """gemm_g4t (gemm_g4t.hip / gen_gemm_g4t.py: persistent, the previous tile's epilogue -- bias, rounding, GELU, LDS transposition,
    stores -- trickled through the MFMA gaps of the next tile's K loop, in generated asm; the product's kernel for the FF1 projection,
    attention.py:1237-1243) against gemm_g4 with the C++ epilogue: same products, same accumulation order, the same epilogue
    arithmetic instruction for instruction -> BIT-IDENTICAL, launch after launch, product build and diagnostics build.  Shapes: the
    minimum depth (K = 1280: the unrolled trickle is the whole loop), the 2B model's K = 1920 on a tile count that does not divide by
    the XCDs (37 x 15 = 555 tiles: uneven ranges, workgroups with two and three tiles), the C3 depths 3072 and 12288."""
    import ctypes

    L = s2v._lib
    D = L.diag_lib()
    D.s2v_set_gemm_g4t.argtypes = [ctypes.c_int]
    g = torch.Generator().manual_seed(M + N + K + 2)
    A = (torch.randn(M, K, generator=g) * 0.5).bfloat16().to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(DEV)
    b = (torch.randn(N, generator=g) * 0.2).bfloat16().to(DEV)
    try:
        D.s2v_set_gemm_g4t(0)
        ref = run(L, A, W, b, M, N, K, epi, 9)  # gemm_g4
        assert torch.isfinite(ref.float()).all()
        D.s2v_set_gemm_g4t(1)
        for rep in range(4):
            out = run(L, A, W, b, M, N, K, epi, 9 if rep % 2 else None)  # gemm_g4t: diagnostics build / product build
            assert torch.equal(out, ref), f"rep {rep}: {(out != ref).sum().item()} elements differ, max {(out.float() - ref.float()).abs().max().item()}"
    finally:
        D.s2v_set_gemm_g4t(1)
    y = A.float() @ W.float().T + b.float()
    if epi == 1:
        y = torch.nn.functional.gelu(y.bfloat16().float(), approximate="tanh")
    rel = ((ref.float() - y).norm() / y.norm()).item()
    assert rel <= 1e-2, rel


def _qkv_qknorm(D_lib, L, A, W, b, ln, cs, M, D, K, tok, text_len):
    C = torch.full((M, 3 * D), float("nan"), device=DEV, dtype=torch.bfloat16)
    rc = D_lib.s2v_diag_qkv_qknorm(L.ptr(A), L.ptr(W), L.ptr(b), L.ptr(ln[0]), L.ptr(ln[1]), L.ptr(ln[2]), L.ptr(ln[3]), L.ptr(cs), L.ptr(C), M, D, K, tok,
                                   text_len, ctypes.c_float(1e-6), L.stream_ptr())
    assert rc == 0, D_lib.s2v_last_error()
    torch.cuda.synchronize()
    return C


@pytest.mark.parametrize("M,D,K,tok,text", [(12288, 1024, 3072, 4096, 226), (16128, 768, 2304, 3000, 0), (9216, 1280, 3072, 1500, 1499), (12288, 1024, 4096, 12288, 12288)])
def test_g4t_trickled_qk_norm_rope_epilogue_is_bit_identical_to_g4(s2v, M, D, K, tok, text):
    """The fused QKV projection (bias, per-head LayerNorm(64) + affine, rotary embedding: attention_processor.py:2049-2080) with its epilogue
    trickled through the next tile's K loop in generated asm (gemm_g4t<qknorm>, gen_gemm_g4t.py qk_program) against gemm_g4 with the C++
    epilogue of gemm_epi.h: BIT-IDENTICAL -- the asm restates the C++ order of operations (octet sums, the three DPP steps, the correctly
    rounded sqrt and reciprocal, the separately rounded normalisation and rotary pair).  Shapes: several samples per launch with text
    rows inside tiles (tok 4096 / text 226), a token count that does not divide the tile (3000: tiles straddle samples) without text
    rows, all but one row text (1499 of 1500), no video row at all; the minimum depth (K = 2304) and the 5B depth (3072)."""
    L = s2v._lib
    Dg = L.diag_lib()
    Dg.s2v_set_gemm_g4t.argtypes = [ctypes.c_int]
    assert (M // 256) * (3 * D // 256) >= 512 and K >= 2304 and D % 256 == 0  # gemm_g4t_ok: two rounds of tiles, the trickle's depth
    g = torch.Generator().manual_seed(M + D + K)
    A = (torch.randn(M, K, generator=g) * 0.5).bfloat16().to(DEV)
    W = (torch.randn(3 * D, K, generator=g) * 0.05).bfloat16().to(DEV)
    b = (torch.randn(3 * D, generator=g) * 0.2).bfloat16().to(DEV)
    ln = [(1.0 + 0.3 * torch.randn(64, generator=g)).bfloat16().to(DEV), (0.2 * torch.randn(64, generator=g)).bfloat16().to(DEV),
          (1.0 + 0.3 * torch.randn(64, generator=g)).bfloat16().to(DEV), (0.2 * torch.randn(64, generator=g)).bfloat16().to(DEV)]
    ang = torch.rand(max(tok - text, 1), 32, generator=g) * 6.28
    cs = torch.cat([ang.cos(), ang.sin()], dim=1).float().contiguous().to(DEV)
    try:
        Dg.s2v_set_gemm_g4t(0)
        ref = _qkv_qknorm(Dg, L, A, W, b, ln, cs, M, D, K, tok, text)  # gemm_g4<qknorm>
        assert torch.isfinite(ref.float()).all()
        Dg.s2v_set_gemm_g4t(1)
        for rep in range(3):
            out = _qkv_qknorm(Dg, L, A, W, b, ln, cs, M, D, K, tok, text)
            bad = (out != ref)
            assert not bad.any(), (f"rep {rep}: {bad.sum().item()} elements differ, max {(out.float() - ref.float()).abs().max().item()}, first at "
                                   f"{bad.nonzero()[0].tolist()}, by column block {bad.view(M, 3 * D // 64, 64).any(2).any(0).nonzero().flatten().tolist()[:12]}")
    finally:
        Dg.s2v_set_gemm_g4t(1)
    # against plain torch: projection rounded to bf16, LayerNorm over each 64-column head of q and k, rotary on the video rows
    y = (A.float() @ W.float().T + b.float()).bfloat16().float()
    q = y[:, :2 * D].reshape(M, 2 * D // 64, 64)
    w = torch.cat([ln[0].float().expand(D // 64, 64), ln[2].float().expand(D // 64, 64)])
    bb = torch.cat([ln[1].float().expand(D // 64, 64), ln[3].float().expand(D // 64, 64)])
    q = (torch.nn.functional.layer_norm(q, (64,), eps=1e-6) * w + bb).bfloat16().float()
    r = torch.arange(M, device=DEV) % tok
    vid = r >= text
    pos = (r - text).clamp_min(0)
    c, s_ = cs[pos, :32].unsqueeze(1), cs[pos, 32:].unsqueeze(1)
    x0, x1 = q[..., 0::2], q[..., 1::2]
    rot = torch.stack([x0 * c - x1 * s_, x1 * c + x0 * s_], dim=-1).reshape(M, 2 * D // 64, 64)
    q = torch.where(vid[:, None, None], rot, q)
    exp = torch.cat([q.reshape(M, 2 * D), y[:, 2 * D:]], dim=1)
    rel = ((ref.float() - exp).norm() / exp.norm()).item()
    assert rel <= 1e-2, rel


@pytest.mark.parametrize("epi", [0, 1])
def test_four_wave_persistent_kernel_matches_pingpong(s2v, epi):
    """gemm_q4 (diagnostics build only: persistent 4-wave kernel whose epilogue trickles through the next tile's K loop): 1536
    output tiles = six per workgroup, so the in-loop epilogue, the tile hand-over and the final flush all run.  Bias goes through
    the matrix pipe there (one more K step) and GELU is evaluated in other slots, so the comparison is to one bf16 ulp."""
    L = s2v._lib
    M, N, K = 8192, 12288, 2304
    g = torch.Generator().manual_seed(7)
    A = (torch.randn(M, K, generator=g) * 0.5).bfloat16().to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(DEV)
    b = (torch.randn(N, generator=g) * 0.1).bfloat16().to(DEV)
    ref = run(L, A, W, b, M, N, K, epi, None).float()
    out = run(L, A, W, b, M, N, K, epi, 8).float()
    assert torch.isfinite(out).all()
    bad = ((out - ref).abs() > 2.0 ** -7 * ref.abs().clamp_min(1.0)).sum().item()
    assert bad == 0, f"{bad} elements differ by more than one bf16 ulp (max {(out - ref).abs().max().item()})"
    assert (out != ref).float().mean().item() < 2e-2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_fused_qk_norm_rope_epilogue_is_bit_identical_to_the_separate_kernel(s2v, dtype):
    """EPI_BIAS_QKNORM (QKV projection + per-head LayerNorm + rotary embedding in the GEMM epilogue) against the plain projection
    followed by qk_norm_rope_k: the diagnostics build can switch the fusion off, everything else being equal the two forwards must
    agree bit for bit -- with rotary tables (5B-style) and without (2B-style), ragged token count, text rows in the middle of a tile;
    in bf16 and (round 5) in the fp16 model dtype, whose engine takes the same fused epilogue on the kernels' fp16 instantiations."""
    L = s2v._lib
    diag = L.diag_lib()
    prev = L._lib
    L.lib()
    try:
        L._lib = diag  # engines created below bind the diagnostics library
        for use_rope in (True, False):
            cfg = s2v.tiny(use_rope=use_rope)
            sd = s2v.weights.synthetic_state_dict(cfg, seed=11, parity=True)
            outs = []
            for fused in (1, 0):
                diag.s2v_set_fused_qk(fused)
                eng = s2v.S2VEngine(cfg, dtype, DEV)
                eng.load_state_dict(sd)
                eng.set_geometry(2, 5, 3, 10, 14)
                eng.prepare_tables(80, 112)
                gq = torch.Generator().manual_seed(12)
                eng.set_conditioning(torch.randn(2, 5, cfg.text_embed_dim, generator=gq), torch.randn(1, 1, 16, 10, 14, generator=gq))
                outs.append(eng.forward(torch.randn(2, 3, 16, 10, 14, generator=gq), torch.tensor([300.0, 300.0])).clone())
                torch.cuda.synchronize()
            assert torch.isfinite(outs[0].float()).all()
            assert torch.equal(outs[0], outs[1]), f"rope={use_rope}: max diff {(outs[0].float() - outs[1].float()).abs().max().item()}"
    finally:
        diag.s2v_set_fused_qk(1)
        L._lib = prev


def test_persistent_attention_launch_is_bit_identical_to_one_workgroup_per_q_block(s2v):
    """the engine launches attention as a persistent, work-pulling grid (one workgroup per CU, per-XCD queues with stealing, counters
    that reset themselves); the diagnostics build can force the one-workgroup-per-q-block launch.  5B width, the full 19126 tokens
    (7200 work items on 256 CUs), two layers, two consecutive forwards (the second one finds the counters the first one left)."""
    L = s2v._lib
    diag = L.diag_lib()
    prev = L._lib
    L.lib()
    try:
        L._lib = diag
        cfg = s2v.cogvideox_5b()
        cfg.num_layers = 2
        sd = s2v.weights.synthetic_state_dict(cfg, seed=3, device=DEV, parity=True)
        eng = s2v.S2VEngine(cfg, torch.bfloat16, DEV)
        eng.load_state_dict(sd)
        del sd
        F, H, W, T = 13, 60, 90, 226
        g = torch.Generator(device=DEV).manual_seed(4)
        eng.set_geometry(2, T, F, H, W)
        eng.prepare_tables(480, 720)
        eng.set_conditioning(torch.randn(2, T, 4096, generator=g, device=DEV), torch.randn(1, 1, 16, H, W, generator=g, device=DEV) * 0.7)
        lat = torch.randn(1, F, 16, H, W, generator=g, device=DEV).bfloat16()
        ts = torch.tensor([500.0, 500.0])
        outs = []
        for variant in (0, 0, 3):
            diag.s2v_set_attn_variant(variant)
            outs.append(eng.forward(lat, ts, shared_latent=True).clone())
            torch.cuda.synchronize()
        assert torch.isfinite(outs[0].float()).all()
        assert torch.equal(outs[0], outs[1]), "second persistent launch differs: the queue was not reset"
        assert torch.equal(outs[0], outs[2]), "persistent launch differs from the per-q-block launch"
    finally:
        diag.s2v_set_attn_variant(0)
        L._lib = prev


@pytest.mark.parametrize("B,H,N", [(2, 30, 1250), (1, 3, 130), (2, 2, 700), (1, 1, 64), (1, 2, 4608), (1, 2, 4700)])
def test_attention_kernel_choice_by_sequence_length(s2v, B, H, N):
    """launch_attn_bf16 runs the eight-wave ping-pong kernel (attn_pp) up to 4608 tokens and the four-wave attn_q4 beyond (attention.hip;
    both replace F.scaled_dot_product_attention, attention_processor.py:2083-2087).  Both kernels are held against fp64 SDPA on the same
    bf16 inputs at every length -- the product no longer reaches attn_q4's short-sequence paths (all tiles in its last-five-tiles phase),
    the diagnostics switch does -- and against each other within bf16 rounding."""
    L = s2v._lib
    diag = L.diag_lib()
    D = H * 64
    g = torch.Generator(device=DEV).manual_seed(N)
    qkv = torch.randn(B * N + 64, 3 * D, generator=g, device=DEV).bfloat16()
    qkv[5, D : D + 64] *= 6.0  # a spiked key row
    vt = torch.zeros(B * H * 64 * ((N + 63) // 64 * 64), dtype=torch.bfloat16, device=DEV)
    outs = {}
    try:
        for variant in (6, 10, 0):  # attn_q4, attn_pp, the product's choice
            diag.s2v_set_attn_variant(variant)
            out = torch.full((B * N, D), float("nan"), device=DEV, dtype=torch.bfloat16)
            L.check(diag.s2v_op_attention(L.ptr(qkv), L.ptr(vt), L.ptr(out), B, H, N, 1, 0, L.stream_ptr()))
            torch.cuda.synchronize()
            outs[variant] = out
    finally:
        diag.s2v_set_attn_variant(0)
    assert torch.equal(outs[0], outs[10] if N <= 4608 else outs[6]), "the product did not pick the kernel its rule names"
    q, k, v = (qkv[: B * N, i * D : (i + 1) * D].reshape(B, N, H, 64).transpose(1, 2).double().cpu() for i in range(3))
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B * N, D)
    scale = max(1.0, ref.abs().max().item())
    for variant in (6, 10):
        got = outs[variant].double().cpu()
        assert torch.isfinite(got).all()
        assert (got - ref).abs().max().item() <= 2e-2 * scale, (variant, (got - ref).abs().max().item())  # test_op_attention's tolerance
    assert (outs[6].double() - outs[10].double()).abs().max().item() <= 1.6e-2 * scale


@pytest.mark.parametrize("M,N,K,epi", [(512, 512, 7680, 0), (256, 1024, 4096, 1), (1024, 256, 3072, 0)])
def test_split_k_few_tile_gemm(s2v, M, N, K, epi):
    """gemm_g4 with K split over S workgroups per tile (api.hip linear() / choose_splitk: the FF2 of the short-sequence geometries,
    attention.py:1241-1243 at M = 2500): partial tiles in fp32, the last workgroup to arrive adds them in split order -- so the result
    is a pure function of the inputs (repeated launches bit-identical, arrival counters back at zero) -- and differs from the
    one-workgroup sum only by fp32 reassociation (<= 1 bf16 ulp of the largest output), both within bf16 rounding of the fp64 product"""
    L = s2v._lib
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().to(DEV)
    b = torch.randn(N, generator=g).bfloat16().to(DEV)
    outs = []
    for impl in (2, 2, 0):
        C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        L.check(L.lib().s2v_op_linear(L.ptr(A), L.ptr(W), L.ptr(b), L.ptr(C), M, N, K, epi, L.DTYPE_BF16, impl, L.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(C.float().cpu())
    assert torch.equal(outs[0], outs[1])
    ref = A.double().cpu() @ W.double().cpu().T + b.double().cpu()
    if epi == 1:
        ref = torch.nn.functional.gelu(ref.float().bfloat16().double(), approximate="tanh")
    scale = ref.abs().max().item()
    assert (outs[0] - outs[2]).abs().max().item() <= 2 ** -7 * scale
    for o in (outs[0], outs[2]):
        assert (o.double() - ref).abs().max().item() <= 2 ** -7 * scale
