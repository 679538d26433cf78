"""GPU parity tests (pytest -m gpu): every call goes through the C ABI of libs2v_hip.so and is compared with the CPU
oracle (oracle/*.py) on the same seeded inputs and with the golden vectors captured from the reference.

Tolerances (stated per BASELINE.json north_star: "<= 1e-3 max-abs latent deviation vs the CPU reference" is the fp32
bar; bf16 / fp16 are compared against the reference's own bf16 / fp16 CPU run at the same rounding points).  The bars are 2 x the
worst value measured over this file (BARS / F32_BAR below):
  fp32 path : max-abs <= 4e-5 (measured <= 1.8e-5; north star 1e-3)
  bf16 path : relative L2 <= 1.3e-2 and max-abs <= 2e-2 * max|ref|  (measured <= 6.4e-3 / 9.5e-3)
  fp16 path : relative L2 <= 1.3e-3 and max-abs <= 2e-3 * max|ref|  (measured <= 6.4e-4 / 9.3e-4)
  scheduler : bit-exact in all three dtypes
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import load_golden, weights_of
from oracle import sched_ref, transformer_ref as tr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(x, dt=torch.float32):
    return torch.from_numpy(np.asarray(x)).to(dt)


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
# Bars = 2 x the worst value any test of this file measured in round 5 (47 comparisons, gpurun_out/measured_tols.txt: bf16 rel-L2 6.4e-3,
# max-abs 9.5e-3 max|ref|; fp16 6.4e-4 / 9.3e-4; fp32 max-abs 1.8e-5) -- a kernel regression of 2 x fails.  Until round 4 the bf16 bars were
# 2e-2 / 6e-2.  fp16 (src/inference.py:191,209 runs every non-5B checkpoint in it): the same rounding points as bf16, 8 x finer ulps.
# fp32: the north-star bar is 1e-3 on the latents; the tests hold the kernels to 4e-5.
BARS = {"bf16": (1.3e-2, 2e-2), "f16": (1.3e-3, 2e-3)}
F32_BAR = 4e-5


def assert_close(got, exp, dt_name, what=""):
    got, exp = got.float().cpu(), exp.float().cpu()
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    err = (got - exp).abs().max().item()
    if dt_name == "f32":
        print(f"MEASURED {dt_name} {what}: max-abs {err:.3e}")
        assert err <= F32_BAR, f"{what}: max-abs {err}"
    else:
        r = rel_l2(got, exp)
        br, ba = BARS[dt_name]
        print(f"MEASURED {dt_name} {what}: rel-l2 {r:.3e} max-abs/max|ref| {err / exp.abs().max().item():.3e}")
        assert r <= br and err <= ba * exp.abs().max().item(), f"{what}: rel-l2 {r}, max-abs {err} (max|ref| {exp.abs().max().item()})"


# ------------------------------------------------------------------------------------------------ operators
@pytest.mark.parametrize("M,N,K,epi", [(256, 256, 128, 0), (384, 128, 3072, 0), (128, 512, 1024, 1), (1024, 384, 64, 0)])
def test_op_linear_mfma_bf16(s2v, M, N, K, epi):
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * 0.5).bfloat16()
    W = (torch.randn(N, K, generator=g) * 0.5).bfloat16()  # asymmetric, non-identity
    b = torch.randn(N, generator=g).bfloat16()
    ref = A.float() @ W.float().T + b.float()
    if epi == 1:
        ref = torch.nn.functional.gelu(ref.bfloat16().float(), approximate="tanh")
    Ad, Wd, bd = A.to(DEV), W.to(DEV), b.to(DEV)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    L = s2v._lib
    L.check(L.lib().s2v_op_linear(L.ptr(Ad), L.ptr(Wd), L.ptr(bd), L.ptr(C), M, N, K, epi, L.DTYPE_BF16, 0, L.stream_ptr()))
    torch.cuda.synchronize()
    got = C.float().cpu()
    assert rel_l2(got, ref) < 4e-3, rel_l2(got, ref)
    assert (got - ref).abs().max() <= 2e-2 * ref.abs().max() + 1e-2


@pytest.mark.parametrize("dt_name", ["f32", "bf16"])
def test_op_linear_generic(s2v, dt_name):
    dt = torch.float32 if dt_name == "f32" else torch.bfloat16
    g = torch.Generator().manual_seed(3)
    M, N, K = 77, 52, 40  # ragged on purpose
    A, W, b = torch.randn(M, K, generator=g).to(dt), torch.randn(N, K, generator=g).to(dt), torch.randn(N, generator=g).to(dt)
    ref = A.float() @ W.float().T + b.float()
    C = torch.empty(M, N, dtype=dt, device=DEV)
    L = s2v._lib
    Ad, Wd, bd = A.to(DEV), W.to(DEV), b.to(DEV)  # keep alive: the C ABI borrows raw pointers
    L.check(L.lib().s2v_op_linear(L.ptr(Ad), L.ptr(Wd), L.ptr(bd), L.ptr(C), M, N, K, 0,
                                  L.DTYPE_OF[dt], 1, L.stream_ptr()))
    torch.cuda.synchronize()
    tol = 1e-4 if dt_name == "f32" else 3e-2
    assert (C.float().cpu() - ref).abs().max() <= tol * ref.abs().max()


@pytest.mark.parametrize("B,H,N,impl,dt_name", [(1, 2, 200, 0, "bf16"), (2, 3, 1250, 0, "bf16"), (1, 1, 64, 0, "bf16"),
                                                (1, 2, 129, 1, "f32"), (1, 2, 129, 1, "bf16"), (2, 2, 300, 1, "f32"),
                                                # impl 4 = attn_q4h (attn_p_format 1: fp16 P / V^T, packed fp16 row sums, deferred maximum 2^14) at
                                                # any length: one tile, every tile through the rare-path handler, ragged tails, phase A + B
                                                (1, 1, 64, 4, "bf16"), (1, 2, 200, 4, "bf16"), (2, 3, 449, 4, "bf16"), (2, 3, 1250, 4, "bf16"),
                                                (1, 2, 5000, 4, "bf16"), (1, 2, 5000, 3, "bf16"), (1, 2, 5000, 0, "bf16")])
def test_op_attention(s2v, B, H, N, impl, dt_name):
    dt = torch.float32 if dt_name == "f32" else torch.bfloat16
    g = torch.Generator().manual_seed(N)
    D = H * 64
    qkv = torch.randn(B * N, 3 * D, generator=g).to(dt)
    qkv[5, D : D + 64] *= 6.0  # a spiked key row: forces large online-softmax rescales
    q, k, v = (qkv.float()[:, i * D : (i + 1) * D].reshape(B, N, H, 64).transpose(1, 2) for i in range(3))
    ref = torch.nn.functional.scaled_dot_product_attention(q.double(), k.double(), v.double()).transpose(1, 2).reshape(B * N, D)
    pad = torch.zeros(64, 3 * D, dtype=dt)
    qd = torch.cat([qkv, pad]).to(DEV)
    out = torch.empty(B * N, D, dtype=dt, device=DEV)
    vt = torch.zeros(B * H * 64 * ((N + 63) // 64 * 64), dtype=torch.bfloat16, device=DEV)
    L = s2v._lib
    L.check(L.lib().s2v_op_attention(L.ptr(qd), L.ptr(vt), L.ptr(out), B, H, N, L.DTYPE_OF[dt], impl, L.stream_ptr()))
    torch.cuda.synchronize()
    got = out.float().cpu().double()
    tol = 2e-5 if dt_name == "f32" else 2e-2
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max() <= tol * max(1.0, ref.abs().max().item()), (got - ref).abs().max()


@pytest.mark.parametrize("M,N,K,epi", [(128, 128, 64, 0), (384, 256, 3072, 0), (256, 512, 1024, 1), (1280, 1920 + 128, 1920, 0),
                                        (4096, 4096, 1024, 0), (4096, 4096, 512, 1)])   # the last two: >= 256 tiles of 256 x 256 -> gemm_g4 on fp16 operands
def test_op_linear_mfma_f16(s2v, M, N, K, epi):
    """fp16 operands on v_mfma_f32_32x32x16_f16 (impl 4: what the fp16 engine's linears run) against fp64 on the same fp16 values"""
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * 0.5).half()
    W = (torch.randn(N, K, generator=g) * 0.5).half()  # asymmetric, non-identity
    b = torch.randn(N, generator=g).half()
    ref = (A.double() @ W.double().T + b.double()).float()
    if epi == 1:
        ref = torch.nn.functional.gelu(ref.half().float(), approximate="tanh")
    Ad, Wd, bd = A.to(DEV), W.to(DEV), b.to(DEV)
    C = torch.full((M, N), float("nan"), dtype=torch.float16, device=DEV)
    L = s2v._lib
    L.check(L.lib().s2v_op_linear(L.ptr(Ad), L.ptr(Wd), L.ptr(bd), L.ptr(C), M, N, K, epi, L.DTYPE_F16, 4, L.stream_ptr()))
    torch.cuda.synchronize()
    got = C.float().cpu()
    assert torch.isfinite(got).all()
    assert rel_l2(got, ref) < 5e-4, rel_l2(got, ref)
    assert (got - ref).abs().max() <= 2.5e-3 * ref.abs().max() + 1.25e-3
    # the generic VALU kernel on the same operands (impl 1) agrees to fp16 rounding of the same fp32-accumulated sums
    C1 = torch.empty_like(C)
    L.check(L.lib().s2v_op_linear(L.ptr(Ad), L.ptr(Wd), L.ptr(bd), L.ptr(C1), M, N, K, epi, L.DTYPE_F16, 1, L.stream_ptr()))
    torch.cuda.synchronize()
    assert (C1.float().cpu() - got).abs().max() <= 2e-3 * ref.abs().max() + 1e-3


@pytest.mark.parametrize("B,H,N,impl", [(1, 2, 129, 5), (2, 3, 300, 5), (1, 1, 32, 5), (1, 2, 1000, 5), (1, 2, 129, 1),
                                        (1, 2, 129, 6), (2, 3, 300, 6), (1, 1, 64, 6), (1, 2, 1000, 6), (2, 2, 5000, 6)])
def test_op_attention_f16(s2v, B, H, N, impl):
    """fp16 storage: the lock-step kernel on v_mfma_f32_32x32x16_f16 (impl 6: what the fp16 engine runs -- q, k, V^T and P in fp16, per-tile row
    maximum so p <= 1), attn_f32m<f16_t> (impl 5; exact fp16 x fp16 products on the fp32 matrix pipe, fp16 probabilities into P.V) and the VALU
    kernel (impl 1) against fp64 SDPA on the same fp16 values"""
    g = torch.Generator().manual_seed(N)
    D = H * 64
    qkv = torch.randn(B * N, 3 * D, generator=g).half()
    qkv[5, D:D + 64] *= 6.0
    q, k, v = (qkv.double()[:, i * D:(i + 1) * D].reshape(B, N, H, 64).transpose(1, 2) for i in range(3))
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B * N, D)
    qd = torch.cat([qkv, torch.zeros(64, 3 * D, dtype=torch.float16)]).to(DEV)
    out = torch.full((B * N, D), float("nan"), dtype=torch.float16, device=DEV)
    L = s2v._lib
    vt = torch.zeros(B * H * 64 * ((N + 63) // 64 * 64), dtype=torch.float16, device=DEV) if impl == 6 else None
    L.check(L.lib().s2v_op_attention(L.ptr(qd), L.ptr(vt), L.ptr(out), B, H, N, L.DTYPE_F16, impl, L.stream_ptr()))
    torch.cuda.synchronize()
    got = out.float().cpu().double()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max() <= 2.5e-3 * max(1.0, ref.abs().max().item()), (got - ref).abs().max()


# ------------------------------------------------------------------------------------------------ golden: tiny model
def _tiny_model(s2v, variant, dt, g, force_simple=False):
    cfg = s2v.tiny(use_rope=variant == "rope")
    m = s2v.HipCogVideoXTransformer3DModel(cfg, dt, DEV, force_simple)
    m.load_state_dict(weights_of(g))
    return m


@pytest.mark.parametrize("variant", ["rope", "sincos"])
@pytest.mark.parametrize("dt_name,simple", [("f32", False), ("bf16", False), ("bf16", True), ("f16", False), ("f16", True)])
def test_transformer_tiny_vs_reference_golden(s2v, variant, dt_name, simple):
    gw = load_golden("transformer_tiny_rope.npz")
    g = load_golden(f"transformer_tiny_{variant}.npz")
    dt = DT[dt_name]
    m = _tiny_model(s2v, variant, dt, gw, simple)
    kw = {}
    if variant == "rope":
        cos, sin = t(gw["rope_cos"]).to(DEV), t(gw["rope_sin"]).to(DEV)
        kw = dict(image_rotary_emb=(cos[16:], sin[16:]), ref_image_rotary_emb=(cos[:16], sin[:16]))
    y = m(hidden_states=t(g["lat"], dt).to(DEV), encoder_hidden_states=t(g["text"], dt).to(DEV),
          ref_img_states=t(g["ref"], dt).to(DEV), timestep=t(g["timestep"], torch.int64).to(DEV),
          return_dict=False, eval=True, **kw)[0]
    torch.cuda.synchronize()
    assert_close(y, t(g[f"out_{dt_name}"]), dt_name, f"transformer {variant}")
    # protocol errors of the reference are kept
    with pytest.raises(RuntimeError):
        m(hidden_states=t(g["lat"], dt).to(DEV)[:1], encoder_hidden_states=t(g["text"], dt).to(DEV)[:1],
          ref_img_states=t(g["ref"], dt).to(DEV), timestep=t(g["timestep"], torch.int64)[:1], eval=True, **kw)


@pytest.mark.parametrize("dt_name", ["f32", "bf16", "f16"])
def test_block_and_attnprocessor_seams_vs_reference_golden(s2v, dt_name):
    g = load_golden("transformer_tiny_rope.npz")
    dt = DT[dt_name]
    m = _tiny_model(s2v, "rope", dt, g)
    cos, sin = t(g["rope_cos"]).to(DEV), t(g["rope_sin"]).to(DEV)
    rope, ref_rope = (cos[16:], sin[16:]), (cos[:16], sin[:16])
    h, e0, e1, temb = (t(g[k], dt).to(DEV) for k in ("blk_h", "blk_e0", "blk_e1", "blk_temb"))
    blk = m.transformer_blocks[1]
    oh, o0, o1 = blk(hidden_states=h, encoder_hidden_states=e0, temb=temb, enc_hidden_states1=e1, image_rotary_emb=rope,
                     embed_ref_img=True, ref_img_seq_start=5, ref_img_seq_end=21, position_delta=0,
                     ref_image_rotary_emb=ref_rope, timestep=None, layer=1)
    torch.cuda.synchronize()
    for got, key in ((oh, "blk_out_h"), (o0, "blk_out_e0"), (o1, "blk_out_e1")):
        assert_close(got, t(g[f"{key}_{dt_name}"]), dt_name, key)

    # AttnProcessor protocol: weights are borrowed from a duck-typed Attention module
    sd = weights_of(g, dt)
    p = "transformer_blocks.1.attn1."

    class Lin:
        def __init__(self, w, b):
            self.weight, self.bias = w.to(DEV), b.to(DEV)

    class Attn:
        heads = 2
        is_cross_attention = False
        to_q = Lin(sd[p + "to_q.weight"], sd[p + "to_q.bias"])
        to_k = Lin(sd[p + "to_k.weight"], sd[p + "to_k.bias"])
        to_v = Lin(sd[p + "to_v.weight"], sd[p + "to_v.bias"])
        to_out = [Lin(sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])]
        norm_q = Lin(sd[p + "norm_q.weight"], sd[p + "norm_q.bias"])
        norm_k = Lin(sd[p + "norm_k.weight"], sd[p + "norm_k.bias"])

    proc = s2v.HipCogVideoXAttnProcessor2_0()
    ah, ae = proc(Attn(), h, torch.cat([e0, e1], dim=1), attention_mask=None, image_rotary_emb=rope,
                  ref_img_seq_start=5, ref_img_seq_end=21, position_delta=0, embed_ref_img=True,
                  ref_image_rotary_emb=ref_rope)
    torch.cuda.synchronize()
    assert_close(ah, t(g[f"attn_out_h_{dt_name}"]), dt_name, "attn hidden")
    assert_close(ae, t(g[f"attn_out_e_{dt_name}"]), dt_name, "attn encoder")
    with pytest.raises(NotImplementedError):
        proc(Attn(), h, torch.cat([e0, e1], dim=1), attention_mask=torch.ones(1, device=DEV), embed_ref_img=True,
             ref_img_seq_start=5, ref_img_seq_end=21)


# ------------------------------------------------------------------------------------------------ schedulers
@pytest.mark.parametrize("kind", ["ddim", "dpm"])
@pytest.mark.parametrize("dt_name", ["f32", "bf16", "f16"])
def test_scheduler_step_bit_exact_vs_reference_golden(s2v, kind, dt_name):
    g = load_golden(f"sched_{kind}_{dt_name}_10.npz")
    dt = DT[dt_name]
    cls = s2v.CogVideoXDDIMScheduler if kind == "ddim" else s2v.CogVideoXDPMScheduler
    sch = cls(snr_shift_scale=float(g["snr"]))
    sch.set_timesteps(10)
    ac = sched_ref.alphas_cumprod(float(g["snr"]))
    for i in range(10):
        tt = sch.timesteps[i]
        v = sched_ref.cfg_combine(t(g[f"noise_pred_{i}"], dt), 6.0).to(DEV)
        lat = t(g[f"lat_in_{i}"], dt).to(DEV)
        if kind == "ddim":
            prev, x0 = sch.step(v, tt, lat, return_dict=False)
        else:
            # reproduce the reference's generator use: seed 1000+i, first draw discarded on multistep steps
            gen = torch.Generator().manual_seed(1000 + i)
            old = t(g[f"x0_{i-1}"]).to(DEV) if i > 0 else None
            prev, x0 = sch.step(v, old, tt, sch.timesteps[i - 1] if i > 0 else None, lat, generator=gen)
        torch.cuda.synchronize()
        assert prev.dtype == torch.float32
        np.testing.assert_array_equal(x0.cpu().numpy(), g[f"x0_{i}"], err_msg=f"x0 step {i}")
        np.testing.assert_array_equal(prev.to(dt).float().cpu().numpy(), g[f"lat_out_{i}"], err_msg=f"latents step {i}")


# ------------------------------------------------------------------------------------------------ denoise loop
def _pipe_from_golden(s2v, g, kind, with_vae=False):
    cfg = s2v.tiny(use_rope=True, text_dim=64, temb=64)
    cfg.max_text_seq_length = 6
    m = s2v.HipCogVideoXTransformer3DModel(cfg, torch.float32, DEV)
    m.load_state_dict(weights_of(g))
    sch = (s2v.CogVideoXDDIMScheduler if kind == "ddim" else s2v.CogVideoXDPMScheduler)(snr_shift_scale=1.0)
    vae = None
    if with_vae:
        vcfg = s2v.VAEConfig(block_out_channels=(8, 8, 8, 8), layers_per_block=1, norm_num_groups=2, latent_channels=16,
                             sample_height=480, sample_width=720, scaling_factor=0.7, temporal_compression_ratio=4)
        vae = s2v.HipAutoencoderKLCogVideoX(vcfg, torch.float32, DEV)
        vae.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("vae:")})
    return s2v.S2VPipeline(m, sch, vae)


def _pipe_args(g):
    return dict(prompt_embeds=t(g["prompt_embeds"]), negative_prompt_embeds=t(g["negative_prompt_embeds"]), ref_img_states=t(g["ref"]),
                height=480, width=720, num_frames=5, num_inference_steps=3, guidance_scale=6.0,
                generator=torch.Generator().manual_seed(int(g["dpm_noise_seed"])), latents=t(g["latents0"]), return_dict=False)


def _check_steps(got, g, name, atol=1e-3):
    x = torch.stack([y.float().cpu() for y in got])
    exp = t(g[name + "_sub2"])
    assert (x[..., ::2, ::2] - exp).abs().max().item() <= atol
    sums = torch.stack([x.double().sum(dim=(1, 2, 3, 4, 5)), x.double().abs().sum(dim=(1, 2, 3, 4, 5))], dim=1).numpy()
    np.testing.assert_allclose(sums[:, 1], g[name + "_sums"][:, 1], rtol=1e-4)


@pytest.mark.parametrize("kind", ["ddim", "dpm"])
@pytest.mark.parametrize("mode", ["fused", "fused_graph", "seams"])
def test_pipeline_three_steps_vs_reference_golden(s2v, kind, mode):
    """S2VPipeline against CustomCogVideoXPipeline.__call__ itself (tests/golden/pipeline_tiny.npz): the latents after EVERY step
    (what scheduler.step returned to the reference loop, custom_cogvideox_pipe.py:280-296), the CFG-combined noise prediction of
    every step (what the loop handed to scheduler.step, :273-277; the fused modes expose it through s2v_last_noise_pred) and the
    final latents, in the three execution modes"""
    g = load_golden("pipeline_tiny.npz")
    pipe = _pipe_from_golden(s2v, g, kind)
    eng = pipe.transformer.engine
    lats, nps = [], []

    def on_step(p_, i, tt, kw):
        lats.append(kw["latents"].clone())
        if mode != "seams":
            npr = eng.last_noise_pred().float()
            u, c = npr.chunk(2)
            nps.append(u + 6.0 * (c - u))

    out = pipe(output_type="latent", fused=mode != "seams", use_graph=mode == "fused_graph", callback_on_step_end=on_step, **_pipe_args(g))[0]
    torch.cuda.synchronize()
    err = (out.float().cpu() - t(g[f"final_{kind}"])).abs().max().item()
    assert err <= 1e-3, err
    _check_steps(lats, g, f"steps_latents_{kind}")
    if nps:
        _check_steps(nps, g, f"steps_noise_pred_{kind}")


@pytest.mark.parametrize("dt_name", ["bf16", "f16"])
@pytest.mark.parametrize("mode", ["fused_graph", "seams"])
def test_pipeline_three_steps_reduced_precision_vs_reference_pipeline_golden(s2v, dt_name, mode):
    """S2VPipeline in bf16 / fp16 against CustomCogVideoXPipeline.__call__ ITSELF run in that dtype on the CPU (round 5 fixtures
    final_ddim_{bf16,f16}, steps_latents_ddim_*: the loop's rounding points -- `latents.to(prompt_embeds.dtype)`, the scheduler on
    reduced-precision samples -- pinned to the reference, which the CPU oracle reproduces bit for bit): latents after every step and at the end"""
    g = load_golden("pipeline_tiny.npz")
    dt = DT[dt_name]
    cfg = s2v.tiny(use_rope=True, text_dim=64, temb=64)
    cfg.max_text_seq_length = 6
    m = s2v.HipCogVideoXTransformer3DModel(cfg, dt, DEV)
    m.load_state_dict(weights_of(g))
    pipe = s2v.S2VPipeline(m, s2v.CogVideoXDDIMScheduler(snr_shift_scale=1.0), None)
    lats = []
    args = _pipe_args(g)
    for k in ("prompt_embeds", "negative_prompt_embeds", "ref_img_states", "latents"):
        args[k] = args[k].to(dt)
    out = pipe(output_type="latent", fused=mode != "seams", use_graph=mode == "fused_graph",
               callback_on_step_end=lambda p_, i, tt, kw: lats.append(kw["latents"].float().cpu().clone()), **args)[0]
    torch.cuda.synchronize()
    assert out.dtype == dt
    # three COARSE steps of a 3-step schedule with CFG 6 compound the per-forward deviation (guidance multiplies the difference of two
    # predictions by six): measured bf16 rel-L2 2.2e-2 / max-abs 3.0e-2 max|ref| at the end, fp16 3.3e-3 / 4.4e-3; bars = 2 x measured
    br, ba = {"bf16": (4.4e-2, 6.0e-2), "f16": (6.6e-3, 9.0e-3)}[dt_name]
    exp_steps = t(g[f"steps_latents_ddim_{dt_name}_sub2"])
    prev = 0.0
    for i, x in enumerate(lats + [out.float().cpu()]):
        e = exp_steps[i] if i < len(lats) else t(g[f"final_ddim_{dt_name}"])
        y = x[..., ::2, ::2] if i < len(lats) else x
        assert torch.isfinite(y).all()
        r, m = rel_l2(y, e), (y - e).abs().max().item() / e.abs().max().item()
        print(f"MEASURED {dt_name} pipeline {mode} {'step %d' % (i + 1) if i < len(lats) else 'final'}: rel-l2 {r:.3e} max-abs/max|ref| {m:.3e}")
        assert r <= br and m <= ba, (i, r, m)
    assert torch.equal(lats[-1], out.float().cpu())


@pytest.mark.parametrize("mode", ["fused_graph", "seams"])
def test_pipeline_dynamic_cfg_vs_reference_golden(s2v, mode):
    """use_dynamic_cfg=True (custom_cogvideox_pipe.py:268-271): the guidance scale changes every step, so the captured graph must read
    it from device memory like the scheduler scalars"""
    g = load_golden("pipeline_tiny.npz")
    pipe = _pipe_from_golden(s2v, g, "ddim")
    lats = []
    out = pipe(output_type="latent", use_dynamic_cfg=True, fused=mode != "seams", use_graph=mode == "fused_graph",
               callback_on_step_end=lambda p_, i, tt, kw: lats.append(kw["latents"].clone()), **_pipe_args(g))[0]
    torch.cuda.synchronize()
    err = (out.float().cpu() - t(g["final_ddim_dyncfg"])).abs().max().item()
    assert err <= 1e-3, err
    _check_steps(lats, g, "steps_latents_ddim_dyncfg")


def test_pipeline_frames_with_tiled_vae_vs_reference_golden(s2v):
    """output_type="np" with VAE tiling enabled, as src/inference.py:204-207 runs the reference: denoise (graph) -> decode_latents over
    nine blended tiles -> postprocess_video, against the REFERENCE PIPELINE's own frames (8 x 480 x 720 x 3)"""
    g = load_golden("pipeline_tiny.npz")
    pipe = _pipe_from_golden(s2v, g, "ddim", with_vae=True)
    pipe.vae.enable_tiling()
    frames = pipe(output_type="np", fused=True, use_graph=True, **_pipe_args(g))[0]
    assert list(frames.shape) == list(g["frames_ddim_tiled_shape"]) == [1, 8, 480, 720, 3]
    assert np.abs(frames[:, :, ::8, ::8, :] - g["frames_ddim_tiled_sub8"]).max() <= 1e-3
    np.testing.assert_allclose(frames.astype(np.float64).sum(axis=(2, 3)), g["frames_ddim_tiled_sums"], rtol=1e-4)


# ------------------------------------------------------------------------------------------------ on-box oracle
@pytest.mark.parametrize("dt_name", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("use_rope", [True, False])
def test_medium_model_vs_oracle(s2v, dt_name, use_rope):
    """heads=3 (D=192: exercises the 128-tile padding), 2 layers, 3 frames of 16x24 latents, T=7: N = 7+96+288."""
    dt = DT[dt_name]
    cfg = s2v.tiny(use_rope=use_rope, heads=3, layers=2, text_dim=128, temb=64)
    cfg.max_text_seq_length = 7
    sd = s2v.weights.synthetic_state_dict(cfg, seed=5, parity=True)
    lora = s2v.weights.synthetic_lora(cfg, rank=8, seed=6, std=0.05)
    g = torch.Generator().manual_seed(17)
    B, F, C, H, W, T = 2, 3, 16, 16, 24, 7
    lat = torch.randn(B, F, C, H, W, generator=g).to(dt)
    text = torch.randn(B, T, 128, generator=g).to(dt)
    ref = (torch.randn(1, 1, C, H, W, generator=g) * 0.7).to(dt)
    ts = torch.tensor([500, 500])
    ocfg = dict(num_heads=3, num_layers=2, use_rope=use_rope, norm_eps=1e-5)
    rope = ref_rope = None
    kw = {}
    if use_rope:
        ref_rope, rope = tr.pipeline_rope(H * 8, W * 8, F)
        kw = dict(image_rotary_emb=tuple(x.to(DEV) for x in rope), ref_image_rotary_emb=tuple(x.to(DEV) for x in ref_rope))
    merged = tr.merge_lora(sd, lora, 0.5)
    with torch.no_grad():
        exp = tr.transformer_forward({k: v.to(dt) for k, v in merged.items()}, ocfg, lat, text, ref, ts, rope, ref_rope)
    m = s2v.HipCogVideoXTransformer3DModel(cfg, dt, DEV)
    m.load_state_dict(sd, lora=lora, lora_scale=0.5)
    y = m(hidden_states=lat.to(DEV), encoder_hidden_states=text.to(DEV), ref_img_states=ref.to(DEV), timestep=ts.to(DEV),
          return_dict=False, eval=True, **kw)[0]
    torch.cuda.synchronize()
    assert_close(y, exp, dt_name, "medium transformer")


# ------------------------------------------------------------------------------------------------ BASELINE configs[0]
@pytest.mark.parametrize("dt_name", ["f32", "bf16", "f16"])
def test_cogvideox_2b_width_c1_geometry_vs_oracle(s2v, dt_name):
    """BASELINE.json configs[0] geometry (CogVideoX-2B width D = 1920, non-RoPE, 9 frames 256x256 -> latents 3x32x32,
    N = 226 + 256 + 768 = 1250 tokens), 2 of the 30 layers, one full denoise step (CFG + DDIM) against the CPU oracle.
    D = 1920 is not a multiple of 256: exercises the padded-tile GEMM paths."""
    dt = DT[dt_name]
    cfg = s2v.cogvideox_2b()
    cfg.num_layers = 2
    sd = s2v.weights.synthetic_state_dict(cfg, seed=11, parity=True)
    g = torch.Generator().manual_seed(12)
    F, H, W, T = 3, 32, 32, 226
    lat = torch.randn(1, F, 16, H, W, generator=g).to(dt)
    text = torch.randn(2, T, 4096, generator=g).to(dt)
    ref = (torch.randn(1, 1, 16, H, W, generator=g) * 0.7).to(dt)
    sch = s2v.CogVideoXDDIMScheduler(snr_shift_scale=3.0)
    sch.set_timesteps(10)
    t = sch.timesteps[2]
    ocfg = dict(num_heads=30, num_layers=2, use_rope=False, norm_eps=1e-5)
    with torch.no_grad():
        npred = tr.transformer_forward({k: v.to(dt) for k, v in sd.items()}, ocfg, torch.cat([lat] * 2), text, ref,
                                       torch.tensor([int(t), int(t)]))
        v = sched_ref.cfg_combine(npred, 6.0)
        exp, _ = sched_ref.ddim_step(sched_ref.alphas_cumprod(3.0), 10, v, int(t), lat)
        exp = exp.to(dt).float()
    m = s2v.HipCogVideoXTransformer3DModel(cfg, dt, DEV)
    m.load_state_dict(sd)
    eng = m.engine
    eng.set_geometry(2, T, F, H, W)
    eng.prepare_tables(256, 256)
    eng.set_conditioning(text, ref)
    x = lat.to(DEV).contiguous().clone()
    eng.denoise_step(x, float(t), sch.coef(t, dt, 6.0))
    torch.cuda.synchronize()
    npred_hip = eng.last_noise_pred()
    assert_close(npred_hip, npred, dt_name, "2B noise_pred")
    assert_close(x, exp, dt_name, "2B latents after one step")


@pytest.mark.parametrize("impl", [0, 4])
def test_op_attention_strongly_negative_and_positive_scores(s2v, impl):
    """scores far outside [-128, 128] in the exp2 domain: the first-tile maximum must be adopted without forming exp2(+-big)
    (0 * inf = NaN otherwise); softmax is shift invariant, so the result is the plain softmax-weighted mean of V."""
    L = s2v._lib
    B, H, N = 1, 2, 200
    D = H * 64
    g = torch.Generator().manual_seed(5)
    for sign in (-1.0, 1.0):
        q = torch.full((B * N, D), 4.0) + 0.05 * torch.randn(B * N, D, generator=g)
        k = sign * (torch.full((B * N, D), 4.0) + 0.05 * torch.randn(B * N, D, generator=g))
        v = torch.randn(B * N, D, generator=g)
        qkv = torch.cat([q, k, v], dim=1).bfloat16()
        pad = torch.zeros(64, 3 * D, dtype=torch.bfloat16)
        qkv_d = torch.cat([qkv, pad]).to(DEV)
        out = torch.full((B * N, D), float("nan"), dtype=torch.bfloat16, device=DEV)
        vt = torch.zeros(B * H * 64 * 256, dtype=torch.bfloat16, device=DEV)
        L.check(L.lib().s2v_op_attention(L.ptr(qkv_d), L.ptr(vt), L.ptr(out), B, H, N, 1, impl, L.stream_ptr()))
        torch.cuda.synchronize()
        assert torch.isfinite(out.float()).all()
        qf, kf, vf = (x.float().reshape(B, N, H, 64).transpose(1, 2) for x in (qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]))
        exp = torch.softmax(qf @ kf.transpose(-1, -2) * 0.125, dim=-1) @ vf
        exp = exp.transpose(1, 2).reshape(B * N, D)
        rel = ((out.float().cpu() - exp).norm() / exp.norm()).item()
        assert rel <= 3e-2, (sign, rel)


@pytest.mark.parametrize("impl", [0, 4])
def test_op_attention_late_score_jump_takes_the_slow_path(s2v, impl):
    """deferred maximum (attention_q4.hip / gen_attn_q4.py): a row keeps the maximum of its first KV tile until a later tile's row sum
    exceeds 2^64.  Forty keys far into the sequence are set to 12 x (query row 17): against that row (and its like) their scores jump by
    60-110 natural units (90-160 in the exp2 domain: one head's first exp2 pass overflows to inf, the other's stays finite) after O and
    l have accumulated eleven tiles at the old scale -- the slow path (true
    maximum, rescale of O / l, exp2 redone, next tile's scores shifted) must take over.  Reference: fp64 softmax on the SAME bf16
    rounding of q * scale * log2(e) the kernel (and the reference's bf16 math path) applies -- at scores of this size that rounding
    alone moves the weights by percent, it is not what this test is about."""
    L = s2v._lib
    B, H, N, k0, nk, r0 = 1, 2, 700, 437, 40, 17
    D = H * 64
    g = torch.Generator().manual_seed(9)
    qkv = torch.randn(B * N, 3 * D, generator=g).bfloat16()
    for h in range(H):
        qkv[k0:k0 + nk, D + h * 64:D + (h + 1) * 64] = (12.0 * qkv[r0, h * 64:(h + 1) * 64].float()).bfloat16()
    qd = torch.cat([qkv, torch.zeros(64, 3 * D, dtype=torch.bfloat16)]).to(DEV)
    out = torch.full((B * N, D), float("nan"), dtype=torch.bfloat16, device=DEV)
    vt = torch.zeros(B * H * 64 * ((N + 63) // 64 * 64), dtype=torch.bfloat16, device=DEV)
    # impl 4: the same jump through attn_q4h, where the first exp2 pass overflows fp16 (2^92 -> +inf in the conversion) and the packed sums carry the inf
    L.check(L.lib().s2v_op_attention(L.ptr(qd), L.ptr(vt), L.ptr(out), B, H, N, L.DTYPE_BF16, impl, L.stream_ptr()))
    torch.cuda.synchronize()
    got = out.float().cpu().double()
    assert torch.isfinite(got).all()
    c0 = 0.125 * 1.4426950408889634
    q, k, v = (qkv.float()[:, i * D:(i + 1) * D].reshape(N, H, 64).transpose(0, 1) for i in range(3))
    qs = (q * c0).bfloat16().double()
    s = qs @ k.double().transpose(-1, -2)                       # exp2-domain scores
    assert (s[:, r0, k0] - s[:, r0, :k0].max(dim=-1).values).min() > 70.0   # the jump really is beyond the 2^64 threshold
    p = torch.exp2(s - s.max(dim=-1, keepdim=True).values)
    ref = ((p @ v.double()) / p.sum(dim=-1, keepdim=True)).transpose(0, 1).reshape(N, D)
    err = (got - ref).abs().max().item()
    assert err <= 2e-2 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("dt_name", ["f32", "bf16"])
def test_lora_adaln_scope_intended_vs_oracle(s2v, dt_name):
    """lora_adaln_scope = "intended" (normalization.py:468-478 as its comments read: base weights for the video / text
    modulation, the LoRA only for the reference-image chunks) against the oracle's restatement of that reading; the shipped
    semantics (LoRA merged into norm{1,2}.linear) must give a visibly different answer on the same weights"""
    dt = torch.float32 if dt_name == "f32" else torch.bfloat16
    cfg = s2v.tiny(use_rope=True, heads=3, layers=2, text_dim=128, temb=64)
    cfg.max_text_seq_length = 7
    cfg.lora_adaln_scope = "intended"
    sd = s2v.weights.synthetic_state_dict(cfg, seed=5, parity=True)
    lora = s2v.weights.synthetic_lora(cfg, rank=8, seed=6, std=0.3)
    g = torch.Generator().manual_seed(17)
    B, F, C, H, W, T = 2, 3, 16, 16, 24, 7
    lat = torch.randn(B, F, C, H, W, generator=g).to(dt)
    text = torch.randn(B, T, 128, generator=g).to(dt)
    ref = (torch.randn(1, 1, C, H, W, generator=g) * 0.7).to(dt)
    ts = torch.tensor([500, 500])
    ocfg = dict(num_heads=3, num_layers=2, use_rope=True, norm_eps=1e-5)
    ref_rope, rope = tr.pipeline_rope(H * 8, W * 8, F)
    kw = dict(image_rotary_emb=tuple(x.to(DEV) for x in rope), ref_image_rotary_emb=tuple(x.to(DEV) for x in ref_rope))
    main, cond = tr.merge_lora_scoped(sd, lora, 0.5)
    with torch.no_grad():
        exp = tr.transformer_forward({k: v.to(dt) for k, v in main.items()}, ocfg, lat, text, ref, ts, rope, ref_rope,
                                     cond_sd={k: v.to(dt) for k, v in cond.items()})
        shipped = tr.transformer_forward({k: v.to(dt) for k, v in tr.merge_lora(sd, lora, 0.5).items()}, ocfg, lat, text, ref, ts,
                                         rope, ref_rope)
    assert rel_l2(shipped, exp) > 5e-2  # the two readings differ by far more than any tolerance below
    m = s2v.HipCogVideoXTransformer3DModel(cfg, dt, DEV)
    m.load_state_dict(sd, lora=lora, lora_scale=0.5)
    y = m(hidden_states=lat.to(DEV), encoder_hidden_states=text.to(DEV), ref_img_states=ref.to(DEV), timestep=ts.to(DEV),
          return_dict=False, eval=True, **kw)[0]
    torch.cuda.synchronize()
    assert_close(y, exp, dt_name, "intended LoRA scope")
    # the block seam carries the same semantics (its modulation rows are computed per call)
    h = torch.randn(1, F * (H // 2) * (W // 2), 192, generator=g).to(dt)
    e0, e1 = torch.randn(1, T, 192, generator=g).to(dt), torch.randn(1, (H // 2) * (W // 2), 192, generator=g).to(dt)
    temb = torch.randn(1, 64, generator=g).to(dt)
    with torch.no_grad():
        eb = tr.block_forward({k: v.to(dt) for k, v in main.items()}, "transformer_blocks.1.", 3, h, e0, e1, temb, rope, ref_rope,
                              cond_sd={k: v.to(dt) for k, v in cond.items()})
    got = m.transformer_blocks[1](hidden_states=h.to(DEV), encoder_hidden_states=e0.to(DEV), temb=temb.to(DEV),
                                  enc_hidden_states1=e1.to(DEV), embed_ref_img=True, ref_img_seq_start=T,
                                  ref_img_seq_end=T + e1.shape[1], position_delta=0, timestep=None, layer=1, **kw)
    torch.cuda.synchronize()
    for a_, b_, nm in zip(got, eb, ("video", "text", "ref")):
        assert_close(a_, b_, dt_name, "intended scope, block seam " + nm)


@pytest.mark.parametrize("dt_name", ["f32", "bf16"])
def test_rotary_tables_without_the_pair_structure_take_the_separate_pass(s2v, dt_name):
    """the fused QKV epilogue needs tables that repeat every value twice (what get_3d_rotary_pos_embed builds); any other table must
    still be honoured -- through qk_norm_rope_k.  Tables with independent values per dim, against the oracle on the same tables."""
    dt = torch.float32 if dt_name == "f32" else torch.bfloat16
    cfg = s2v.tiny(use_rope=True, heads=2, layers=2, text_dim=64, temb=64)
    sd = s2v.weights.synthetic_state_dict(cfg, seed=21, parity=True)
    g = torch.Generator().manual_seed(22)
    B, F, C, H, W, T = 2, 2, 16, 12, 20, 5
    lat = torch.randn(B, F, C, H, W, generator=g).to(dt)
    text = torch.randn(B, T, 64, generator=g).to(dt)
    ref = (torch.randn(1, 1, C, H, W, generator=g) * 0.7).to(dt)
    ts = torch.tensor([400, 400])
    R = (H // 2) * (W // 2)
    ang = torch.rand(R * (F + 1), 64, generator=g) * 6.28
    cos, sin = torch.cos(ang), torch.sin(ang)          # no two neighbouring dims share an angle
    ref_rope, rope = (cos[:R], sin[:R]), (cos[R:], sin[R:])
    ocfg = dict(num_heads=2, num_layers=2, use_rope=True, norm_eps=1e-5)
    with torch.no_grad():
        exp = tr.transformer_forward({k: v.to(dt) for k, v in sd.items()}, ocfg, lat, text, ref, ts, rope, ref_rope)
    m = s2v.HipCogVideoXTransformer3DModel(cfg, dt, DEV)
    m.load_state_dict(sd)
    kw = dict(image_rotary_emb=tuple(x.to(DEV) for x in rope), ref_image_rotary_emb=tuple(x.to(DEV) for x in ref_rope))
    y = m(hidden_states=lat.to(DEV), encoder_hidden_states=text.to(DEV), ref_img_states=ref.to(DEV), timestep=ts.to(DEV),
          return_dict=False, eval=True, **kw)[0]
    torch.cuda.synchronize()
    assert_close(y, exp, dt_name, "unpaired rotary tables")


@pytest.mark.parametrize("B,K,rows", [(2, 512, 6 * 3072 * 4 + 5), (1, 512, 8192), (4, 1024, 4099), (2, 64, 5000)])
def test_op_mod_gemv_register_form_equals_the_row_form_and_the_fp64_sum(s2v, B, K, rows):
    """the step's stacked AdaLN linears on silu(temb) (normalization.py CogVideoXLayerNormZero.linear, called from
    cogvideox_transformer_3d.py:122-186): the product dispatch (a lane keeps its slice of silu(temb) in registers and streams eight rows)
    must give the bits of the one-wave-per-row kernel, and both the fp64 sum on the bf16-rounded silu within bf16 rounding; the ragged
    row tail (rows not a multiple of 32) is written exactly once"""
    L = s2v._lib
    g = torch.Generator().manual_seed(B * 7 + K)
    emb = torch.randn(B, K, generator=g).bfloat16()
    W = (torch.randn(rows, K, generator=g) / K ** 0.5).bfloat16()
    bias = torch.randn(rows, generator=g).bfloat16()
    outs = []
    emb_d, W_d, bias_d = emb.to(DEV), W.to(DEV), bias.to(DEV)
    for impl in (0, 1):
        out = torch.full((B, rows + 8), 7.0, dtype=torch.bfloat16, device=DEV)
        L.check(L.lib().s2v_op_mod_gemv(L.ptr(emb_d), L.ptr(W_d), L.ptr(bias_d), L.ptr(out), B, K, rows,
                                        L.DTYPE_BF16, impl, L.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1])
    flat = outs[0].flatten()
    assert (flat[B * rows:] == 7.0).all(), "nothing past [B, rows] may be written"
    got = flat[:B * rows].reshape(B, rows).double()
    x = torch.nn.functional.silu(emb.float()).bfloat16().double()
    ref = x @ W.double().T + bias.double()
    assert (got - ref).abs().max().item() <= 2 ** -7 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("mode", ["fused_graph", "seams"])
def test_pipeline_callback_overrides_are_honoured(s2v, mode):
    """custom_cogvideox_pipe.py:298-305: what `callback_on_step_end` returns replaces `latents` and `prompt_embeds` (the concatenated
    [negative | positive] pair at that point of the loop, :196) for the following steps.  The reference's own branch cannot be put in
    a fixture (`locals()` inside a comprehension: KeyError under Python < 3.12), so the expectation is the oracle loop with the same
    two overrides applied after step 0; fused mode must keep its one latent buffer (captured graph) and re-project the new text."""
    from oracle import sched_ref

    g = load_golden("pipeline_tiny.npz")
    pipe = _pipe_from_golden(s2v, g, "ddim")
    seen = []

    def on_step(p_, i, tt, kw):
        seen.append(sorted(kw))
        if i == 0:
            return {"latents": kw["latents"] * 0.5 + 0.1, "prompt_embeds": kw["prompt_embeds"].flip(0)}
        return {}

    out = pipe(output_type="latent", fused=mode != "seams", use_graph=mode == "fused_graph", callback_on_step_end=on_step,
               callback_on_step_end_tensor_inputs=["latents", "prompt_embeds"], **_pipe_args(g))[0]
    torch.cuda.synchronize()
    assert seen == [["latents", "prompt_embeds"]] * 3
    sd = weights_of(g)
    cfg = dict(num_heads=2, num_layers=2, use_rope=True, norm_eps=1e-5)  # s2v.tiny(), as _pipe_from_golden builds it
    pe, ne, ref, lat = t(g["prompt_embeds"]), t(g["negative_prompt_embeds"]), t(g["ref"]), t(g["latents0"])
    text = torch.cat([ne, pe], dim=0)
    ref_rope, rope = tr.pipeline_rope(480, 720, lat.shape[1])
    ac = sched_ref.alphas_cumprod(1.0)
    with torch.no_grad():
        for i, tt in enumerate(sched_ref.trailing_timesteps(3)):
            npred = tr.transformer_forward(sd, cfg, torch.cat([lat] * 2), text, ref, torch.tensor([tt, tt]), rope, ref_rope)
            lat, _ = sched_ref.ddim_step(ac, 3, sched_ref.cfg_combine(npred, 6.0), int(tt), lat)
            lat = lat.float()
            if i == 0:
                lat, text = lat * 0.5 + 0.1, text.flip(0)
    err = (out.float().cpu() - lat).abs().max().item()
    assert err <= 1e-3, err
    assert (lat - t(g["final_ddim"])).abs().max().item() > 1e-2, "the overrides must change the result"
    with pytest.raises(ValueError, match="callback_on_step_end_tensor_inputs"):
        pipe(output_type="latent", callback_on_step_end=on_step, callback_on_step_end_tensor_inputs=["noise_pred"], **_pipe_args(g))


@pytest.mark.parametrize("spread", [1.0, 4.0, 8.0])
def test_op_attention_fp16_p_moderate_jumps_and_tails_vs_fp64(s2v, spread):
    """attn_q4h (attn_p_format 1) where its fp16 range is exercised without overflowing to inf: score spread 1 / 4 / 8 natural units (at 4 and 8
    later tiles exceed the adopted maximum by more than 9.7 -- the 2^14 threshold -- so rows re-adopt their maximum repeatedly, and most keys sit
    2^-14 .. 2^-24 below it, in fp16's subnormal range), 5000 keys.  Reference: fp64 softmax on the bf16-rounded q * scale * log2 e the kernels use."""
    L = s2v._lib
    B, H, N = 1, 2, 5000
    D = H * 64
    g = torch.Generator().manual_seed(int(spread * 10))
    qkv = torch.randn(B * N, 3 * D, generator=g)
    qkv[:, :D] *= spread
    qkv = qkv.bfloat16()
    qd = torch.cat([qkv, torch.zeros(64, 3 * D, dtype=torch.bfloat16)]).to(DEV)
    vt = torch.zeros(B * H * 64 * ((N + 63) // 64 * 64), dtype=torch.bfloat16, device=DEV)
    outs = {}
    for impl in (0, 4):
        out = torch.full((B * N, D), float("nan"), dtype=torch.bfloat16, device=DEV)
        L.check(L.lib().s2v_op_attention(L.ptr(qd), L.ptr(vt), L.ptr(out), B, H, N, L.DTYPE_BF16, impl, L.stream_ptr()))
        torch.cuda.synchronize()
        outs[impl] = out.float().cpu().double()
        assert torch.isfinite(outs[impl]).all()
    c0 = 0.125 * 1.4426950408889634
    q, k, v = (qkv.float()[:, i * D:(i + 1) * D].reshape(N, H, 64).transpose(0, 1) for i in range(3))
    s_ = (q * c0).bfloat16().double() @ k.double().transpose(-1, -2)
    p = torch.exp2(s_ - s_.max(dim=-1, keepdim=True).values)
    ref = ((p @ v.double()) / p.sum(dim=-1, keepdim=True)).transpose(0, 1).reshape(N, D)
    e0, e4 = (outs[0] - ref).abs().max().item(), (outs[4] - ref).abs().max().item()
    tol = 2e-2 * max(1.0, ref.abs().max().item())
    assert e0 <= tol and e4 <= tol, (e0, e4)
    r0, r4 = ((outs[0] - ref).norm() / ref.norm()).item(), ((outs[4] - ref).norm() / ref.norm()).item()
    assert r4 <= max(1.5 * r0, 5e-3), (r0, r4)   # fp16 P (11 significant bits) is not less accurate than bf16 P (8)


def test_fp16_engine_long_sequence_runs_the_four_wave_kernel_and_matches_fp32(s2v):
    """fp16 model dtype at 5 127 tokens (> 4 608: launch_attn_f16 routes to attn_q4hh, the four-wave asm kernel with fp16 q / k / V^T / P, persistent
    launch inside the engine) against the fp32 engine on the same fp16-representable weights: one forward, and graph replay == eager over three steps"""
    cfg = s2v.tiny(use_rope=True, heads=4, layers=1, text_dim=128, temb=64)
    cfg.max_text_seq_length = 7
    g = torch.Generator().manual_seed(41)
    F, H, W = 4, 64, 64
    sd = {k: v.half().float() for k, v in s2v.weights.synthetic_state_dict(cfg, seed=6, parity=True).items()}
    lat0 = torch.randn(1, F, 16, H, W, generator=g).half()
    text = torch.randn(2, 7, 128, generator=g).half()
    ref = (torch.randn(1, 1, 16, H, W, generator=g) * 0.7).half()
    outs = {}
    for dt in (torch.float16, torch.float32):
        m = s2v.HipCogVideoXTransformer3DModel(cfg, dt, DEV)
        m.load_state_dict(sd)
        eng = m.engine
        eng.set_geometry(2, 7, F, H, W)
        eng.prepare_tables(H * 8, W * 8)
        eng.set_conditioning(text.to(dt), ref.to(dt))
        outs[dt] = eng.forward(lat0.to(dt), torch.tensor([500.0, 500.0]), shared_latent=True).float().cpu()
        if dt == torch.float16:
            sch = s2v.CogVideoXDDIMScheduler(snr_shift_scale=1.0)
            sch.set_timesteps(50)
            a, b = lat0.to(DEV).clone(), lat0.to(DEV).clone()
            for x, graph in ((a, True), (b, False)):
                for i in range(3):
                    t_ = sch.timesteps[i]
                    eng.denoise_step(x, float(t_), sch.coef(t_, dt, 6.0), use_graph=graph)
            torch.cuda.synchronize()
            assert torch.isfinite(a.float()).all() and torch.equal(a, b)
            slow, total = eng.attn_slow_stats()
            assert total > 0   # the four-wave kernel ran (the census is its own)
    assert_close(outs[torch.float16], outs[torch.float32], "f16", "fp16 engine at 5127 tokens vs fp32 engine")


def test_fp16_engine_row_tail_of_a_split_projection(s2v):
    """fp16 model dtype at a geometry whose QKV projection is split into full 256-row tiles + a row tail on the side stream (43 x 6 tiles would
    spill into a second round of 256 CUs, 42 x 6 do not: api.hip linear()).  The tail launch (gemm_bf16_128 at m_begin > 0) must run the fp16
    instantiation -- until the last day of round 5 that branch launched the bf16 kernel on fp16 operands, which no test geometry reached.
    Product build (q/k-norm fused into the projection's epilogue, also for fp16) and diagnostics build with the fusion off (plain-bias tail +
    qk_norm_rope_k) against the fp32 engine, and bit-identical to each other."""
    cfg = s2v.tiny(use_rope=True, heads=8, layers=1, text_dim=128, temb=64)
    cfg.max_text_seq_length = 7
    g = torch.Generator().manual_seed(43)
    F, H, W = 4, 64, 68   # 7 + 5 * 32 * 34 = 5447 tokens, CFG pair: 10 894 rows = 42 tiles + 142 rows
    sd = {k: v.half().float() for k, v in s2v.weights.synthetic_state_dict(cfg, seed=8, parity=True).items()}
    lat0 = torch.randn(1, F, 16, H, W, generator=g).half()
    text = torch.randn(2, 7, 128, generator=g).half()
    ref = (torch.randn(1, 1, 16, H, W, generator=g) * 0.7).half()
    L = s2v._lib
    diag = L.diag_lib()
    prev = L._lib
    L.lib()

    def fwd(dt):
        m = s2v.HipCogVideoXTransformer3DModel(cfg, dt, DEV)
        m.load_state_dict(sd)
        eng = m.engine
        eng.set_geometry(2, 7, F, H, W)
        eng.prepare_tables(H * 8, W * 8)
        eng.set_conditioning(text.to(dt), ref.to(dt))
        out = eng.forward(lat0.to(dt), torch.tensor([500.0, 500.0]), shared_latent=True).float().cpu()
        torch.cuda.synchronize()
        return out

    f32 = fwd(torch.float32)
    fused = fwd(torch.float16)
    try:
        L._lib = diag
        diag.s2v_set_fused_qk(0)
        unfused = fwd(torch.float16)
    finally:
        diag.s2v_set_fused_qk(1)
        L._lib = prev
    assert torch.isfinite(fused).all() and torch.isfinite(unfused).all()
    assert_close(fused, f32, "f16", "fp16 engine (fused q/k-norm, split QKV) vs fp32 engine")
    assert torch.equal(fused, unfused), (fused - unfused).abs().max().item()


def test_attn_q4h_saturates_v_beyond_the_fp16_range(s2v):
    """|V| > 65504 (finite in bf16) must not become an fp16 infinity in V^T -- 0 * inf in P.V would turn the head-dim column of EVERY
    query into NaN.  The fp16 V^T pass saturates; the result stays finite, and equals the bf16-P kernel's wherever the huge key carries no weight."""
    L = s2v._lib
    B, H, N = 1, 2, 300
    D = H * 64
    g = torch.Generator().manual_seed(9)
    qkv = torch.randn(B * N, 3 * D, generator=g)
    qkv[7, 2 * D + 5] = 1.0e6          # one V element far beyond 65504 ...
    qkv[7, D:D + 64] = -50.0 * qkv[0:1, 0:64].sign()   # ... on a key that query 0 all but ignores
    qkv = qkv.bfloat16()
    qd = torch.cat([qkv, torch.zeros(64, 3 * D, dtype=torch.bfloat16)]).to(DEV)
    outs = {}
    for impl in (0, 4):
        out = torch.full((B * N, D), float("nan"), dtype=torch.bfloat16, device=DEV)
        vt = torch.zeros(B * H * 64 * 320, dtype=torch.bfloat16, device=DEV)
        L.check(L.lib().s2v_op_attention(L.ptr(qd), L.ptr(vt), L.ptr(out), B, H, N, L.DTYPE_BF16, impl, L.stream_ptr()))
        torch.cuda.synchronize()
        outs[impl] = out.float().cpu()
    assert torch.isfinite(outs[4]).all() and torch.isfinite(outs[0]).all()
    # columns other than the poisoned one are untouched by the saturation
    mask = torch.ones(D, dtype=torch.bool)
    mask[5] = False
    assert (outs[4][:, mask] - outs[0][:, mask]).abs().max() <= 2e-2 * max(1.0, outs[0][:, mask].abs().max().item())


def test_attn_p_format_auto_settles_on_the_census_of_the_first_step(s2v):
    """attn_p_format = "auto" (opt-in; the default is "bf16"): an engine starts with fp16 P where the four-wave attention kernel runs (> 4608 tokens), its first
    denoise step runs eagerly and reads the kernel's slow-path census (s2v_attn_slow_stats); smooth scores keep fp16, spiky ones (q / k LayerNorm
    weights x 12: score jumps far beyond the 2^14 threshold in most tiles) switch the engine to bf16 P for good.  Either way graph replay == eager
    afterwards, and the result stays within the bf16 tolerance of a bf16-P engine on the same weights."""
    import copy

    cfg = s2v.tiny(use_rope=True, heads=4, layers=1, text_dim=128, temb=64)
    cfg.max_text_seq_length = 7
    g = torch.Generator().manual_seed(31)
    F, H, W = 4, 64, 64                                  # 7 + 5 * 1024 = 5127 tokens: attn_q4 / attn_q4h
    lat0 = torch.randn(1, F, 16, H, W, generator=g).bfloat16().to(DEV)
    text = torch.randn(2, 7, 128, generator=g).bfloat16()
    ref = (torch.randn(1, 1, 16, H, W, generator=g) * 0.7).bfloat16()
    sch = s2v.CogVideoXDDIMScheduler(snr_shift_scale=1.0)
    sch.set_timesteps(50)
    for spiky in (False, True):
        sd = s2v.weights.synthetic_state_dict(cfg, seed=5, parity=True)
        if spiky:
            for k in list(sd):
                if k.endswith("norm_q.weight") or k.endswith("norm_k.weight"):
                    sd[k] = sd[k] * 12.0
        outs = {}
        for fmt in ("auto", "bf16"):
            c = copy.copy(cfg)
            c.attn_p_format = fmt
            m = s2v.HipCogVideoXTransformer3DModel(c, torch.bfloat16, DEV)
            m.load_state_dict(sd)
            eng = m.engine
            eng.set_geometry(2, 7, F, H, W)
            eng.prepare_tables(H * 8, W * 8)
            eng.set_conditioning(text, ref)
            a, b = lat0.clone(), lat0.clone()
            for x, graph in ((a, True), (b, False)):   # `a` settles the format in its first step; `b` repeats the three steps eagerly
                for i in range(3):
                    t = sch.timesteps[i]
                    eng.denoise_step(x, float(t), sch.coef(t, torch.bfloat16, 6.0), use_graph=graph)
            torch.cuda.synchronize()
            assert torch.isfinite(a.float()).all()
            if fmt == "auto":
                assert eng.attn_slow_fraction is not None
                assert eng.attn_p_format == ("bf16" if spiky else "f16"), (spiky, eng.attn_slow_fraction)
                assert (eng.attn_slow_fraction > eng.AUTO_SLOW_FRACTION) == spiky, eng.attn_slow_fraction
                if not spiky:
                    assert torch.equal(a, b)   # fp16 throughout: graph replay == eager
            else:
                assert eng.attn_p_format == "bf16" and torch.equal(a, b)
            outs[fmt] = a.float().cpu()
            slow, total = eng.attn_slow_stats()
            assert total > 0
        rel = ((outs["auto"] - outs["bf16"]).norm() / outs["bf16"].norm()).item()
        assert rel <= 2e-2, (spiky, rel)
