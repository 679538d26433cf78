"""End-to-end GPU run of the caller (`video_generate.inference`, src/video_generate.py:7-66) on tiny fp32 modules: reference
image -> VAE encode -> T5 prompt embeddings -> 3 DDIM steps with CFG -> VAE decode -> frames, against the composition of
the CPU oracles fed with the same random draws."""
import numpy as np
import pytest
import torch

from oracle import sched_ref, t5_ref, transformer_ref as tr, vae_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
VAE = dict(block_out_channels=(16, 16, 32, 32), layers_per_block=1, norm_num_groups=4, latent_channels=16,
           sample_height=96, sample_width=160, scaling_factor=0.7, temporal_compression_ratio=4)
T5 = dict(vocab_size=100, d_model=64, d_kv=64, num_heads=2, d_ff=128, num_layers=2, relative_attention_num_buckets=32,
          relative_attention_max_distance=128, layer_norm_epsilon=1e-6)


def test_image_and_ids_to_frames_fp32_vs_oracles(s2v):
    dt = torch.float32
    H, W, F, Tn, steps, gs = 96, 160, 5, 7, 3, 6.0
    cfg = s2v.tiny(use_rope=True, heads=2, layers=2, text_dim=64, temb=64)
    cfg.max_text_seq_length = Tn
    cfg.vae_scaling_factor = 0.7
    sd_tr = s2v.weights.synthetic_state_dict(cfg, seed=71, parity=True)
    vcfg = s2v.VAEConfig(**VAE)
    sd_vae = dict(s2v.weights.synthetic_vae_state_dict(vcfg, seed=72))
    sd_vae.update(s2v.weights.synthetic_vae_encoder_state_dict(vcfg, seed=73))
    tcfg = s2v.T5Config(**T5)
    sd_t5 = s2v.weights.synthetic_t5_state_dict(tcfg, seed=74, gain=0.6)
    g = torch.Generator().manual_seed(75)
    image = torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8).numpy()
    ids = torch.randint(1, 100, (1, Tn), generator=g)
    neg = torch.zeros((1, Tn), dtype=torch.long)

    model = s2v.HipCogVideoXTransformer3DModel(cfg, dt, DEV)
    model.load_state_dict(sd_tr)
    vae = s2v.HipAutoencoderKLCogVideoX(vcfg, dt, DEV)
    vae.load_state_dict(sd_vae)
    t5 = s2v.HipT5EncoderModel(tcfg, dt, DEV)
    t5.load_state_dict(sd_t5)
    pipe = s2v.S2VPipeline(model, s2v.CogVideoXDDIMScheduler(snr_shift_scale=1.0), vae)
    frames = s2v.video_generate.inference(pipe, t5, image, ids, neg, height=H, width=W, num_frames=F,
                                          num_inference_steps=steps, guidance_scale=gs, seed=1234)
    assert frames.dtype == np.float32 and frames.shape[1:] == (H, W, 3)

    # ---- the same run on the CPU oracles; the generator on cuda:0 is replayed to obtain the two draws
    gen = torch.Generator(device=DEV).manual_seed(1234)
    noise = torch.randn((1, 16, 1, H // 8, W // 8), generator=gen, device=DEV, dtype=dt).cpu()
    lat0 = torch.randn((1, (F - 1) // 4 + 1, 16, H // 8, W // 8), generator=gen, device=DEV, dtype=dt).cpu()
    with torch.no_grad():
        x = (torch.from_numpy(image)[None].float() / 255.0 * 2.0 - 1.0).permute(0, 3, 1, 2).unsqueeze(0).permute(0, 2, 1, 3, 4)
        ref = vae_ref.encode_image(sd_vae, VAE, x, noise, False)
        pe = t5_ref.encoder_forward(sd_t5, T5, ids)
        ne = t5_ref.encoder_forward(sd_t5, T5, neg)
        text = torch.cat([ne, pe], dim=0)
        ac = sched_ref.alphas_cumprod(1.0)
        ts = sched_ref.trailing_timesteps(steps)
        ref_rope, rope = tr.pipeline_rope(H, W, lat0.shape[1])
        ocfg = dict(num_heads=2, num_layers=2, use_rope=True, norm_eps=1e-5)
        lat = lat0.clone()
        for t in ts:
            tt = torch.tensor([int(t), int(t)])
            npred = tr.transformer_forward(sd_tr, ocfg, torch.cat([lat, lat]), text, ref, tt, rope, ref_rope).float()
            lat = sched_ref.ddim_step(ac, steps, sched_ref.cfg_combine(npred, gs), int(t), lat)[0]
        video = vae_ref.decode_latents(sd_vae, VAE, lat, False)
        exp = vae_ref.postprocess_np(video)[0]
    assert frames.shape == exp.shape  # 2 latent frames decode to 8 (even branch of the temporal upsampling), as in the reference
    assert np.abs(frames - exp).max() <= 2e-3, np.abs(frames - exp).max()


def test_seeded_draws_equal_the_reference_global_generator_order(s2v):
    """The reference seeds the GLOBAL device generator (seed_everything, src/inference.py:28-35) and draws twice from it: the posterior sample
    of the reference image (src/video_generate.py:37), then the initial latents (pipeline_cogvideox.py:320-344).  inference(seed=S) must hand
    the pipeline exactly those two tensors: torch.cuda.manual_seed_all(S); randn(posterior shape); randn(latent shape)."""
    dt = torch.float32
    H, W, F, Tn = 96, 160, 5, 7
    cfg = s2v.tiny(use_rope=True, heads=2, layers=1, text_dim=64, temb=64)
    cfg.max_text_seq_length = Tn
    cfg.vae_scaling_factor = 0.7
    vcfg = s2v.VAEConfig(**VAE)
    sd_vae = dict(s2v.weights.synthetic_vae_state_dict(vcfg, seed=72))
    sd_vae.update(s2v.weights.synthetic_vae_encoder_state_dict(vcfg, seed=73))
    tcfg = s2v.T5Config(**T5)
    model = s2v.HipCogVideoXTransformer3DModel(cfg, dt, DEV)
    model.load_state_dict(s2v.weights.synthetic_state_dict(cfg, seed=71, parity=True))
    vae = s2v.HipAutoencoderKLCogVideoX(vcfg, dt, DEV)
    vae.load_state_dict(sd_vae)
    t5 = s2v.HipT5EncoderModel(tcfg, dt, DEV)
    t5.load_state_dict(s2v.weights.synthetic_t5_state_dict(tcfg, seed=74, gain=0.6))
    pipe = s2v.S2VPipeline(model, s2v.CogVideoXDDIMScheduler(snr_shift_scale=1.0), vae)
    g = torch.Generator().manual_seed(75)
    image = torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8).numpy()
    ids = torch.randint(1, 100, (1, Tn), generator=g)
    neg = torch.zeros((1, Tn), dtype=torch.long)
    seen = {}
    prep, call = pipe.prepare_latents, pipe.__class__.__call__

    def spy_prepare(*a, **k):
        seen["latents0"] = prep(*a, **k).clone()
        return seen["latents0"].clone()

    pipe.prepare_latents = spy_prepare
    orig_call = s2v.S2VPipeline.__call__

    def spy_call(self, *a, **k):
        seen["ref"] = k["ref_img_states"].clone()
        return orig_call(self, *a, **k)

    s2v.S2VPipeline.__call__ = spy_call
    try:
        s2v.video_generate.inference(pipe, t5, image, ids, neg, height=H, width=W, num_frames=F, num_inference_steps=1, guidance_scale=6.0,
                                     seed=4321, output_type="latent")
    finally:
        s2v.S2VPipeline.__call__ = orig_call
    # the reference's order on the GLOBAL generator of the device
    torch.manual_seed(4321)
    torch.cuda.manual_seed_all(4321)
    noise = torch.randn((1, 16, 1, H // 8, W // 8), device=DEV, dtype=dt)
    lat0 = torch.randn((1, (F - 1) // 4 + 1, 16, H // 8, W // 8), device=DEV, dtype=dt)
    assert torch.equal(seen["latents0"], lat0)   # init_noise_sigma = 1
    # and the posterior sample was made from the FIRST draw: mean + std * noise, times the scaling factor
    x = (torch.from_numpy(image)[None].float() / 255.0 * 2.0 - 1.0).permute(0, 3, 1, 2).unsqueeze(0).permute(0, 2, 1, 3, 4)
    with torch.no_grad():
        exp_ref = vae_ref.encode_image({k: v for k, v in sd_vae.items()}, VAE, x, noise.cpu(), False)
    assert (seen["ref"].float().cpu() - exp_ref).abs().max().item() <= 1e-3
