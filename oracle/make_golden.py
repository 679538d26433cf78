"""Generates tests/golden/*.npz by IMPORTING the reference (read-only, by path) in the build container.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

Never runs on the GPU box (the reference does not exist there); only its outputs -- inputs and expected outputs,
weights of the tiny seeded modules included -- are committed.  Nothing from the reference is copied.
"""
import functools
import os
import sys
import tempfile

import numpy as np
import torch

REF = os.environ.get("S2V_REFERENCE", "/root/reference")
OUT = os.environ.get("S2V_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    stub = tempfile.mkdtemp()
    open(os.path.join(stub, "imageio.py"), "w").close()  # the fork imports imageio at module top (export_utils.py:11)
    sys.path[:0] = [stub, os.path.join(REF, "diffusers", "src"), os.path.join(REF, "src")]
    import transformers.utils as tu

    tu.FLAX_WEIGHTS_NAME = getattr(tu, "FLAX_WEIGHTS_NAME", "flax_model.msgpack")
    import diffusers  # noqa: F401


def npsd(sd, prefix="w:"):
    return {prefix + k: v.detach().float().numpy() for k, v in sd.items()}


def randomize(module, gen, std=0.15):
    """larger-variance weights, non-zero biases and LN affines so every term contributes."""
    with torch.no_grad():
        for n, p in module.named_parameters():
            if p.ndim >= 2:
                p.copy_(torch.randn(p.shape, generator=gen) * (std if "linear" not in n else 0.05))
            elif "norm" in n and n.endswith("weight"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=gen))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=gen))


def bf16_np(t):
    """bf16 tensor -> uint16 bit pattern (numpy has no bf16)."""
    return t.detach().contiguous().view(torch.int16).numpy().view(np.uint16)


# ----------------------------------------------------------------------------------------------------------------
def gen_tables():
    from diffusers.models.embeddings import get_3d_rotary_pos_embed, get_3d_sincos_pos_embed, get_timestep_embedding
    from diffusers.pipelines.cogvideo.pipeline_cogvideox import get_resize_crop_region_for_grid
    from diffusers import CogVideoXDDIMScheduler

    out = {}
    t = torch.tensor([999.0, 979.0, 500.0, 19.0, 0.0])
    out["ts_t"] = t.numpy()
    for dim in (128, 1920, 3072):
        out[f"ts_{dim}"] = get_timestep_embedding(t, dim, flip_sin_to_cos=True, downscale_freq_shift=0).numpy()
    for (h, w, frames) in ((256, 256, 3), (480, 720, 13), (720, 1280, 13)):
        gh, gw = h // 16, w // 16
        crops = get_resize_crop_region_for_grid((gh, gw), 720 // 16, 480 // 16)
        cos, sin = get_3d_rotary_pos_embed(64, crops, (gh, gw), frames + 1)
        out[f"rope_crops_{h}x{w}"] = np.array(crops, dtype=np.int64)
        if cos.shape[0] <= 4096:
            out[f"rope_cos_{h}x{w}"] = cos.numpy()
            out[f"rope_sin_{h}x{w}"] = sin.numpy()
        else:  # keep the fixture small: every 61st row + fp64 checksums of the whole table
            out[f"rope_cos_{h}x{w}_rows61"] = cos[::61].numpy()
            out[f"rope_sin_{h}x{w}_rows61"] = sin[::61].numpy()
            out[f"rope_sum_{h}x{w}"] = np.array([cos.double().sum().item(), sin.double().sum().item(),
                                                  (cos.double() * torch.arange(cos.shape[0])[:, None]).sum().item(),
                                                  (sin.double() * torch.arange(sin.shape[0])[:, None]).sum().item()])
    for (D, wp, hp, fr) in ((128, 4, 4, 2), (1920, 16, 16, 3), (192, 6, 4, 3)):
        pe = get_3d_sincos_pos_embed(D, (wp, hp), fr, 1.875, 1.0)
        pe = torch.from_numpy(pe).flatten(0, 1).float()
        if D == 1920:
            out[f"sincos_{D}_{wp}x{hp}x{fr}_rows7"] = pe[::7].numpy()
            out[f"sincos_sum_{D}_{wp}x{hp}x{fr}"] = np.array([pe.double().sum().item(), pe.double().abs().sum().item()])
        else:
            out[f"sincos_{D}_{wp}x{hp}x{fr}"] = pe.numpy()
    for snr in (1.0, 3.0):
        s = CogVideoXDDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                                   set_alpha_to_one=True, prediction_type="v_prediction", timestep_spacing="trailing",
                                   rescale_betas_zero_snr=True, snr_shift_scale=snr)
        out[f"alphas_{snr}"] = s.alphas_cumprod.numpy()
        for n in (10, 50, 3):
            s.set_timesteps(n)
            out[f"timesteps_{n}"] = s.timesteps.numpy()
    np.savez_compressed(os.path.join(OUT, "tables.npz"), **out)


def sched_kwargs(snr):
    return dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                set_alpha_to_one=True, prediction_type="v_prediction", timestep_spacing="trailing",
                rescale_betas_zero_snr=True, snr_shift_scale=snr)


def gen_sched():
    from diffusers import CogVideoXDDIMScheduler, CogVideoXDPMScheduler

    gen = torch.Generator().manual_seed(7)
    shape = (1, 3, 16, 6, 10)
    # fp16 (round 5: the model dtype src/inference.py:191,209 selects for every non-5B checkpoint) comes LAST: the files of the other two
    # dtypes keep their bits (one generator feeds all three in turn)
    for dt_name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16), ("f16", torch.float16)):
        for kind, cls in (("ddim", CogVideoXDDIMScheduler), ("dpm", CogVideoXDPMScheduler)):
            for n_steps, snr in ((10, 3.0), (50, 1.0)):
                s = cls(**sched_kwargs(snr))
                s.set_timesteps(n_steps)
                ts = s.timesteps
                out = {"timesteps": ts.numpy(), "snr": np.array(snr), "n_steps": np.array(n_steps)}
                latents = torch.randn(shape, generator=gen).to(dt)
                out["latents0"] = latents.float().numpy()
                old = None
                steps = list(range(len(ts))) if n_steps == 10 else [0, 1, 2, 25, 48, 49]
                out["step_ids"] = np.array(steps)
                for i in range(len(ts)):
                    t = ts[i]
                    noise_pred = torch.randn((2,) + shape[1:], generator=gen).to(dt)  # transformer output, model dtype
                    npf = noise_pred.float()
                    u, c = npf.chunk(2)
                    g = 6.0
                    v = u + g * (c - u)
                    lat_in = latents
                    if kind == "ddim":
                        latents, x0 = s.step(v, t, latents, return_dict=False)
                        n1 = n2 = None
                    else:
                        # capture the randn draws of randn_tensor (scheduling_dpm_cogvideox.py:423,431)
                        g2 = torch.Generator().manual_seed(1000 + i)
                        st = g2.get_state()
                        latents, x0 = s.step(v, old, t, ts[i - 1] if i > 0 else None, latents, generator=g2,
                                             return_dict=False)
                        g3 = torch.Generator()
                        g3.set_state(st)
                        n1 = torch.randn(shape, generator=g3, dtype=dt)
                        n2 = torch.randn(shape, generator=g3, dtype=dt)
                        old = x0
                    latents = latents.to(dt)
                    if i in steps:
                        out[f"noise_pred_{i}"] = noise_pred.float().numpy()
                        out[f"lat_in_{i}"] = lat_in.float().numpy()
                        out[f"lat_out_{i}"] = latents.float().numpy()
                        out[f"x0_{i}"] = x0.float().numpy()
                        if n1 is not None:
                            out[f"n1_{i}"] = n1.float().numpy()
                            out[f"n2_{i}"] = n2.float().numpy()
                np.savez_compressed(os.path.join(OUT, f"sched_{kind}_{dt_name}_{n_steps}.npz"), **out)


TINY = dict(num_attention_heads=2, attention_head_dim=64, in_channels=16, out_channels=16, time_embed_dim=64,
            text_embed_dim=64, num_layers=2, sample_width=8, sample_height=8, sample_frames=5, max_text_seq_length=5)


def gen_transformer():
    from diffusers import CogVideoXTransformer3DModel
    from diffusers.models.embeddings import get_3d_rotary_pos_embed

    gen = torch.Generator().manual_seed(11)
    B, Fr, C, H, W, T = 2, 2, 16, 8, 8, 5
    lat = torch.randn(B, Fr, C, H, W, generator=gen)
    ref = torch.randn(1, 1, C, H, W, generator=gen) * 0.7
    text = torch.randn(B, T, 64, generator=gen)
    tstep = torch.tensor([979, 979])
    n = (H // 2) * (W // 2)
    cos, sin = get_3d_rotary_pos_embed(64, ((0, 0), (H // 2, W // 2)), (H // 2, W // 2), Fr + 1)
    ref_rope = (cos[:n], sin[:n])
    vid_rope = (cos[n:], sin[n:])
    weights = None
    for variant, use_rope in (("rope", True), ("sincos", False)):
        m = CogVideoXTransformer3DModel(use_rotary_positional_embeddings=use_rope, **TINY).float().eval()
        if weights is None:
            randomize(m, torch.Generator().manual_seed(5))
            weights = {k: v.clone() for k, v in m.state_dict().items() if "pos_embedding" not in k}
        else:
            m.load_state_dict(weights, strict=False)
        out = {"lat": lat.numpy(), "ref": ref.numpy(), "text": text.numpy(), "timestep": tstep.numpy()}
        if variant == "rope":
            out.update(npsd(weights))
            out.update(rope_cos=cos.numpy(), rope_sin=sin.numpy())
        for dt_name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16), ("f16", torch.float16)):
            mm = m.to(dt)
            kw = dict(image_rotary_emb=tuple(x for x in vid_rope), ref_image_rotary_emb=tuple(x for x in ref_rope)) \
                if use_rope else dict(image_rotary_emb=None, ref_image_rotary_emb=None)
            with torch.no_grad():
                y = mm(hidden_states=lat.to(dt), encoder_hidden_states=text.to(dt), ref_img_states=ref.to(dt),
                       timestep=tstep, return_dict=False, eval=True, **kw)[0]
                out[f"out_{dt_name}"] = y.float().numpy()
                if use_rope:
                    # block- and processor-level seams (same weights): CogVideoXBlock.forward / AttnProcessor.__call__
                    D = 128
                    g2 = torch.Generator().manual_seed(21)
                    h = torch.randn(B, Fr * n, D, generator=g2).to(dt)
                    e0 = torch.randn(B, T, D, generator=g2).to(dt)
                    e1 = torch.randn(B, n, D, generator=g2).to(dt)
                    temb = torch.randn(B, 64, generator=g2).to(dt)
                    blk = mm.transformer_blocks[1]
                    oh, oe0, oe1 = blk(hidden_states=h, encoder_hidden_states=e0, temb=temb, enc_hidden_states1=e1,
                                       image_rotary_emb=vid_rope, embed_ref_img=True, ref_img_seq_start=T,
                                       ref_img_seq_end=T + n, position_delta=0, ref_image_rotary_emb=ref_rope)
                    ah, ae = blk.attn1(hidden_states=h, encoder_hidden_states=torch.cat([e0, e1], dim=1),
                                       image_rotary_emb=vid_rope, embed_ref_img=True, ref_img_seq_start=T,
                                       ref_img_seq_end=T + n, position_delta=0, ref_image_rotary_emb=ref_rope)
                    if dt_name == "f32":
                        out.update(blk_h=h.float().numpy(), blk_e0=e0.float().numpy(), blk_e1=e1.float().numpy(),
                                   blk_temb=temb.float().numpy())
                    out[f"blk_out_h_{dt_name}"] = oh.float().numpy()
                    out[f"blk_out_e0_{dt_name}"] = oe0.float().numpy()
                    out[f"blk_out_e1_{dt_name}"] = oe1.float().numpy()
                    out[f"attn_out_h_{dt_name}"] = ah.float().numpy()
                    out[f"attn_out_e_{dt_name}"] = ae.float().numpy()
            m = mm.float()
            m.load_state_dict(weights, strict=False)
        np.savez_compressed(os.path.join(OUT, f"transformer_tiny_{variant}.npz"), **out)


VAE_TINY = dict(block_out_channels=(16, 16, 32, 32), layers_per_block=1, norm_num_groups=4, latent_channels=16,
                sample_height=96, sample_width=160, scaling_factor=0.7, temporal_compression_ratio=4)


def gen_vae():
    from diffusers import AutoencoderKLCogVideoX

    vae = AutoencoderKLCogVideoX(**VAE_TINY).float().eval()
    randomize(vae.decoder, torch.Generator().manual_seed(3), std=0.08)
    sd = {k: v for k, v in vae.state_dict().items() if k.startswith("decoder.")}
    gen = torch.Generator().manual_seed(4)
    lat = torch.randn(1, 5, 16, 12, 20, generator=gen)  # pipeline layout [B,F,C,H,W]
    out = npsd(sd)
    out["latents"] = lat.numpy()
    z = lat.permute(0, 2, 1, 3, 4) / vae.config.scaling_factor
    with torch.no_grad():
        # full outputs would be 3 MB each: keep every 3rd pixel + fp64 per-frame checksums
        for name, tiling in (("dec_untiled", False), ("dec_tiled", True)):
            (vae.enable_tiling if tiling else vae.disable_tiling)()
            y = vae.decode(z).sample
            out[name + "_s3"] = y[..., ::3, ::3].numpy()
            out[name + "_sum"] = y.double().sum(dim=(0, 1, 3, 4)).numpy()
            out[name + "_sq"] = (y.double() ** 2).sum(dim=(0, 1, 3, 4)).numpy()
        vae.disable_tiling()
        # a 2-frame latent (even branch everywhere) and a single-frame latent
        out["dec_2f"] = vae.decode(z[:, :, :2, :6, :8]).sample.numpy()
        out["dec_1f"] = vae.decode(z[:, :, :1, :6, :8]).sample.numpy()
    # Round 5: the same decoder in bf16 and fp16 (src/inference.py:239 moves the VAE to the pipeline dtype): untiled decode of the first two
    # latent frames of a 6 x 8 window, every pixel -- the reduced-precision rounding points of the conv / norm stack pinned to the reference
    for dt_name, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
        vh = AutoencoderKLCogVideoX(**VAE_TINY).eval()
        vh.load_state_dict(vae.state_dict())
        vh = vh.to(dt)
        with torch.no_grad():
            yh = vh.decode(z[:, :, :2, :6, :8].to(dt)).sample
        assert yh.dtype == dt and torch.isfinite(yh.float()).all()
        out[f"dec_2f_{dt_name}"] = yh.float().numpy()
    from diffusers.video_processor import VideoProcessor

    vp = VideoProcessor(vae_scale_factor=8)
    out["post_np"] = vp.postprocess_video(torch.from_numpy(out["dec_2f"]), output_type="np")
    np.savez_compressed(os.path.join(OUT, "vae_tiny.npz"), **out)


def gen_vae_enc():
    """Reference-image encode (src/video_generate.py:26-38): AutoencoderKLCogVideoX.encode of ONE frame, untiled and tiled,
    the posterior sample with captured noise, times scaling_factor."""
    from diffusers import AutoencoderKLCogVideoX

    vae = AutoencoderKLCogVideoX(**VAE_TINY).float().eval()
    randomize(vae.encoder, torch.Generator().manual_seed(5), std=0.08)
    sd = {k: v for k, v in vae.state_dict().items() if k.startswith("encoder.")}
    gen = torch.Generator().manual_seed(6)
    img = torch.rand(1, 3, 1, 96, 160, generator=gen) * 2.0 - 1.0  # [B,C,F,H,W] in [-1,1] like the normalised PNG
    out = npsd(sd)
    out["image"] = img.numpy()
    with torch.no_grad():
        for name, tiling in (("untiled", False), ("tiled", True)):
            (vae.enable_tiling if tiling else vae.disable_tiling)()
            post = vae.encode(img).latent_dist
            out[f"moments_{name}"] = post.parameters.numpy()
            g = torch.Generator().manual_seed(7)
            noise = torch.randn(post.mean.shape, generator=g)
            g = torch.Generator().manual_seed(7)
            smp = post.sample(generator=g) * vae.config.scaling_factor
            out[f"noise_{name}"] = noise.numpy()
            out[f"latent_{name}"] = smp.permute(0, 2, 1, 3, 4).numpy()  # [B,F,C,h,w] as handed to the pipeline
            assert torch.equal(post.mean + post.std * noise, post.sample(generator=torch.Generator().manual_seed(7)))
        vae.disable_tiling()
        out["moments_small"] = vae.encode(img[..., :40, :56]).latent_dist.parameters.numpy()
    np.savez_compressed(os.path.join(OUT, "vae_enc_tiny.npz"), **out)
    return sorted(sd)


T5_TINY = dict(vocab_size=100, d_model=128, d_kv=64, num_heads=2, d_ff=256, num_layers=2, feed_forward_proj="gated-gelu",
               relative_attention_num_buckets=32, relative_attention_max_distance=128)


def gen_t5():
    """Prompt embeddings (pipeline_cogvideox.py:227): transformers.T5EncoderModel(input_ids)[0], no attention mask, on a
    tiny seeded T5 v1.1 encoder; 40 tokens so that every relative-position bucket class (exact, log, clipped) occurs."""
    import transformers
    from transformers import T5Config, T5EncoderModel

    m = T5EncoderModel(T5Config(**T5_TINY)).float().eval()
    gen = torch.Generator().manual_seed(21)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "layer_norm" in n:
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=gen))
            elif "relative_attention_bias" in n:
                p.copy_(torch.randn(p.shape, generator=gen))
            elif n.endswith("shared.weight") or "embed_tokens" in n:
                p.copy_(torch.randn(p.shape, generator=gen))
            else:
                p.copy_(torch.randn(p.shape, generator=gen) * (1.5 / p.shape[1] ** 0.5))
    sd = {k: v for k, v in m.state_dict().items() if k != "encoder.embed_tokens.weight"}  # tied to shared.weight
    ids = torch.randint(0, 100, (2, 40), generator=gen)
    ids[1, 25:] = 0  # a padded prompt: pad id 0 repeated (attended like any token: no mask is passed)
    out = npsd(sd)
    out["input_ids"] = ids.numpy()
    with torch.no_grad():
        out["last_hidden_state"] = m(ids)[0].numpy()
        out["last_hidden_state_T9"] = m(ids[:, :9])[0].numpy()
        # fp16 (round 5): src/inference.py:209,214 moves the text encoder to the pipeline dtype -- fp16 for every non-5B checkpoint.  Two runs:
        # the plain one, and one whose block-0 feed-forward output overflows fp16 (its wo scaled by 3e4) so that T5Block's inf clamp acts
        mh = T5EncoderModel(T5Config(**T5_TINY)).eval()
        mh.load_state_dict(m.state_dict())
        mh = mh.half()
        out["last_hidden_state_f16"] = mh(ids)[0].float().numpy()
        mb = T5EncoderModel(T5Config(**T5_TINY)).eval()
        mb.load_state_dict(m.state_dict())
        out["last_hidden_state_bf16"] = mb.bfloat16()(ids)[0].float().numpy()
        key = "encoder.block.0.layer.1.DenseReluDense.wo.weight"
        sd2 = {k: v.clone() for k, v in m.state_dict().items()}
        sd2[key] = sd2[key] * 3.0e4
        mo = T5EncoderModel(T5Config(**T5_TINY)).eval()
        mo.load_state_dict(sd2)
        mo = mo.half()
        yo = mo(ids)[0].float()
        assert torch.isfinite(yo).all()
        out["last_hidden_state_f16_overflow"] = yo.numpy()
        out["f16_overflow_wo_scale"] = np.array(3.0e4)
    out["transformers_version"] = np.array(transformers.__version__)
    np.savez_compressed(os.path.join(OUT, "t5_tiny.npz"), **out)


def put_steps(out, name, tensors):
    """per-step tensors [1, F, C, H, W]: every second row and column (a quarter of the elements) + fp64 sum and abs-sum of the whole
    tensor per step, to keep the fixture small"""
    x = torch.stack(tensors).float()
    out[name + "_sub2"] = x[..., ::2, ::2].numpy()
    out[name + "_sums"] = torch.stack([x.double().sum(dim=(1, 2, 3, 4, 5)), x.double().abs().sum(dim=(1, 2, 3, 4, 5))], dim=1).numpy()


def gen_pipeline():
    """Full CustomCogVideoXPipeline.__call__ (src/custom_cogvideox_pipe.py:125-326) with tiny modules, 480x720
    (the only geometry the shipped harness supports: 1350 tokens per frame), 3 steps, DDIM and DPM."""
    from diffusers import AutoencoderKLCogVideoX, CogVideoXDDIMScheduler, CogVideoXDPMScheduler, CogVideoXTransformer3DModel
    from custom_cogvideox_pipe import CustomCogVideoXPipeline

    cfg = dict(TINY)
    cfg.update(sample_width=90, sample_height=60, sample_frames=5, max_text_seq_length=6)
    tr = CogVideoXTransformer3DModel(use_rotary_positional_embeddings=True, **cfg).float().eval()
    randomize(tr, torch.Generator().manual_seed(9), std=0.1)
    vae = AutoencoderKLCogVideoX(block_out_channels=(8, 8, 8, 8), layers_per_block=1, norm_num_groups=2,
                                 latent_channels=16, scaling_factor=0.7).float().eval()
    gen = torch.Generator().manual_seed(13)
    pe = torch.randn(1, 6, 64, generator=gen)
    ne = torch.randn(1, 6, 64, generator=gen)
    ref = torch.randn(1, 1, 16, 60, 90, generator=gen) * 0.7
    lat0 = torch.randn(1, 2, 16, 60, 90, generator=gen)
    weights = {k: v for k, v in tr.state_dict().items() if "pos_embedding" not in k}
    out = npsd(weights)
    out.update(prompt_embeds=pe.numpy(), negative_prompt_embeds=ne.numpy(), ref=ref.numpy(), latents0=lat0.numpy())
    randomize(vae, torch.Generator().manual_seed(10), std=0.15)
    out.update({"vae:" + k: v.numpy() for k, v in vae.state_dict().items()})

    def run(pipe, trace=None, **kw):
        """one CustomCogVideoXPipeline.__call__; trace collects what the loop hands to scheduler.step (the CFG-combined noise
        prediction, custom_cogvideox_pipe.py:273-277) and what it gets back (the latents after the step, :280-295).  Not through
        callback_on_step_end: the reference builds its callback arguments with `{k: locals()[k] for k in ...}` (:300), which raises
        KeyError under Python < 3.12 (a comprehension has its own locals there) -- that seam cannot be exercised with this
        interpreter, so the trace wraps scheduler.step instead."""
        g = torch.Generator().manual_seed(77)
        args = dict(prompt=None, prompt_embeds=pe, negative_prompt_embeds=ne, ref_img_states=ref, height=480, width=720,
                    num_frames=5, num_inference_steps=3, guidance_scale=6.0, latents=lat0.clone(), generator=g,
                    output_type="latent", return_dict=False)
        args.update(kw)
        if trace is not None:
            step = pipe.scheduler.step

            @functools.wraps(step)  # prepare_extra_step_kwargs inspects the signature for `eta` / `generator` (pipeline_cogvideox.py:355-370)
            def spy(model_output, *a, **k):
                trace["noise_pred"].append(model_output.detach().float().clone())
                ret = step(model_output, *a, **k)
                trace["latents"].append(ret[0].detach().to(args["prompt_embeds"].dtype).float().clone())  # `latents.to(prompt_embeds.dtype)`, :296
                return ret

            pipe.scheduler.step = spy
        try:
            return pipe(**args)[0], g
        finally:
            if trace is not None:
                pipe.scheduler.step = step

    for kind, cls in (("ddim", CogVideoXDDIMScheduler), ("dpm", CogVideoXDPMScheduler)):
        sched = cls(**sched_kwargs(1.0))
        pipe = CustomCogVideoXPipeline(tokenizer=None, text_encoder=None, transformer=tr, vae=vae, scheduler=sched,
                                       customization=True)
        st = torch.Generator().manual_seed(77).get_state()
        trace = {"noise_pred": [], "latents": []}
        res, _ = run(pipe, trace)
        out[f"final_{kind}"] = res.float().numpy()
        put_steps(out, f"steps_noise_pred_{kind}", trace["noise_pred"])  # [steps, 1, F, C, H, W], after CFG, fp32
        put_steps(out, f"steps_latents_{kind}", trace["latents"])
        assert torch.equal(trace["latents"][-1], res.float())
        if kind == "dpm":  # the randn draws the DPM steps consumed, in order (2 per step after step 0, 1 on step 0... )
            g3 = torch.Generator()
            g3.set_state(st)
            # the tests regenerate the draws from torch.Generator().manual_seed(77); pin them by checksum
            nz = torch.stack([torch.randn(lat0.shape, generator=g3) for _ in range(6)])
            out["dpm_noise_seed"] = np.array(77)
            out["dpm_noise_sums"] = nz.double().sum(dim=(1, 2, 3, 4, 5)).numpy()
        if kind == "ddim":
            # use_dynamic_cfg=True (custom_cogvideox_pipe.py:268-271): the guidance scale follows a cosine power schedule
            trace = {"noise_pred": [], "latents": []}
            res, _ = run(pipe, trace, use_dynamic_cfg=True)
            out["final_ddim_dyncfg"] = res.float().numpy()
            put_steps(out, "steps_latents_ddim_dyncfg", trace["latents"])
            # output_type="np" through the VAE with tiling on, as src/inference.py:204-207 runs it: decode_latents (:309-311) +
            # VideoProcessor.postprocess_video.  The frames are 8 x 480 x 720 x 3 floats: the fixture keeps every 8th pixel of every
            # frame plus fp64 sums per frame and per colour plane (a checksum of the rest)
            vae.enable_tiling()
            frames, _ = run(pipe, None, output_type="np")
            vae.disable_tiling()
            frames = np.asarray(frames)
            out["frames_ddim_tiled_shape"] = np.array(frames.shape)
            out["frames_ddim_tiled_sub8"] = frames[:, :, ::8, ::8, :].astype(np.float32)
            out["frames_ddim_tiled_sums"] = frames.astype(np.float64).sum(axis=(2, 3))  # [1, F, 3]
    # Round 5: the same three DDIM steps with the WHOLE pipeline in bf16 and in fp16 (src/inference.py:191,209: bf16 for 5B checkpoints, fp16 for
    # the others): the loop's own rounding points -- `latents.to(prompt_embeds.dtype)` (:296), the scheduler on reduced-precision samples, the
    # transformer in that dtype -- pinned against the reference itself, not only against the oracle.  Appended after everything else: the fp32
    # entries above keep their bits.
    import copy

    for dt_name, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
        trh = copy.deepcopy(tr).to(dt)
        sched = CogVideoXDDIMScheduler(**sched_kwargs(1.0))
        pipe = CustomCogVideoXPipeline(tokenizer=None, text_encoder=None, transformer=trh, vae=vae, scheduler=sched, customization=True)
        trace = {"noise_pred": [], "latents": []}
        res, _ = run(pipe, trace, prompt_embeds=pe.to(dt), negative_prompt_embeds=ne.to(dt), ref_img_states=ref.to(dt), latents=lat0.to(dt).clone())
        assert res.dtype == dt and torch.isfinite(res.float()).all()
        out[f"final_ddim_{dt_name}"] = res.float().numpy()
        put_steps(out, f"steps_latents_ddim_{dt_name}", trace["latents"])
        put_steps(out, f"steps_noise_pred_ddim_{dt_name}", trace["noise_pred"])
    np.savez_compressed(os.path.join(OUT, "pipeline_tiny.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    import_reference()
    torch.manual_seed(0)
    which = sys.argv[1:] or ["tables", "sched", "transformer", "vae", "vae_enc", "t5", "pipeline"]
    for w in which:
        print("generating", w, flush=True)
        globals()["gen_" + w]()
    print("done")
