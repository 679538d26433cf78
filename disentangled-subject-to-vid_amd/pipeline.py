"""The denoise loop of the hot path, mirroring CustomCogVideoXPipeline.__call__
(src/custom_cogvideox_pipe.py:125-326) from step 4 on: prompt embeddings are inputs (the T5 encoder is a caller-side
step), everything after them runs here.

Two execution modes with identical results:
  fused=True  (default) one s2v_denoise_step per iteration: transformer on the CFG pair, fp32 CFG, scheduler step
               and the round to the model dtype in one (optionally hipGraph-captured) launch sequence;
  fused=False the reference's own sequence of seam calls: transformer(...) -> .float() -> CFG -> scheduler.step ->
               .to(dtype), each through its drop-in object.

Deviations from the shipped harness, both explicit: the 1350-tokens-per-frame constant (:228-235) is generalised to
(H/16)(W/16) and non-RoPE models skip the RoPE slicing instead of crashing (:223-231) -- needed for the 2B and
non-480x720 configurations of BASELINE.json; module arithmetic is unchanged.
"""
import math

import numpy as np
import torch

from . import _lib, tables
from .schedulers import CogVideoXDPMScheduler


class S2VPipeline:
    _callback_tensor_inputs = ["latents", "prompt_embeds", "negative_prompt_embeds"]  # pipeline_cogvideox.py:166-170

    def __init__(self, transformer, scheduler, vae=None, vae_scale_factor_spatial=8, vae_scale_factor_temporal=4):
        self.transformer, self.scheduler, self.vae = transformer, scheduler, vae
        self.vae_scale_factor_spatial = vae_scale_factor_spatial
        self.vae_scale_factor_temporal = vae_scale_factor_temporal
        self._guidance_scale = 1.0
        self.interrupt = False

    @property
    def guidance_scale(self):
        return self._guidance_scale

    def check_inputs(self, height, width, prompt_embeds, negative_prompt_embeds):
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if prompt_embeds is None:
            raise ValueError("Provide `prompt_embeds` (the text encoder is not part of this path).")
        if negative_prompt_embeds is not None and prompt_embeds.shape != negative_prompt_embeds.shape:
            raise ValueError("`prompt_embeds` and `negative_prompt_embeds` must have the same shape when passed "
                             f"directly, but got {prompt_embeds.shape} != {negative_prompt_embeds.shape}.")

    def prepare_latents(self, num_frames, height, width, dtype, device, generator, latents=None):
        shape = (1, (num_frames - 1) // self.vae_scale_factor_temporal + 1, self.transformer.config.in_channels,
                 height // self.vae_scale_factor_spatial, width // self.vae_scale_factor_spatial)
        if latents is None:
            gdev = generator.device if generator is not None else device
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)
        else:
            latents = latents.to(device)
        return latents * self.scheduler.init_noise_sigma

    @torch.no_grad()
    def __call__(self, prompt_embeds=None, negative_prompt_embeds=None, ref_img_states=None, height=480, width=720,
                 num_frames=49, num_inference_steps=50, guidance_scale=6.0, use_dynamic_cfg=False, generator=None,
                 latents=None, output_type="latent", return_dict=True, fused=True, use_graph=False,
                 callback_on_step_end=None, callback_on_step_end_tensor_inputs=("latents",), cfg_parallel=None):
        """cfg_parallel: a dist.CfgPair -- this process runs ONE sample of the CFG pair (slot 0: negative prompt, slot 1: prompt) on its GPU and its
        peer the other; every rank of the pair passes the SAME arguments (embeddings, reference latent, latents or an equally seeded generator) and
        returns the same latents / video bit for bit.  fused mode only."""
        if num_frames > 49:
            raise ValueError("The number of frames must be less than or equal to 49 due to static positional embeddings.")
        self.check_inputs(height, width, prompt_embeds, negative_prompt_embeds)
        bad = [k for k in callback_on_step_end_tensor_inputs if k not in self._callback_tensor_inputs]
        if bad:  # pipeline_cogvideox.py:385-390
            raise ValueError(f"`callback_on_step_end_tensor_inputs` has to be in {self._callback_tensor_inputs}, but found {bad}")
        if negative_prompt_embeds is None:
            raise ValueError("Provide `negative_prompt_embeds`: the reference builds them from the empty prompt with its text "
                             "encoder (pipeline_cogvideox.py:239-318), which is a caller-side step here "
                             "(video_generate.inference does it)")
        if ref_img_states is None:
            raise ValueError("Provide `ref_img_states` (the VAE latent of the reference image, [1, 1, C, H/8, W/8])")
        if prompt_embeds.shape[0] != 1:
            raise RuntimeError("one prompt per call: the transformer duplicates the reference tokens exactly x2 "
                               "(cogvideox_transformer_3d.py:503-504)")
        if guidance_scale <= 1.0:
            raise RuntimeError("guidance_scale must be > 1: eval=True duplicates the reference tokens for the CFG pair")
        tr, sch = self.transformer, self.scheduler
        eng, dt, dev = tr.engine, tr.dtype, tr.device
        text = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0).to(dev, dt)
        sch.set_timesteps(num_inference_steps, device="cpu")
        timesteps = sch.timesteps
        latents = self.prepare_latents(num_frames, height, width, dt, dev, generator, latents).to(dt).contiguous()
        F, H, W = latents.shape[1], latents.shape[3], latents.shape[4]
        ref = ref_img_states.to(dev, dt)
        is_dpm = isinstance(sch, CogVideoXDPMScheduler)
        rope = ref_rope = None
        if tr.config.use_rotary_positional_embeddings:
            cos, sin = tables.rope_tables(height, width, F)
            n = (H // 2) * (W // 2)
            cos, sin = torch.from_numpy(cos).to(dev), torch.from_numpy(sin).to(dev)
            ref_rope, rope = (cos[:n], sin[:n]), (cos[n:], sin[n:])

        if cfg_parallel is not None and not fused:
            raise ValueError("cfg_parallel runs the fused step (one sample of the CFG pair per rank); fused=False is the reference's seam sequence")
        if cfg_parallel is not None:  # both ranks compute from the same inputs or the video is garbage: checked once, collectively
            cfg_parallel.assert_same(latents=latents, ref_img_states=ref, text=text)
        if fused:
            # CFG-parallel: B = 1 with this rank's half of [negative | positive] (custom_cogvideox_pipe.py:196) and the un-duplicated reference tokens
            my_text = text if cfg_parallel is None else text[cfg_parallel.slot:cfg_parallel.slot + 1]
            eng.set_geometry(2 if cfg_parallel is None else 1, text.shape[1], F, H, W)
            eng.prepare_tables(height, width)
            eng.set_conditioning(my_text, ref)
            x0_hist = torch.zeros(latents.shape, dtype=torch.float32, device=dev) if is_dpm else None
            noise = torch.empty_like(latents) if is_dpm else None
        old = None
        for i, t in enumerate(timesteps):
            if self.interrupt:
                break
            g = guidance_scale
            if use_dynamic_cfg:
                g = 1 + guidance_scale * ((1 - math.cos(math.pi * ((num_inference_steps - i) / num_inference_steps) ** 5.0)) / 2)
            self._guidance_scale = g
            if fused:
                if is_dpm:
                    coef = sch.coef(t, timesteps[i - 1] if i > 0 else None, i == 0, dt, g)
                    self._draw(noise, generator)
                    if coef.kind == 2:
                        self._draw(noise, generator)  # the reference discards its first draw on multistep steps
                else:
                    coef = sch.coef(t, dt, g)
                if cfg_parallel is None:
                    eng.denoise_step(latents, float(t), coef, x0_hist, noise, use_graph)
                else:
                    cfg_parallel.step(eng, latents, float(t), coef, x0_hist, noise, use_graph)
            else:
                x = torch.cat([latents] * 2)
                x = sch.scale_model_input(x, t)
                noise_pred = tr(hidden_states=x, encoder_hidden_states=text, ref_img_states=ref,
                                timestep=t.expand(2), image_rotary_emb=rope, ref_image_rotary_emb=ref_rope,
                                return_dict=False, eval=True)[0].float()
                u, c = noise_pred.chunk(2)
                noise_pred = u + g * (c - u)
                if not is_dpm:
                    latents = sch.step(noise_pred, t, latents, return_dict=False)[0]
                else:
                    latents, old = sch.step(noise_pred, old, t, timesteps[i - 1] if i > 0 else None, latents,
                                            generator=generator, return_dict=False)
                latents = latents.to(dt)
            if callback_on_step_end is not None:
                # custom_cogvideox_pipe.py:298-305: the callback sees the requested tensors -- `prompt_embeds` is, at that point of
                # the reference loop, the CONCATENATED [negative | positive] pair (:196) -- and what it returns replaces them for
                # the following steps (a returned `negative_prompt_embeds` is rebound there too, but nothing reads it after :196).
                # A callback that returns nothing keeps everything (the reference would raise on `None.get`).
                avail = {"latents": latents, "prompt_embeds": text, "negative_prompt_embeds": negative_prompt_embeds}
                outs = callback_on_step_end(self, i, t, {k: avail[k] for k in callback_on_step_end_tensor_inputs}) or {}
                new_lat = outs.get("latents", latents)
                if new_lat is not latents:
                    if fused:  # the step (and its captured graph) updates ONE buffer in place: keep it, take the values
                        latents.copy_(new_lat.to(dev, dt).reshape(latents.shape))
                    else:
                        latents = new_lat.to(dev, dt)
                new_text = outs.get("prompt_embeds", text)
                if new_text is not text:
                    text = new_text.to(dev, dt)
                    if fused:  # the hoisted text projection follows the new embeddings
                        eng.set_conditioning(text if cfg_parallel is None else text[cfg_parallel.slot:cfg_parallel.slot + 1], ref)
                negative_prompt_embeds = outs.get("negative_prompt_embeds", negative_prompt_embeds)
        # attn_p_format "auto" settled on the census of the FIRST step; the whole run's census is kept for the caller and, should later (less
        # noisy, sharper) steps have taken the fp16 kernel's slow path too often, the next video of this engine runs on bf16 probabilities
        self.attn_slow_fraction = None
        eng_ = getattr(tr, "engine", None)
        if fused and eng_ is not None and eng_.cfg.attn_p_format == "auto" and eng_.attn_p_format == "f16":
            slow, total = eng_.attn_slow_stats(reset=True)
            self.attn_slow_fraction = slow / total if total else 0.0
            if self.attn_slow_fraction > eng_.AUTO_SLOW_FRACTION:
                eng_.set_attn_p_format("bf16")
        if output_type == "latent":
            video = latents
        else:
            if self.vae is None:
                raise ValueError("a VAE object is needed for output_type != 'latent'")
            video = self.decode_latents(latents)
            video = self.vae.postprocess_video(video, output_type)
        return (video,) if not return_dict else {"frames": video}

    @staticmethod
    def _draw(buf, generator):
        gdev = generator.device.type if generator is not None else buf.device.type
        if gdev == "cpu":
            buf.copy_(torch.randn(buf.shape, generator=generator, dtype=buf.dtype))
        else:
            buf.normal_(generator=generator)

    def decode_latents(self, latents):
        """pipeline_cogvideox.py:346-351"""
        return self.vae.decode_latents(latents)
