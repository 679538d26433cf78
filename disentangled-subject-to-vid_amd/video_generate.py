"""The caller of the path, mirroring `inference()` of src/video_generate.py:7-66 on the drop-in objects: reference image ->
VAE encode -> posterior sample * scaling_factor -> [B,F,C,h,w]; prompt ids -> T5 embeddings; denoise loop; VAE decode ->
frames.  Host-only steps that stay with the caller: reading the PNG (PIL), tokenisation (sentencepiece) and the mp4
export (ffmpeg, utils/export_utils.py:143-186): this function takes the decoded image array and the token ids."""
import numpy as np
import torch


def reference_latents(vae, ref_image_uint8, generator=None):
    """src/video_generate.py:26-38.  ref_image_uint8: [H, W, 3] uint8 (np.array(Image.open(..).convert('RGB')))."""
    img = np.asarray(ref_image_uint8)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError("reference image must be [H, W, 3]")
    x = torch.from_numpy(np.expand_dims(img, axis=0)).float() / 255.0 * 2.0 - 1.0  # [1,H,W,3]
    x = x.permute(0, 3, 1, 2)                                                     # [1,3,H,W]
    x = x.unsqueeze(0).permute(0, 2, 1, 3, 4).to(device=vae.device, dtype=vae.dtype)  # [1,3,1,H,W]
    z = vae.encode(x).latent_dist.sample(generator) * vae.config.scaling_factor
    return z.permute(0, 2, 1, 3, 4)                                                # [1,1,C,h,w]


def prompt_embeddings(text_encoder, input_ids, dtype=None):
    """pipeline_cogvideox.py:227-228: text_encoder(ids)[0], cast to the pipeline dtype"""
    emb = text_encoder(input_ids)[0]
    return emb.to(dtype or emb.dtype)


def inference(pipe, text_encoder, ref_image_uint8, prompt_ids, negative_prompt_ids, height=480, width=720, num_frames=49,
              num_inference_steps=50, guidance_scale=6.0, use_dynamic_cfg=False, seed=None, latents=None, output_type="np",
              **pipe_kwargs):
    """Returns the frames [F, H, W, 3] float32 in [0, 1] (what the reference hands to export_to_video), or whatever
    `output_type` selects ("latent" / "pt").  One generator drives the reference-image posterior draw and then the initial
    latents, in that order, like the reference's single `torch.Generator` (video_generate.py:21-23,37)."""
    dev = pipe.transformer.device
    generator = torch.Generator(device=dev)
    if seed is not None:
        generator.manual_seed(seed)
    ref = reference_latents(pipe.vae, ref_image_uint8, generator)
    pe = prompt_embeddings(text_encoder, prompt_ids, pipe.transformer.dtype)
    ne = prompt_embeddings(text_encoder, negative_prompt_ids, pipe.transformer.dtype)
    out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, ref_img_states=ref, height=height, width=width,
               num_frames=num_frames, num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
               use_dynamic_cfg=use_dynamic_cfg, generator=generator, latents=latents, output_type=output_type,
               return_dict=True, **pipe_kwargs)
    video = out["frames"]
    return video[0] if output_type == "np" else video


def export_to_video(frames_uint8, output_video_path, fps=8):
    """utils/export_utils.py:143-186 with the frames already uint8 [F,H,W,3] (HipAutoencoderKLCogVideoX.frames_uint8): the mp4
    container / codec work is imageio-ffmpeg's (host side, third party); this function only hands the frames over and fails
    loudly when that backend is absent -- there is no other encoder on the path."""
    try:
        import imageio
        imageio.plugins.ffmpeg.get_exe()
    except Exception as e:  # ImportError / AttributeError / missing binary
        raise RuntimeError("export_to_video needs imageio + imageio-ffmpeg (src/video_generate.py:65-66); not present here") from e
    frames = frames_uint8.cpu().numpy() if hasattr(frames_uint8, "cpu") else np.asarray(frames_uint8)
    with imageio.get_writer(output_video_path, fps=fps) as writer:
        for frame in frames:
            writer.append_data(frame)
    return output_video_path
