"""The caller of the path, mirroring `inference()` of src/video_generate.py:7-66 on the drop-in objects: reference image ->
VAE encode -> posterior sample * scaling_factor -> [B,F,C,h,w]; prompt ids -> T5 embeddings; denoise loop; VAE decode ->
frames -> video file.  Host-only steps that stay with the caller: reading the PNG (PIL) and tokenisation (sentencepiece):
`inference` takes the decoded image array and the token ids; `export_to_video` writes the file (imageio-ffmpeg when present,
a built-in Motion-JPEG AVI writer otherwise)."""
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
    `output_type` selects ("latent" / "pt").

    Random draws.  The reference creates and seeds a `torch.Generator` (video_generate.py:21-23) and then never passes it anywhere:
    `.latent_dist.sample()` (:37) and the pipeline call (:44-56) carry no generator, so BOTH draws -- the posterior sample of the reference
    image first, the initial latents second (pipeline_cogvideox.py:320-344) -- come from the device's GLOBAL generator, seeded by
    seed_everything (src/inference.py:28-35: torch.manual_seed + torch.cuda.manual_seed_all).  Here a private device generator seeded with
    `seed` produces the same two draws in the same order (a freshly seeded Philox stream is the same stream whichever generator object
    holds it: tests/test_gpu_end_to_end.py pins the equality) without touching the process-wide RNG state; seed=None draws from the global
    device generator exactly as the reference does.

    CFG-parallel (round 6): pass `cfg_parallel=dist.CfgPair(...)` through **pipe_kwargs on BOTH ranks of a pair with the same image, ids and seed: each
    rank encodes the reference image and the prompts itself (42 ms + 10 ms, step-invariant), runs its half of every step, and both return the same
    frames."""
    dev = pipe.transformer.device
    generator = None
    if seed is not None:
        generator = torch.Generator(device=dev)
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


def _write_mjpeg_avi(frames, path, fps, quality=95):
    """Minimal Motion-JPEG AVI writer (RIFF 'AVI ', one 'vids'/'MJPG' stream, idx1 index): every frame is a baseline JPEG
    encoded by PIL.  Used when imageio-ffmpeg is absent; plays in ffplay / VLC / browsers' <video> via ffmpeg remux."""
    import io
    import struct

    from PIL import Image

    F, H, W, _ = frames.shape
    jpegs = []
    for f in frames:
        buf = io.BytesIO()
        Image.fromarray(f, "RGB").save(buf, format="JPEG", quality=quality)
        b = buf.getvalue()
        jpegs.append(b + (b"\0" if len(b) & 1 else b""))

    def chunk(tag, data):
        return tag + struct.pack("<I", len(data)) + data + (b"\0" if len(data) & 1 else b"")

    def lst(tag, data):
        return b"LIST" + struct.pack("<I", len(data) + 4) + tag + data

    scale, rate = 1000, int(round(fps * 1000))
    maxb = max(len(j) for j in jpegs)
    avih = struct.pack("<14I", int(round(1e6 / fps)), maxb * int(max(1, round(fps))), 0, 0x10, F, 0, 1, maxb, W, H, 0, 0, 0, 0)
    strh = b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIII4H", 0, 0, 0, 0, scale, rate, 0, F, maxb, 0xFFFFFFFF, 0, 0, 0, W, H)
    strf = struct.pack("<IiiHH4sIiiII", 40, W, H, 1, 24, b"MJPG", W * H * 3, 0, 0, 0, 0)
    hdrl = lst(b"hdrl", chunk(b"avih", avih) + lst(b"strl", chunk(b"strh", strh) + chunk(b"strf", strf)))
    movi_body, index, off = b"", b"", 4
    for j in jpegs:
        movi_body += b"00dc" + struct.pack("<I", len(j)) + j
        index += b"00dc" + struct.pack("<III", 0x10, off, len(j))
        off += 8 + len(j)
    body = hdrl + lst(b"movi", movi_body) + chunk(b"idx1", index)
    with open(path, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", len(body) + 4) + b"AVI " + body)


def read_avi_info(path):
    """(frames, fps, width, height, first frame as uint8 [H, W, 3]) of a file written by _write_mjpeg_avi -- the check that the
    container really holds what export_to_video was given"""
    import io
    import struct

    from PIL import Image

    d = open(path, "rb").read()
    assert d[:4] == b"RIFF" and d[8:12] == b"AVI "
    i = d.index(b"avih") + 8
    us_per_frame, _, _, _, total, _, _, _, w, h = struct.unpack("<10I", d[i:i + 40])
    i = d.index(b"strh") + 8
    scale, rate = struct.unpack("<II", d[i + 20:i + 28])
    m = d.index(b"movi") + 4
    assert d[m:m + 4] == b"00dc"
    n = struct.unpack("<I", d[m + 4:m + 8])[0]
    first = np.asarray(Image.open(io.BytesIO(d[m + 8:m + 8 + n])).convert("RGB"))
    nidx = struct.unpack("<I", d[d.rindex(b"idx1") + 4:d.rindex(b"idx1") + 8])[0] // 16
    assert nidx == total
    return total, rate / scale, w, h, first


def export_to_video(frames_uint8, output_video_path, fps=8):
    """utils/export_utils.py:143-186 with the frames already uint8 [F,H,W,3] (HipAutoencoderKLCogVideoX.frames_uint8).  With
    imageio + imageio-ffmpeg present the frames go to `imageio.get_writer` exactly as in the reference (mp4 / H.264).  Without
    them (this image has neither, and no ffmpeg binary) the frames are written as a Motion-JPEG AVI next to the requested path
    (same stem, `.avi`): a real video file with the same frames and frame rate, not an mp4.  Returns the path written."""
    frames = frames_uint8.cpu().numpy() if hasattr(frames_uint8, "cpu") else np.asarray(frames_uint8)
    if frames.dtype != np.uint8 or frames.ndim != 4 or frames.shape[3] != 3:
        raise ValueError("export_to_video takes uint8 frames [F, H, W, 3] (HipAutoencoderKLCogVideoX.frames_uint8)")
    try:
        import imageio
        imageio.plugins.ffmpeg.get_exe()
    except Exception:  # ImportError / AttributeError / missing binary
        import os

        path = os.path.splitext(output_video_path)[0] + ".avi"
        _write_mjpeg_avi(frames, path, fps)
        return path
    with imageio.get_writer(output_video_path, fps=fps) as writer:
        for frame in frames:
            writer.append_data(frame)
    return output_video_path
