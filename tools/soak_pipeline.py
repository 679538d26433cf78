"""Soak: the full 50-step denoise loop at C3 (hipGraph replay, persistent attention launches: 2 x 2100 queue hand-overs) + tiled VAE
decode, twice -- outputs finite and bit-identical between the two runs."""
import importlib, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
DEV = "cuda:0"
cfg = s2v.cogvideox_5b()
m = s2v.HipCogVideoXTransformer3DModel(cfg, torch.bfloat16, DEV)
m.load_state_dict(s2v.weights.synthetic_state_dict(cfg, seed=1, device=DEV))
vcfg = s2v.VAEConfig(scaling_factor=cfg.vae_scaling_factor)
vae = s2v.HipAutoencoderKLCogVideoX(vcfg, torch.bfloat16, DEV)
vae.load_state_dict(s2v.weights.synthetic_vae_state_dict(vcfg, seed=2, device=DEV))
vae.enable_tiling()
pipe = s2v.S2VPipeline(m, s2v.CogVideoXDDIMScheduler(snr_shift_scale=cfg.snr_shift_scale), vae)
g = torch.Generator(device=DEV).manual_seed(3)
pe = torch.randn(1, 226, 4096, generator=g, device=DEV).bfloat16()
ne = torch.randn(1, 226, 4096, generator=g, device=DEV).bfloat16()
ref = (torch.randn(1, 1, 16, 60, 90, generator=g, device=DEV) * 0.7).bfloat16()
lat0 = torch.randn(1, 13, 16, 60, 90, generator=g, device=DEV).bfloat16()
outs = []
for run in range(2):
    t0 = time.time()
    lat = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, ref_img_states=ref, height=480, width=720, num_frames=49,
               num_inference_steps=50, guidance_scale=6.0, latents=lat0.clone(), output_type="latent", return_dict=False, use_graph=True)[0]
    video = vae.decode_latents(lat)
    torch.cuda.synchronize()
    print(f"run {run}: {time.time() - t0:.1f} s, latents finite {bool(torch.isfinite(lat.float()).all())}, video finite {bool(torch.isfinite(video.float()).all())}, "
          f"|lat| max {lat.float().abs().max().item():.3f}")
    outs.append((lat.clone(), video.clone()))
print("bit-identical runs:", torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]))
