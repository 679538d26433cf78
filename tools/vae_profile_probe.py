"""tiled / untiled VAE decode of one 13 x 60 x 90 latent (C3) for rocprofv3 --kernel-trace --stats"""
import importlib, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
DEV = "cuda:0"
cfg = s2v.VAEConfig(scaling_factor=0.7)
vae = s2v.HipAutoencoderKLCogVideoX(cfg, torch.bfloat16, DEV)
vae.load_state_dict(s2v.weights.synthetic_vae_state_dict(cfg, seed=7, device=DEV))
if "tiled" in sys.argv:
    vae.enable_tiling()
lat = torch.randn(1, 13, 16, 60, 90, device=DEV).bfloat16()
vae.decode_latents(lat); torch.cuda.synchronize()
if "once" in sys.argv:  # tools/vae_conv_rates.py: one more decode only (the conv log then holds exactly two decodes)
    vae.decode_latents(lat); torch.cuda.synchronize()
    sys.exit(0)
t0 = time.time()
for _ in range(3):
    vae.decode_latents(lat)
torch.cuda.synchronize()
print(f"decode {'tiled' if 'tiled' in sys.argv else 'untiled'}: {(time.time() - t0) / 3 * 1e3:.1f} ms")
