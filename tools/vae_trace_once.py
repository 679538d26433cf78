"""one warm-up + one untiled decode at 480 x 720 for a rocprofv3 --kernel-trace --stats run (per-kernel times of the VAE)"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
vcfg = s2v.VAEConfig(scaling_factor=0.7)
vae = s2v.HipAutoencoderKLCogVideoX(vcfg, torch.bfloat16, "cuda:0")
vae.load_state_dict(s2v.weights.synthetic_vae_state_dict(vcfg, seed=7, device="cuda:0"))
lat = torch.randn(1, 13, 16, 60, 90, generator=torch.Generator().manual_seed(3)).to("cuda:0", torch.bfloat16)
vae.use_tiling = "tiled" in sys.argv
for _ in range(2):
    vae.decode_latents(lat)
torch.cuda.synchronize()
