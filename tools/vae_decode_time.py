"""Tiled / untiled VAE decode times at the real widths, one process, nothing else on the GPU.
    python tools/vae_decode_time.py [480x720] [720x1280]      (S2V_LIB=<diag library> S2V_VAE_NO_DIRECT_CONV_OUT=1: the implicit-GEMM conv_out for an A/B)
Order per geometry: tiled, untiled, tiled again (the third shows what a tiled decode costs AFTER an untiled one grew the workspace: round 6 fix)."""
import importlib, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
DEV = "cuda:0"
geos = [a for a in sys.argv[1:] if "x" in a] or ["480x720", "720x1280"]
dt = torch.bfloat16
vcfg = s2v.VAEConfig(scaling_factor=0.7)
vae = s2v.HipAutoencoderKLCogVideoX(vcfg, dt, DEV)
vae.load_state_dict(s2v.weights.synthetic_vae_state_dict(vcfg, seed=7, device=DEV))


def timed(lat, tiling, n=3):
    vae.use_tiling = tiling
    out = vae.decode_latents(lat)
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        out = vae.decode_latents(lat)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts), out


for geo in geos:
    H, W = (int(x) // 8 for x in geo.split("x"))
    lat = torch.randn(1, 13, 16, H, W, generator=torch.Generator().manual_seed(3)).to(DEV, dt)
    ref = None
    for label, tiling in (("tiled", True), ("untiled", False), ("tiled after untiled", True)):
        t, out = timed(lat, tiling)
        sets, nbytes = vae.workspace_info()
        same = ""
        if tiling:
            if ref is None:
                ref = out.clone()
            else:
                same = f"  bit-identical to the first tiled decode: {torch.equal(out, ref)}"
        print(f"{geo} {label:20s}: {t * 1e3:8.1f} ms   workspace sets {sets} x {nbytes / 1e9:.1f} GB   finite {bool(torch.isfinite(out.float()).all())}{same}", flush=True)
