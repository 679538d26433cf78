"""Experiment: tiled VAE decode time against what else lives in the process (a loaded transformer engine costs it 40 ms: why?).
Usage: python tools/vae_inproc_probe.py [order]   order = vae_first | engine_first"""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
import bench
dev = "cuda:0"; dt = torch.bfloat16
order = sys.argv[1] if len(sys.argv) > 1 else "vae_first"


def make_vae():
    vcfg = s2v.VAEConfig(scaling_factor=0.7)
    vae = s2v.HipAutoencoderKLCogVideoX(vcfg, dt, dev)
    vae.load_state_dict(s2v.weights.synthetic_vae_state_dict(vcfg, seed=7, device=dev))
    return vae


def vae_time(vae, tag):
    lat = torch.randn(1, 13, 16, 60, 90, device=dev).to(dt)
    for tiling in (False, True):
        vae.use_tiling = tiling
        vae.decode_latents(lat); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); vae.decode_latents(lat); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"[{order}] {tag} tiling={tiling}: " + " ".join(f"{t*1e3:.0f}" for t in ts) + f" ms   free {torch.cuda.mem_get_info()[0]/2**30:.0f} GiB", flush=True)


def make_engine():
    preset, F, H, W, T = bench.WORKLOADS["cogvideox-5b-49x480x720"]
    cfg = s2v.config.PRESETS[preset]()
    eng = s2v.S2VEngine(cfg, dt, dev)
    bench.load_synthetic(s2v, eng, cfg, 1234)
    torch.cuda.empty_cache()
    return eng


if order == "vae_first":
    vae = make_vae()
    vae_time(vae, "alone")
    eng = make_engine()
    vae_time(vae, "engine created afterwards")
elif order == "engine_first":
    eng = make_engine()
    vae = make_vae()
    vae_time(vae, "engine created before")
else:  # streams only: a few idle non-blocking streams instead of an engine
    vae = make_vae()
    vae_time(vae, "alone")
    ss = [torch.cuda.Stream() for _ in range(3)]
    vae2 = make_vae()
    vae_time(vae2, "second VAE object after 3 extra torch streams")
