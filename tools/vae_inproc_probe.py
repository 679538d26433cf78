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


def vae_time(vae, tag, modes=(False, True)):
    lat = torch.randn(1, 13, 16, 60, 90, device=dev).to(dt)
    for tiling in modes:
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
elif order in ("blame_torch", "blame_create", "blame_load_nolora"):  # which part of "an engine exists" costs the decoder its 40 ms?
    if order == "blame_torch":      # only torch work of the size the synthetic load does
        for _ in range(20):
            x = torch.randn(64 << 20, device=dev); y = (x * 0.02).to(dt); del x, y
        torch.cuda.synchronize(); torch.cuda.empty_cache()
    elif order == "blame_create":   # engine object (arena, streams, events), no weights
        preset, F, H, W, T = bench.WORKLOADS["cogvideox-5b-49x480x720"]
        eng = s2v.S2VEngine(s2v.config.PRESETS[preset](), dt, dev)
    else:                           # engine + base weights, no LoRA merge
        preset, F, H, W, T = bench.WORKLOADS["cogvideox-5b-49x480x720"]
        cfg = s2v.config.PRESETS[preset]()
        eng = s2v.S2VEngine(cfg, dt, dev)
        eng.load_state_dict(s2v.weights.synthetic_state_dict(cfg, seed=1234, device=dev))
        torch.cuda.empty_cache()
    vae = make_vae()
    vae_time(vae, order, modes=(True,))
elif order in ("blame_stream", "blame_alloc", "blame_hipstream"):
    if order == "blame_stream":      # one idle torch stream (non-blocking) created before the decoder
        keep = torch.cuda.Stream()
    elif order == "blame_alloc":     # one live 11 GB allocation before the decoder
        keep = torch.empty(11 << 30, dtype=torch.uint8, device=dev)
    else:                            # a used side stream: one kernel launched on it
        keep = torch.cuda.Stream()
        with torch.cuda.stream(keep):
            z = torch.zeros(1 << 20, device=dev) + 1
        torch.cuda.synchronize()
    vae = make_vae()
    vae_time(vae, order, modes=(True,))
elif order == "tiled_only":  # workspace sets sized for the TILE window only (no untiled decode before: 5.4 GB per set instead of 21.7)
    vae = make_vae()
    vae_time(vae, "alone, tiled only", modes=(True,))
    print("sets, bytes per set:", vae.workspace_info(), flush=True)
    vae.close(); del vae; torch.cuda.empty_cache()
    eng = make_engine()
    vae = make_vae()
    vae_time(vae, "engine created before, tiled only", modes=(True,))
    vae_time(vae, "then untiled + tiled", modes=(False, True))
elif order == "engine_first_shift":  # does the decoder's time depend on how many streams were created before its own?
    eng = make_engine()
    extra = []
    for k in range(5):
        vae = make_vae()
        vae_time(vae, f"engine + {len(extra)} idle non-blocking streams before the decoder")
        vae.close()
        del vae
        torch.cuda.empty_cache()
        extra.append(torch.cuda.Stream())
else:  # streams only: a few idle non-blocking streams instead of an engine
    vae = make_vae()
    vae_time(vae, "alone")
    ss = [torch.cuda.Stream() for _ in range(3)]
    vae2 = make_vae()
    vae_time(vae2, "second VAE object after 3 extra torch streams")
