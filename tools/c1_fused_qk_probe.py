"""Experiment (diagnostics library): the C1 step (2B, 9 x 256 x 256) with the q/k LayerNorm + rotary embedding fused into the QKV epilogue
(product) against the stand-alone pass -- at one round of tiles per GEMM the fused epilogue is fully exposed."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
import bench
L = s2v._lib
diag = L.diag_lib()
L.lib()
L._apply_sigs(diag, L._SIGS)  # the engine runs on the diagnostics library in this process: bind every signature
L._lib = diag
dev, dt = "cuda:0", torch.bfloat16
name = sys.argv[1] if len(sys.argv) > 1 else "cogvideox-2b-9x256x256"
preset, F, H, W, T = bench.WORKLOADS[name]
cfg = s2v.config.PRESETS[preset]()
eng = s2v.S2VEngine(cfg, dt, dev)
bench.load_synthetic(s2v, eng, cfg, 1234)
g = torch.Generator(device=dev).manual_seed(100)
text = torch.randn(2, T, cfg.text_embed_dim, generator=g, device=dev)
ref = torch.randn(1, 1, cfg.in_channels, H, W, generator=g, device=dev) * 0.7
lat0 = torch.randn(1, F, cfg.in_channels, H, W, generator=g, device=dev).to(dt).contiguous()
eng.set_geometry(2, T, F, H, W); eng.prepare_tables(H * 8, W * 8); eng.set_conditioning(text, ref)
sch = s2v.CogVideoXDDIMScheduler(snr_shift_scale=cfg.snr_shift_scale); sch.set_timesteps(50)
coefs = [sch.coef(t, dt, 6.0) for t in sch.timesteps]
outs = {}
for fused in (1, 0, 1, 0):
    diag.s2v_set_fused_qk(fused)
    lat = lat0.clone()
    for i in range(5):
        eng.denoise_step(lat, float(sch.timesteps[i]), coefs[i], use_graph=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = int(os.environ.get("N_STEPS", "40"))
    for i in range(n):
        eng.denoise_step(lat, float(sch.timesteps[(5 + i) % 50]), coefs[(5 + i) % 50], use_graph=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    outs.setdefault(fused, lat.clone())
    print(f"{name} fused_qk={fused}: {ms:.3f} ms/step (eager)", flush=True)
print("fused and stand-alone latents bit-identical:", torch.equal(outs[0], outs[1]))
