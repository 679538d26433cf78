"""Same-box A/B of the C1 step (2B, 9 x 256 x 256, eager launches, diagnostics library): attention on attn_q4 (variant 6) against the product's
choice for short sequences, attn_pp (variant 0)."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
import bench
L = s2v._lib
diag = L.diag_lib()
L.lib()
L._apply_sigs(diag, L._SIGS)
L._lib = diag
dev, dt = "cuda:0", torch.bfloat16
preset, F, H, W, T = bench.WORKLOADS["cogvideox-2b-9x256x256"]
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
for variant in (6, 0, 6, 0):
    diag.s2v_set_attn_variant(variant)
    lat = lat0.clone()
    for i in range(5):
        eng.denoise_step(lat, float(sch.timesteps[i]), coefs[i], use_graph=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    for i in range(n):
        eng.denoise_step(lat, float(sch.timesteps[(5 + i) % 50]), coefs[(5 + i) % 50], use_graph=False)
    torch.cuda.synchronize()
    print(f"attention variant {variant} ({'attn_q4' if variant == 6 else 'product: attn_pp'}): {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step (eager)", flush=True)
diag.s2v_set_attn_variant(0)
# the FF1 (bias + GELU epilogue, 300 tiles of 30 K-tiles) on gemm_g4 (mask 31) against the eight-wave gemm_bf16_pp64 (mask 29: g4 for every epilogue but GELU)
for mask in ("31", "29", "31", "29"):
    os.environ["S2V_G4_EPI_MASK"] = mask
    lat = lat0.clone()
    for i in range(5):
        eng.denoise_step(lat, float(sch.timesteps[i]), coefs[i], use_graph=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    for i in range(n):
        eng.denoise_step(lat, float(sch.timesteps[(5 + i) % 50]), coefs[(5 + i) % 50], use_graph=False)
    torch.cuda.synchronize()
    print(f"S2V_G4_EPI_MASK={mask}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step (eager)", flush=True)
