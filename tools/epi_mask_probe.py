"""Same-box A/B (diagnostics library, eager steps) of which epilogues run on gemm_g4 (S2V_G4_EPI_MASK bit per epilogue: 1 bias, 2 GELU, 4 gate+residual,
8 add, 16 q/k-norm; the rest on gemm_bf16_pp64).  Usage: python tools/epi_mask_probe.py <workload> <mask> [<mask> ...]
S2V_PROBE_VAR=fused_qk: where the q/k LayerNorm + rotary runs (values 1 / 3 / 0, s2v_set_fused_qk).  S2V_PROBE_VAR=S2V_G4T_EPI_MASK: the same for the persistent trickled-epilogue kernel gemm_g4t (3 = FF1 only, 19 = + the fused QKV projection)."""
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
name = sys.argv[1]
VAR = os.environ.get("S2V_PROBE_VAR", "S2V_G4_EPI_MASK")
masks = sys.argv[2:]
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
n = 6 if F > 3 else 40
outs = []
for rep in range(2):
    for mask in masks:
        if VAR == "fused_qk":   # diag.s2v_set_fused_qk: 1 product (fp8 QK^T: q/k-norm in the quantisation pass), 3 always the projection's epilogue, 0 stand-alone kernel
            diag.s2v_set_fused_qk(int(mask))
        else:
            os.environ[VAR] = mask
        lat = lat0.clone()
        for i in range(2):
            eng.denoise_step(lat, float(sch.timesteps[i]), coefs[i], use_graph=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            eng.denoise_step(lat, float(sch.timesteps[(2 + i) % 50]), coefs[(2 + i) % 50], use_graph=False)
        torch.cuda.synchronize()
        outs.append(lat.clone())
        print(f"{name} {VAR}={mask}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step (eager)", flush=True)
        if os.environ.get("S2V_PROBE_PROFILE") and rep == 1:   # per-class launch times of two more steps (HIP events around every launch)
            import ctypes
            L.check(diag.s2v_profile_enable(eng._h, 1))
            for i in range(2):
                eng.denoise_step(lat, float(sch.timesteps[i]), coefs[i], use_graph=False)
            torch.cuda.synchronize()
            ms_, cnt_ = (ctypes.c_float * 8)(), (ctypes.c_int32 * 8)()
            L.check(diag.s2v_profile_read(eng._h, ms_, cnt_, 8))
            L.check(diag.s2v_profile_enable(eng._h, 0))
            print("   per class:", {c: round(ms_[k] / cnt_[k], 4) for k, c in enumerate(bench.CLASSES) if cnt_[k]}, flush=True)
print("all latents bit-identical:", all(torch.equal(outs[0], o) for o in outs))
