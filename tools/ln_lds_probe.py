"""A/B of the LayerNorm + modulate pass: parameters in registers per row (ln_modulate_k, S2V_LN_LDS_ROWS=0) against parameters staged in LDS per
workgroup of sixteen rows (ln_modulate_lds_k, the default from 16384 rows): sha256 of one forward's output (5B width, 2 layers, 19126 tokens; bf16 and
fp8 engines) -- the two must agree bit for bit -- and the pass's time from the engine's per-kernel profile.  Run once per setting:
    export S2V_LIB=$PWD/disentangled-subject-to-vid_amd/libs2v_hip_diag.so   (the knob exists in the diagnostics library only)
    S2V_LN_LDS_ROWS=0 python tools/ln_lds_probe.py; python tools/ln_lds_probe.py"""
import copy, hashlib, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
DEV = "cuda:0"
cfg0 = s2v.cogvideox_5b()
cfg0.num_layers = 2
sd = s2v.weights.synthetic_state_dict(cfg0, seed=3, device=DEV, parity=True)
g = torch.Generator(device=DEV).manual_seed(4)
F, H, W, T = 13, 60, 90, 226
t1 = torch.randn(1, T, 4096, generator=g, device=DEV)
text = torch.cat([t1, t1 * 0.5])
ref = torch.randn(1, 1, 16, H, W, generator=g, device=DEV) * 0.7
lat = torch.randn(1, F, 16, H, W, generator=g, device=DEV).bfloat16()
for fmt in (None, "fp8"):
    cfg = copy.copy(cfg0)
    cfg.weight_format = fmt
    m = s2v.HipCogVideoXTransformer3DModel(cfg, torch.bfloat16, DEV)
    m.load_state_dict(sd)
    eng = m.engine
    eng.set_geometry(2, T, F, H, W)
    eng.prepare_tables(H * 8, W * 8)
    eng.set_conditioning(text, ref)
    y = eng.forward(lat, torch.tensor([500.0, 500.0]), shared_latent=True)
    torch.cuda.synchronize()
    h = hashlib.sha256(y.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16]
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    L = s2v._lib
    L.check(L.lib().s2v_profile_enable(eng._h, 1))
    for _ in range(3):
        eng.forward(lat, torch.tensor([500.0, 500.0]), shared_latent=True)
    torch.cuda.synchronize()
    import ctypes
    ms = (ctypes.c_float * 8)(); cnt = (ctypes.c_int32 * 8)()
    L.check(L.lib().s2v_profile_read(eng._h, ms, cnt, 8))
    per = {i: (ms[i] / max(cnt[i], 1), cnt[i]) for i in range(8) if cnt[i]}   # class 5 = LayerNorm + modulate
    print(f"LN_LDS_ROWS={os.environ.get('S2V_LN_LDS_ROWS', 'default')} {fmt or 'bf16'}: sha256 {h}  per-class avg ms " + " ".join(f"[{i}] {v[0]:.4f}x{v[1]}" for i, v in per.items()), flush=True)
    del m, eng
