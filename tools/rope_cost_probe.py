import importlib, os, sys, time, torch
sys.path.insert(0, "/root/repo")
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
DEV = "cuda:0"
F, H, W, T = 13, 60, 90, 226
def run(use_rope):
    cfg = s2v.cogvideox_5b(); cfg.num_layers = 6; cfg.use_rotary_positional_embeddings = use_rope
    sd = s2v.weights.synthetic_state_dict(cfg, seed=1, device=DEV)
    e = s2v.S2VEngine(cfg, torch.bfloat16, DEV); e.load_state_dict(sd); del sd
    e.set_geometry(2, T, F, H, W); e.prepare_tables(480, 720)
    g = torch.Generator(device=DEV).manual_seed(2)
    e.set_conditioning(torch.randn(2, T, 4096, generator=g, device=DEV), torch.randn(1, 1, 16, H, W, generator=g, device=DEV))
    lat = torch.randn(1, F, 16, H, W, device=DEV).bfloat16(); ts = torch.tensor([500.0, 500.0])
    for _ in range(2): e.forward(lat, ts, shared_latent=True)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5): e.forward(lat, ts, shared_latent=True)
    torch.cuda.synchronize(); return (time.time() - t0) / 5 * 1e3
a = run(True); b = run(False); a2 = run(True); b2 = run(False)
print(f"6 layers: rope {a:.2f} / {a2:.2f} ms, no rope {b:.2f} / {b2:.2f} ms -> table cost {(a + a2 - b - b2) / 2 / 6:.3f} ms per layer")
