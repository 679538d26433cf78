"""Parity at the REAL depth and width of CogVideoX-5B (42 layers, D = 3072, 48 heads, rotary embedding) on a small geometry the CPU
oracle finishes in minutes: 9 frames 64 x 96 (latents 3 x 8 x 12 -> 226 + 24 + 72 = 322 tokens), 3 DDIM steps with CFG 6.
fp32 HIP path vs the fp32 oracle (the north-star bar: max-abs <= 1e-3), bf16 HIP path vs the same oracle (drift reported).
Complements tests/test_gpu_c1_end_to_end.py (2B at real depth, no rotary embedding).  Output kept in profiles/."""
import importlib, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
from oracle import sched_ref, transformer_ref as tr

DEV = "cuda:0"
STEPS, GS = 3, 6.0
F, H, W, T = 3, 8, 12, 226
cfg = s2v.cogvideox_5b()
t0 = time.time()
sd = s2v.weights.synthetic_state_dict(cfg, seed=51, parity=True)
print(f"synthetic 5B state dict: {time.time() - t0:.0f} s, {sum(v.numel() for v in sd.values()) / 1e9:.2f} G parameters", flush=True)
g = torch.Generator().manual_seed(52)
lat0 = torch.randn(1, F, 16, H, W, generator=g)
pe, ne = torch.randn(1, T, 4096, generator=g), torch.randn(1, T, 4096, generator=g)
ref = torch.randn(1, 1, 16, H, W, generator=g) * 0.7
ref_rope, rope = tr.pipeline_rope(H * 8, W * 8, F)
ocfg = dict(num_heads=cfg.num_attention_heads, num_layers=cfg.num_layers, use_rope=True, norm_eps=1e-5)
ac = sched_ref.alphas_cumprod(cfg.snr_shift_scale)
torch.set_num_threads(min(64, os.cpu_count() or 8))
t0 = time.time()
lat, per_step = lat0.clone(), []
with torch.no_grad():
    for t in sched_ref.trailing_timesteps(STEPS):
        tt = torch.tensor([int(t), int(t)])
        npred = tr.transformer_forward(sd, ocfg, torch.cat([lat, lat]), torch.cat([ne, pe]), ref, tt, rope, ref_rope).float()
        lat = sched_ref.ddim_step(ac, STEPS, sched_ref.cfg_combine(npred, GS), int(t), lat)[0]
        per_step.append(lat.clone())
print(f"CPU oracle: {STEPS} steps of 42 layers in {time.time() - t0:.0f} s", flush=True)
# the same loop with the oracle in bf16 (the reference's own bf16 arithmetic on the CPU): how much of the bf16 drift is the format
t0 = time.time()
sd16 = {k: v.bfloat16() for k, v in sd.items()}
lat, per_step16 = lat0.bfloat16(), []
with torch.no_grad():
    for t in sched_ref.trailing_timesteps(STEPS):
        tt = torch.tensor([int(t), int(t)])
        npred = tr.transformer_forward(sd16, ocfg, torch.cat([lat, lat]), torch.cat([ne, pe]).bfloat16(), ref.bfloat16(), tt, rope, ref_rope)
        lat = sched_ref.ddim_step(ac, STEPS, sched_ref.cfg_combine(npred, GS).bfloat16(), int(t), lat)[0].bfloat16()
        per_step16.append(lat.float().clone())
del sd16
print(f"CPU oracle in bf16: {time.time() - t0:.0f} s", flush=True)


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


for dt, name in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
    m = s2v.HipCogVideoXTransformer3DModel(cfg, dt, DEV)
    m.load_state_dict(sd)
    pipe = s2v.S2VPipeline(m, s2v.CogVideoXDDIMScheduler(snr_shift_scale=cfg.snr_shift_scale), None)
    got = []
    pipe(prompt_embeds=pe, negative_prompt_embeds=ne, ref_img_states=ref, height=H * 8, width=W * 8, num_frames=9,
         num_inference_steps=STEPS, guidance_scale=GS, latents=lat0, output_type="latent", return_dict=False, use_graph=True,
         callback_on_step_end=lambda p, i, t, kw: got.append(kw["latents"].float().cpu().clone()))
    torch.cuda.synchronize()
    errs = [(a - b).abs().max().item() for a, b in zip(got, per_step)]
    rels = [rel_l2(a, b) for a, b in zip(got, per_step)]
    print(f"{name}: per-step max-abs " + " ".join(f"{e:.2e}" for e in errs) + " | rel-L2 " + " ".join(f"{e:.2e}" for e in rels)
          + f" | max|ref| {per_step[-1].abs().max().item():.2f}", flush=True)
    if name == "fp32":
        assert errs[-1] <= 1e-3, errs
    else:
        o16 = [rel_l2(a, b) for a, b in zip(per_step16, per_step)]
        h16 = [rel_l2(a, b) for a, b in zip(got, per_step16)]
        print("      bf16 oracle vs fp32 oracle rel-L2 " + " ".join(f"{e:.2e}" for e in o16) + " | HIP bf16 vs bf16 oracle rel-L2 "
              + " ".join(f"{e:.2e}" for e in h16), flush=True)
        assert rels[-1] <= 2.0 * o16[-1] + 1e-2, (rels, o16)  # no worse than the format itself drifts
    del m, pipe
print("parity at 5B depth: ok")
