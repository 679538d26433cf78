"""fp8 (e4m3 W8A8) linears vs bf16 linears at the real depth / width of CogVideoX-5B (42 layers), small geometry (322 tokens) and the
C3 token count (2 layers are not enough to see depth effects; 42 layers at 19126 tokens take ~1.5 s per forward): relative L2 of
the noise prediction of ONE forward and of the latents after 3 DDIM steps; the same for "fp8-qk" (weight_format 2: additionally MX e4m3 q / k and
QK^T on the scaled fp8 MFMA).  There is no reference for these paths (parity unpinned)."""
import copy, importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
DEV = "cuda:0"
cfg = s2v.cogvideox_5b()
sd = s2v.weights.synthetic_state_dict(cfg, seed=51, device=DEV, parity=True)


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


GEOS = ((3, 8, 12, 226, "322 tokens"), (13, 60, 90, 226, "19126 tokens"))
if "c5" in sys.argv:   # configs[4] geometry: one forward and three steps only (a 42-layer forward at 50626 tokens takes ~2.5 s)
    GEOS = ((13, 90, 160, 226, "50626 tokens"),)
for (F, H, W, T, tag) in GEOS:
    g = torch.Generator(device=DEV).manual_seed(52)
    lat0 = torch.randn(1, F, 16, H, W, generator=g, device=DEV).bfloat16()
    pe, ne = (torch.randn(1, T, 4096, generator=g, device=DEV).bfloat16() for _ in range(2))
    ref = (torch.randn(1, 1, 16, H, W, generator=g, device=DEV) * 0.7).bfloat16()
    res = {}
    for fmt in ("bf16", "fp8", "fp8-qk"):
        c = copy.copy(cfg)
        c.weight_format = None if fmt == "bf16" else fmt
        m = s2v.HipCogVideoXTransformer3DModel(c, torch.bfloat16, DEV)
        m.load_state_dict(sd)
        eng = m.engine
        eng.set_geometry(2, T, F, H, W)
        eng.prepare_tables(H * 8, W * 8)
        eng.set_conditioning(torch.cat([ne, pe]), ref)
        npred = eng.forward(lat0, torch.tensor([999.0, 999.0]), shared_latent=True).float().clone()
        pipe = s2v.S2VPipeline(m, s2v.CogVideoXDDIMScheduler(snr_shift_scale=cfg.snr_shift_scale), None)
        lat = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, ref_img_states=ref, height=H * 8, width=W * 8, num_frames=(F - 1) * 4 + 1,
                   num_inference_steps=3, guidance_scale=6.0, latents=lat0.clone(), output_type="latent", return_dict=False, use_graph=True)[0]
        torch.cuda.synchronize()
        long_steps = 50 if F == 3 else (10 if H == 60 else 3)   # a full 50-step run at 322 tokens, 10 of 50 at 19126, 3 at 50626
        sch = s2v.CogVideoXDDIMScheduler(snr_shift_scale=cfg.snr_shift_scale)
        lat50 = pipe.__class__(m, sch, None)(prompt_embeds=pe, negative_prompt_embeds=ne, ref_img_states=ref, height=H * 8, width=W * 8,
                                            num_frames=(F - 1) * 4 + 1, num_inference_steps=long_steps, guidance_scale=6.0, latents=lat0.clone(),
                                            output_type="latent", return_dict=False, use_graph=True)[0]
        torch.cuda.synchronize()
        res[fmt] = (npred, lat.float().clone(), lat50.float().clone(), long_steps)
        del m, pipe, eng
    for fmt in ("fp8", "fp8-qk"):
        print(f"{tag}: {fmt} vs bf16 rel-L2  one forward (42 layers) {rel(res[fmt][0], res['bf16'][0]):.3e}   latents after 3 DDIM steps {rel(res[fmt][1], res['bf16'][1]):.3e}"
              f"   after a {res[fmt][3]}-step DDIM run {rel(res[fmt][2], res['bf16'][2]):.3e}", flush=True)
