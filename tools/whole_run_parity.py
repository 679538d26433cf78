"""Whole-run deviation of the low-precision engines AT THE HEADLINE CONFIGURATION, against an oracle-pinned on-GPU reference.

What is compared: the latents after every step of the denoise loop (custom_cogvideox_pipe.py:237-311: CFG-pair transformer forward,
CFG, DDIM step, round to the model dtype) of CogVideoX-5B at real depth and width (42 layers, D = 3072) and the real token count
(49 x 480 x 720 -> N = 19 126; `--geometry c5`: 49 x 720 x 1280 -> N = 50 626), for >= 10 steps of a 50-step trailing DDIM schedule
(`--steps`, `--schedule`).

Reference: the SAME engine in the fp32 model dtype.  fp32 is the CPU-reference-parity mode: pinned to the reference's goldens at
<= 6e-7 per forward / <= 1e-5 per pipeline (tests/test_gpu_parity.py), one block at these 19 126 tokens against the CPU oracle at
<= 1e-3 (tests/test_gpu_fullsize_oracle.py [5b-f32-mfma]), its GEMMs bit-identical to the VALU kernels (tests/test_gpu_f32m.py).
Since round 5 it runs on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32), seconds per step instead of minutes, which is what makes
this comparison affordable.  Both engines hold the same (bf16-representable) weights and start from the same bf16-representable
latents / embeddings, so the figures are ARITHMETIC deviation: storage rounding of activations and latents, bf16 / fp16 / fp8
products, the exp2-domain softmax.

Output: one line per (format, step) with max-abs and relative L2 of the latents, and a summary line per format.
    python tools/whole_run_parity.py [--steps 10] [--schedule 50] [--geometry c3|c5|small] [--formats bf16,bf16-p16,fp8,fp8-qk]
"""
import argparse
import copy
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
DEV = "cuda:0"
GEOMETRY = {"c3": (13, 60, 90, 226), "c5": (13, 90, 160, 226), "small": (3, 8, 12, 226), "c1": (3, 32, 32, 226)}
#            label       weight_format  attn_p_format
FORMATS = {"bf16": (None, "bf16"), "bf16-p16": (None, "f16"), "fp8": ("fp8", "bf16"), "fp8-qk": ("fp8-qk", "bf16"), "fp8-qk-p16": ("fp8-qk", "f16"),
           "f16": (None, "bf16")}  # "f16": the fp16 MODEL dtype (inference.py:191: every non-5B checkpoint), latents stored in fp16 between steps


def _run(s2v, cfg, sd, dt, geometry, steps, schedule, inputs, sink, guidance=6.0, use_graph=True, round_latents=False):
    F, H, W, T = geometry
    lat0, pe, ne, ref = inputs
    m = s2v.HipCogVideoXTransformer3DModel(cfg, dt, DEV)
    m.load_state_dict(sd)
    pipe = s2v.S2VPipeline(m, s2v.CogVideoXDDIMScheduler(snr_shift_scale=cfg.snr_shift_scale), None)

    def cb(p, i, t, kw):
        x = kw["latents"]
        if round_latents:  # fp32 arithmetic, but the latents stored in bf16 between steps as the bf16 pipeline stores them (custom_cogvideox_pipe.py:296)
            x = x.bfloat16().to(x.dtype)
        sink(i, x)
        if i + 1 >= steps:
            p.interrupt = True
        return {"latents": x} if round_latents else {}

    t0 = time.time()
    pipe(prompt_embeds=pe, negative_prompt_embeds=ne, ref_img_states=ref, height=H * 8, width=W * 8, num_frames=(F - 1) * 4 + 1,
         num_inference_steps=schedule, guidance_scale=guidance, latents=lat0.clone(), output_type="latent", return_dict=False,
         use_graph=use_graph, callback_on_step_end=cb)
    torch.cuda.synchronize()
    dt_s = time.time() - t0
    m.engine.close()
    del pipe, m
    torch.cuda.empty_cache()
    return dt_s


def whole_run(s2v, preset="cogvideox_5b", geometry="c3", steps=10, schedule=50, formats=("bf16", "bf16-p16", "fp8", "fp8-qk"), seed=71,
              layers=None, log=print, arith_ref=True):
    """returns {format: [(step, max_abs, rel_l2, ref_max_abs, arith_max_abs, arith_rel_l2), ...]} and the wall-clock of every run.
    max_abs / rel_l2: against the pure fp32 run; arith_*: against the fp32 run whose latents are rounded to bf16 after every step
    (the bf16 pipeline's own storage rounding taken out: what is left is the transformer arithmetic)."""
    cfg0 = getattr(s2v, preset)()
    if layers:
        cfg0.num_layers = layers
    geo = GEOMETRY[geometry]
    F, H, W, T = geo
    sd = s2v.weights.synthetic_state_dict(cfg0, seed=seed, device=DEV, parity=True)
    sd = {k: v.bfloat16().float() for k, v in sd.items()}  # both engines hold exactly these values
    g = torch.Generator(device=DEV).manual_seed(seed + 1)
    lat0 = torch.randn(1, F, 16, H, W, generator=g, device=DEV).bfloat16().float()
    pe, ne = (torch.randn(1, T, cfg0.text_embed_dim, generator=g, device=DEV).bfloat16().float() for _ in range(2))
    ref = (torch.randn(1, 1, 16, H, W, generator=g, device=DEV) * 0.7).bfloat16().float()
    inputs = (lat0, pe, ne, ref)

    ref_lat, ref_rl = {}, {}
    c = copy.copy(cfg0)
    c.attn_p_format = "bf16"
    secs = {"f32": _run(s2v, c, sd, torch.float32, geo, steps, schedule, inputs, lambda i, x: ref_lat.__setitem__(i, x.float().cpu()), use_graph=False)}
    log(f"fp32 reference run: {steps} steps of a {schedule}-step DDIM schedule, {T + (F + 1) * (H // 2) * (W // 2)} tokens, {cfg0.num_layers} layers: "
        f"{secs['f32']:.1f} s wall-clock (incl. weight load)")
    if arith_ref:
        secs["f32-bf16lat"] = _run(s2v, c, sd, torch.float32, geo, steps, schedule, inputs, lambda i, x: ref_rl.__setitem__(i, x.float().cpu()), use_graph=False,
                                   round_latents=True)
        for i in sorted(ref_rl):
            a, b = ref_rl[i].double(), ref_lat[i].double()
            log(f"{'f32-bf16lat':10s} step {i + 1:2d}: latents max-abs {(a - b).abs().max().item():.3e} (max|ref| {b.abs().max().item():.2f})  rel-L2 {((a - b).norm() / b.norm()).item():.3e}"
                "   <- fp32 arithmetic, latents stored in bf16 between steps: the storage share of every figure below")
    res = {}
    for name in formats:
        wf, pf = FORMATS[name]
        c = copy.copy(cfg0)
        c.weight_format, c.attn_p_format = wf, pf
        rows = []

        def sink(i, x, rows=rows):
            a, b = x.float().cpu().double(), ref_lat[i].double()
            row = [i, (a - b).abs().max().item(), ((a - b).norm() / b.norm()).item(), b.abs().max().item(), float("nan"), float("nan")]
            if arith_ref:
                b2 = ref_rl[i].double()
                row[4], row[5] = (a - b2).abs().max().item(), ((a - b2).norm() / b2.norm()).item()
            rows.append(tuple(row))

        secs[name] = _run(s2v, c, sd, torch.float16 if name == "f16" else torch.bfloat16, geo, steps, schedule, inputs, sink)
        res[name] = rows
        for (i, ma, rl, rm, ama, arl) in rows:
            log(f"{name:10s} step {i + 1:2d}: latents max-abs {ma:.3e} (max|ref| {rm:.2f})  rel-L2 {rl:.3e}   | vs f32-bf16lat: max-abs {ama:.3e}  rel-L2 {arl:.3e}")
        log(f"{name:10s} SUMMARY over {len(rows)} steps: worst max-abs {max(r[1] for r in rows):.3e}, worst rel-L2 {max(r[2] for r in rows):.3e}, "
            f"final rel-L2 {rows[-1][2]:.3e}   | vs f32-bf16lat: worst max-abs {max(r[4] for r in rows):.3e}, worst rel-L2 {max(r[5] for r in rows):.3e}, "
            f"final rel-L2 {rows[-1][5]:.3e}  ({secs[name]:.1f} s)")
    return res, secs


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--schedule", type=int, default=50)
    ap.add_argument("--geometry", default="c3", choices=sorted(GEOMETRY))
    ap.add_argument("--preset", default="cogvideox_5b")
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--formats", default="bf16,bf16-p16,fp8,fp8-qk")
    ap.add_argument("--no-arith-ref", action="store_true", help="skip the second fp32 run (latents rounded to bf16 between steps)")
    a = ap.parse_args()
    s2v = importlib.import_module("disentangled-subject-to-vid_amd")
    print(f"# tools/whole_run_parity.py --steps {a.steps} --schedule {a.schedule} --geometry {a.geometry} --preset {a.preset} --formats {a.formats}"
          + (f" --layers {a.layers}" if a.layers else ""), flush=True)
    whole_run(s2v, a.preset, a.geometry, a.steps, a.schedule, tuple(a.formats.split(",")), layers=a.layers or None, log=lambda s: print(s, flush=True),
              arith_ref=not a.no_arith_ref)
