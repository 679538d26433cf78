"""Benchmark of the denoise hot path on MI355X.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N --steps K --warmup W          (no wrapper: spawns its own N ranks, one per GPU, backend nccl = RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             (the driver's form: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env)

    python bench.py --batch 1                              (ONE sample of the CFG pair per step: the half step a GPU of a CFG-parallel pair runs)
    python bench.py --gpus 2 --cfg-parallel                (one video on two GPUs: dist.CfgPair; N even: N / 2 videos in flight)

--gpus N is a CLAIM the run has to earn: with no torchrun environment the script launches N rank processes itself and refuses
(non-zero exit) when fewer than N GPUs are visible; with a torchrun environment WORLD_SIZE must equal N.  `value` is computed from
what the ranks report (sum of their timed steps / max elapsed over ranks), never from the flag.

One "step" = one iteration of the denoise loop of src/custom_cogvideox_pipe.py:241-296 on one video: the B=2 (CFG pair)
transformer forward + fp32 CFG + DDIM scheduler step + round to bf16, all inside libs2v_hip.so, inputs resident in
HBM; the timed region replays it from ONE captured hipGraph (--graph 1, default) and carries no profiling; the eager time and
the per-kernel HIP-event durations come from separate passes after it.  Workload (BASELINE.json metric): CogVideoX-5B, 49 frames 480x720 -> latents 13x60x90, N = 226+1350+17550 tokens,
bf16, synthetic seeded weights (no checkpoints offline).  N > 1: independent replicas, one prompt per GPU, weights
broadcast once rank0 -> all over RCCL before the timed region; value = total steps/s.
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3    # fp32-input MFMA (v_mfma_f32_32x32x2_f32) = the fp32 vector rate, same guide
PEAK_FP8_TFLOPS = 5000.0   # dense fp8 MFMA peak (block-scaled f8f6f4 instructions), same guide
PEAK_HBM_GBS = 8000.0     # HBM3E spec peak; ~6300 GB/s is what a float4 copy reaches (same guide)
PEAK_CLOCK_MHZ = 2400.0   # the shader clock the datasheet peaks are quoted at (same guide: 157.3 TF fp32 = 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz)
CLASSES = ["gemm_qkv", "attention", "gemm_out", "gemm_ff1_gelu", "gemm_ff2", "ln_modulate", "qknorm_rope_vt", "mod_gemv"]

WORKLOADS = {
    # name: (preset, latent F, H, W, text tokens)
    "cogvideox-5b-49x480x720": ("cogvideox-5b", 13, 60, 90, 226),
    "cogvideox-2b-49x480x720": ("cogvideox-2b", 13, 60, 90, 226),
    "cogvideox-2b-9x256x256": ("cogvideox-2b", 3, 32, 32, 226),
    # the geometry of BASELINE configs[4] (49 frames 720x1280, N = 50626 tokens): bf16, and with W8A8 fp8 linears (configs[4] itself)
    "cogvideox-5b-49x720x1280": ("cogvideox-5b", 13, 90, 160, 226),
    "cogvideox-5b-fp8-49x720x1280": ("cogvideox-5b-fp8", 13, 90, 160, 226),
    "cogvideox-5b-fp8-49x480x720": ("cogvideox-5b-fp8", 13, 60, 90, 226),
    # the opt-in throughput preset BEYOND configs[4]'s "fp8 weights" (config.cogvideox_5b_fp8_auto): fp8 QK^T from 40 000 tokens on + fp16 P
    "cogvideox-5b-fp8auto-49x720x1280": ("cogvideox-5b-fp8-auto", 13, 90, 160, 226),
    "cogvideox-5b-fp8lin-49x720x1280": ("cogvideox-5b-fp8lin", 13, 90, 160, 226),   # round-5 name of cogvideox-5b-fp8-49x720x1280
    # an option BEYOND configs[4]'s "fp8 weights": additionally q / k as MX e4m3 and QK^T on the scaled fp8 MFMA (weight_format 2)
    "cogvideox-5b-fp8qk-49x720x1280": ("cogvideox-5b-fp8qk", 13, 90, 160, 226),
    "cogvideox-5b-fp8qk-49x480x720": ("cogvideox-5b-fp8qk", 13, 60, 90, 226),
}


def load_synthetic(s2v, eng, cfg, seed, lora_rank=128, parity=False):
    """near-identity N(0, 0.02^2) weights generated tensor by tensor on the GPU (timing weights); parity=True: the larger-variance
    weights / non-zero biases / LayerNorm affines of weights.synthetic_state_dict(parity=True) (what the parity tests use: sharper attention)"""
    shapes = s2v.weights.state_dict_shapes(cfg)
    gen = torch.Generator(device=eng.device).manual_seed(seed)
    for k, shp in shapes.items():
        is_norm = ".norm" in k or k.startswith("norm_final") or "norm_q" in k or "norm_k" in k
        if len(shp) >= 2:
            std = 0.02
            if parity:
                fan_in = 1
                for d in shp[1:]:
                    fan_in *= d
                std = (0.5 if ".linear." in k else 0.7) / fan_in ** 0.5
            t = torch.randn(shp, generator=gen, device=eng.device, dtype=torch.float32) * std
        elif k.endswith("weight") and is_norm and ".linear." not in k:
            t = torch.ones(shp, device=eng.device)
            if parity:
                t = t + 0.2 * torch.randn(shp, generator=gen, device=eng.device)
        else:
            t = torch.zeros(shp, device=eng.device)
            if parity:
                t = 0.1 * torch.randn(shp, generator=gen, device=eng.device)
        eng.load_weight(k, t)
        eng._keep.clear()
        del t
    if parity:
        s2v._lib.check(s2v.lib().s2v_finalize_weights(eng._h, s2v._lib.stream_ptr()))
        torch.cuda.synchronize()
        return 0
    # BASELINE configs[2] is "5B + subject-LoRA merged": a synthetic rank-128 adapter on every target of the reference's LoRA
    # (src/inference.py:218-229: alpha / r = 64 / 128) is merged as W + 0.5 B A before the weights are finalised (and, for the fp8
    # workloads, quantised) -- shapes and therefore timing are those of the merged model, as the reference runs it
    n_lora = 0
    for k in s2v.weights.lora_target_keys(cfg):
        shp = shapes[k]
        A = (torch.randn((lora_rank,) + tuple(shp[1:]), generator=gen, device=eng.device) * 0.02).reshape(lora_rank, -1).contiguous()
        Bm = (torch.randn((shp[0], lora_rank), generator=gen, device=eng.device) * 0.02).contiguous()
        s2v._lib.check(s2v.lib().s2v_merge_lora(eng._h, k.encode(), s2v._lib.ptr(A), s2v._lib.ptr(Bm), lora_rank, 0.5, s2v._lib.stream_ptr()))
        torch.cuda.synchronize()
        n_lora += 1
    s2v._lib.check(s2v.lib().s2v_finalize_weights(eng._h, s2v._lib.stream_ptr()))
    torch.cuda.synchronize()
    return n_lora


def best_cpu_threads():
    """the thread count at which torch's fp32 GEMM is fastest on this host: the GPU boxes show 256 CPUs to a container that is
    granted far fewer (measured: 1559 / 1397 / 1283 / 1111 / 730 GFLOP/s at 16 / 32 / 64 / 128 / 256 threads), and a baseline timed on
    256 spinning threads would flatter the GPU"""
    n = os.cpu_count() or 1
    a, b = torch.randn(2048, 2048), torch.randn(2048, 2048)
    best, best_t = 1, None
    for nt in sorted({min(n, c) for c in (8, 16, 32, 64, n)}):
        torch.set_num_threads(nt)
        torch.mm(a, b)
        t0 = time.time()
        for _ in range(3):
            torch.mm(a, b)
        dt = time.time() - t0
        if best_t is None or dt < 0.95 * best_t:
            best, best_t = nt, dt
    return best


FULL_STEP_MAX_TOKENS = 2048   # "full runs only for C1" (1250 tokens, ~15 s per step on 8 cores); larger geometries time one block


def cpu_baseline(s2v, cfg, F, H, W, T, dev, dt=torch.bfloat16):
    """the oracle (CPU restatement, torch fp32, on the thread count best_cpu_threads() picks) timed beside EVERY workload:
      * configs[0]'s geometry (N <= FULL_STEP_MAX_TOKENS): the FULL denoise step -- all layers, the CFG pair, CFG + DDIM step -- and its latents are
        compared with s2v_denoise_step in the line's model dtype on the same weights (fp32 line: un-rounded fp32 weights, the CPU-reference-parity mode);
      * every other geometry: ONE transformer block for ONE of the two CFG samples at the full token count, extrapolated x2 x num_layers, its
        output compared with s2v_block_forward in the line's dtype / weight format on the same (bf16- / fp16-rounded) weights and inputs.
    fp8 engines: the oracle holds the bf16-rounded weights -- there is no fp8 reference arithmetic, the comparison is labelled unpinned.
    The oracle is the checker here, never the thing shipped."""
    import copy

    from oracle import sched_ref, transformer_ref as tr

    cores = best_cpu_threads()
    torch.set_num_threads(cores)
    R = (H // 2) * (W // 2)
    V = F * R
    N = T + R + V
    D, heads = cfg.inner_dim, cfg.num_attention_heads
    fp8 = cfg.weight_format is not None
    rnd = (lambda x: x.float()) if dt == torch.float32 else (lambda x: x.to(dt).float())
    blk_flop = 2 * N * D * 12 * D + 4 * N * N * D  # one block, one sample: QKV + out + FF1 + FF2 linears, QK^T + PV
    vae_leg = cpu_baseline_vae(s2v, dev, cores, vae_scaling=cfg.vae_scaling_factor, F=F, H=H, W=W)
    rope = ref_rope = None
    if cfg.use_rotary_positional_embeddings:
        ref_rope, rope = tr.pipeline_rope(H * 8, W * 8, F)
    dname = {torch.float32: "fp32", torch.bfloat16: "bf16", torch.float16: "fp16"}[dt]
    if N <= FULL_STEP_MAX_TOKENS:
        sd = {k: rnd(v) for k, v in s2v.weights.synthetic_state_dict(cfg, seed=21, parity=True).items()}
        g = torch.Generator().manual_seed(22)
        lat = rnd(torch.randn(1, F, cfg.in_channels, H, W, generator=g))
        text = rnd(torch.randn(2, T, cfg.text_embed_dim, generator=g))
        ref = rnd(torch.randn(1, 1, cfg.in_channels, H, W, generator=g) * 0.7)
        ocfg = dict(num_heads=heads, num_layers=cfg.num_layers, use_rope=cfg.use_rotary_positional_embeddings, norm_eps=cfg.norm_eps,
                    spatial_scale=cfg.spatial_interpolation_scale, temporal_scale=cfg.temporal_interpolation_scale)
        sch = s2v.CogVideoXDDIMScheduler(snr_shift_scale=cfg.snr_shift_scale)
        sch.set_timesteps(10)
        t = sch.timesteps[1]
        with torch.no_grad():
            t0 = time.time()
            npred = tr.transformer_forward(sd, ocfg, torch.cat([lat] * 2), text, ref, torch.tensor([int(t), int(t)]), rope, ref_rope)
            v = sched_ref.cfg_combine(npred, 6.0)
            exp, _ = sched_ref.ddim_step(sched_ref.alphas_cumprod(cfg.snr_shift_scale), 10, v, int(t), lat)
            dts = time.time() - t0
        m = s2v.HipCogVideoXTransformer3DModel(cfg, dt, dev)
        m.load_state_dict(sd)
        eng = m.engine
        eng.set_geometry(2, T, F, H, W)
        eng.prepare_tables(H * 8, W * 8)
        eng.set_conditioning(text, ref)
        x = lat.to(dev, dt).contiguous().clone()
        eng.denoise_step(x, float(t), sch.coef(t, dt, 6.0))
        torch.cuda.synchronize()
        y, e = x.float().cpu().flatten(), exp.float().flatten()
        eng.close()
        step_s = dts
        sample = (f"the FULL denoise step (all {cfg.num_layers} layers, the CFG pair of {N} tokens each, CFG + DDIM step) through oracle.transformer_ref / "
                  f"sched_ref, torch fp32 on {cores} threads: {dts:.1f} s, nothing extrapolated")
        note = f"the same step through s2v_denoise_step in {dname} on the same weights / inputs: deviation of the latents after the step"
        gflops = 2 * cfg.num_layers * blk_flop / dts / 1e9
    else:
        c1 = copy.copy(cfg)
        c1.num_layers = 1
        sd = {k: rnd(v) for k, v in s2v.weights.synthetic_state_dict(c1, seed=21, parity=True).items()}
        g = torch.Generator().manual_seed(22)
        h, e0, e1 = (rnd(torch.randn(1, n, D, generator=g)) for n in (V, T, R))
        temb = rnd(torch.randn(1, c1.time_embed_dim, generator=g))
        with torch.no_grad():
            t0 = time.time()
            exp = tr.block_forward(sd, "transformer_blocks.0.", heads, h, e0, e1, temb, rope, ref_rope)
            dts = time.time() - t0
        m = s2v.HipCogVideoXTransformer3DModel(c1, dt, dev)
        m.load_state_dict(sd)
        kw = {}
        if rope is not None:
            kw = dict(image_rotary_emb=tuple(x.to(dev) for x in rope), ref_image_rotary_emb=tuple(x.to(dev) for x in ref_rope))
        got = m.transformer_blocks[0](hidden_states=h.to(dev, dt), encoder_hidden_states=e0.to(dev, dt), temb=temb.to(dev, dt),
                                      enc_hidden_states1=e1.to(dev, dt), embed_ref_img=True, ref_img_seq_start=T,
                                      ref_img_seq_end=T + R, position_delta=0, timestep=None, layer=0, **kw)
        torch.cuda.synchronize()
        y, e = torch.cat([x.float().cpu().flatten() for x in got]), torch.cat([x.flatten() for x in exp])
        m.engine.close()
        step_s = dts * 2 * cfg.num_layers
        sample = (f"1 of {cfg.num_layers} transformer blocks x 1 of 2 CFG samples at the full token count ({N}), torch "
                  f"fp32 on {cores} threads: {dts:.1f} s, extrapolated x{2 * cfg.num_layers}")
        note = (f"the same block through s2v_block_forward in {dname}" + (f" with weight_format {cfg.weight_format!r}" if fp8 else "")
                + f" on the same {dname}-rounded weights / inputs")
        gflops = blk_flop / dts / 1e9
    out = {"value": 1.0 / step_s, "unit": "steps/s", "cores": cores, "kind": "port", "sample": sample,
           "gflops": round(gflops, 1), "vae_decode": vae_leg,
           "max_abs_vs_hip": float(f"{(y - e).abs().max().item():.3e}"), "rel_l2_vs_hip": float(f"{((y - e).norm() / e.norm()).item():.3e}"),
           "max_abs_ref": round(e.abs().max().item(), 3), "vs_hip_note": note}
    if fp8:
        out["parity"] = "unpinned: the reference has no fp8 arithmetic (SURVEY 8c-7); the oracle computes the bf16 model the fp8 engine quantises"
    return out


def cpu_baseline_vae(s2v, dev, cores, vae_scaling=0.7, F=13, H=60, W=90):
    """the second half of the metric (wall-clock per video) on the host cores: ONE frame batch (2 latent frames -> 8 frames) of a
    12 x 16 latent window (96 x 128 pixels) on the threads of the transformer leg, at most 16 -- with all 256 host threads torch's conv3d
    on windows of this size collapses to 10 GFLOP/s (87 s for a 6 x 8 window that eight threads finish in 0.5 s) --
    of the real-width decoder through oracle.vae_ref, extrapolated by area to a tile and then to the tiled decode of 13 x 60 x 90 latents
    (9 tiles x [one 3-frame + five 2-frame batches] = 58.5 tile-batches; autoencoder_kl_cogvideox.py:1237-1245, 1400-1406); the same
    window through s2v_vae_decode is the check"""
    from oracle import vae_ref

    vcfg = s2v.VAEConfig(scaling_factor=vae_scaling)
    cfgd = dict(block_out_channels=tuple(vcfg.block_out_channels), layers_per_block=vcfg.layers_per_block, norm_num_groups=vcfg.norm_num_groups,
                latent_channels=vcfg.latent_channels, out_channels=vcfg.out_channels, temporal_compression_ratio=vcfg.temporal_compression_ratio,
                sample_height=vcfg.sample_height, sample_width=vcfg.sample_width, scaling_factor=vcfg.scaling_factor)
    dt = torch.bfloat16
    sd = {k: v.to(dt).float() for k, v in s2v.weights.synthetic_vae_state_dict(vcfg, seed=7).items()}
    wh, ww = 12, 16
    all_threads = torch.get_num_threads()
    cores = min(cores, 16)
    torch.set_num_threads(cores)
    lat = torch.randn(1, 2, 16, wh, ww, generator=torch.Generator().manual_seed(23)).to(dt).float()
    with torch.no_grad():
        t0 = time.time()
        exp = vae_ref.decode_latents(sd, cfgd, lat, False)
        dts = time.time() - t0
    torch.set_num_threads(all_threads)
    vae = s2v.HipAutoencoderKLCogVideoX(vcfg, dt, dev)
    vae.load_state_dict(sd)
    got = vae.decode_latents(lat.to(dev, dt)).float().cpu()
    torch.cuda.synchronize()
    vae.close()
    # tiles of tiled_decode (autoencoder_kl_cogvideox.py:1400-1406: 30 x 45 latent tiles every 25 rows / 36 columns, the last ones partial), each
    # decoded in F / 2 two-frame-batch units (one 3-frame batch = 1.5, then 2-frame batches: :1237-1245)
    tiles = [(min(30, H - r), min(45, W - c)) for r in range(0, H, 25) for c in range(0, W, 36)] if (H > 30 or W > 45) else [(H, W)]
    units = (F / 2.0) * sum(th * tw for th, tw in tiles) / (wh * ww)
    unit_flop = 441e12 / (6.5 * 7560 / (wh * ww))  # 441 TFLOP for the tiled decode of 49 x 480 x 720 = 6.5 units x 7560 latent pixels
    # (its nine tiles are 30/30/10 rows x 45/45/18 columns: 1.4 x the 5400 pixels of the untiled 315 TFLOP; until round 5 this leg extrapolated with nine FULL tiles, 1.6 x too much CPU time)
    return {"value": round(units * dts, 1), "unit": f"s per tiled decode of {(F - 1) * 4 + 1} x {H * 8} x {W * 8} (extrapolated)", "cores": cores, "kind": "port",
            "sample": f"one two-frame batch of a {wh} x {ww} latent window of the real-width decoder, torch fp32 on {cores} threads: {dts:.1f} s, "
                      f"extrapolated x{units:.0f} (area of the {len(tiles)} tiles, {F / 2.0:g} two-frame batches)",
            "gflops": round(unit_flop / dts / 1e9, 1), "rel_l2_vs_hip": round(((got - exp).norm() / exp.norm()).item(), 6),
            "max_abs_vs_hip": round((got - exp).abs().max().item(), 5), "max_abs_ref": round(exp.abs().max().item(), 3)}


# kernel names as rocprofv3 prints them (template arguments included)
KERNEL_OF_CLASS = {"attention": "attn_qx_persist_k<2>", "gemm_qkv": "gemm_g4t<4>", "gemm_ff1_gelu": "gemm_g4t<1>",
                   "gemm_out": "gemm_g4<2>", "gemm_ff2": "gemm_g4<2>"}  # <4>: QKV with the fused q/k norm + rotary epilogue; g4t: trickled epilogue
# what the same classes run on at other geometries / widths (the first name present in the PMC pass counts): short sequences take attn_pp_k, few-tile
# and non-256-multiple shapes the eight-wave / four-wave non-persistent kernels
KERNEL_ALTERNATIVES = {"attention": ["attn_pp_k<false>", "attn_pp_k<true>"], "gemm_qkv": ["gemm_g4<4>", "gemm_bf16_pp64<4>"],
                       "gemm_ff1_gelu": ["gemm_g4<1>", "gemm_bf16_pp64<1>"], "gemm_out": ["gemm_bf16_stag<2>"], "gemm_ff2": ["gemm_bf16_stag<2>"]}


# which committed rocprofv3 PMC passes belong to which workload (profiles/README.md): rNN_ = the default workload,
# rNNfp8_ = the fp8 engine at the same geometry, rNN_c1_ = the C1 geometry; the other workloads have no PMC pass
PMC_PREFIX = {"cogvideox-5b-49x480x720": r"r\d+_pmc", "cogvideox-5b-fp8-49x480x720": r"r\d+fp8_pmc", "cogvideox-2b-9x256x256": r"r\d+_c1_pmc",
              "cogvideox-5b-fp8-49x720x1280": r"r\d+_c5fp8_pmc",   # rNN_c5fp8_ = BASELINE configs[4]
              "cogvideox-2b-49x480x720": r"r\d+_c2_pmc"}          # rNN_c2_ = BASELINE configs[1] (round 6)


def pmc_traffic_bytes(kernel_class, workload):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes OF THIS WORKLOAD (separate
    --pmc FETCH_SIZE / WRITE_SIZE runs of this same command).  FETCH_SIZE is doubled: on gfx950 it reports half the bytes of a
    wide coalesced stream (MI355X_MICROARCH.md, HBM section).  Returns (bytes or None, [file names])."""
    import csv
    import re

    name = KERNEL_OF_CLASS.get(kernel_class)
    pat = PMC_PREFIX.get(workload)
    if not name or not pat:
        return None, []
    pdir = os.path.join(ROOT, "profiles")
    have = sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []
    fetch = [f for f in have if re.fullmatch(pat + "_fetch\\.csv", f)]
    write = [f for f in have if re.fullmatch(pat + "_write\\.csv", f)]
    if not fetch or not write:
        return None, []

    def norm(n):
        """rocprofv3 prints the return type and every template argument, defaults included"""
        n = n.replace("void ", "").replace(", unsigned short>", ">").replace(", 0, false>", ">").replace(", 0>", ">")
        return n.replace(" ", "")

    def mean_kb(path):
        rows = [r for r in csv.reader(open(path)) if r and r[0] != "kernel"]
        for cand in [name] + KERNEL_ALTERNATIVES.get(kernel_class, []):
            tot = cnt = 0.0
            for row in rows:
                got = norm(row[0])
                # the attention kernel's name carries its format flags (<2, false, false> bf16 P, <2, false, true> fp16 P, <2, true, ...> fp8 QK^T):
                # any four-wave form of the pass counts
                if got == cand.replace(" ", "") or (cand.startswith("attn_qx_persist_k<2") and got.startswith("attn_qx_persist_k<2")):
                    tot += float(row[2]) * float(row[1])
                    cnt += float(row[1])
            if cnt:
                return tot / cnt
        return None

    f, w = mean_kb(os.path.join(pdir, fetch[-1])), mean_kb(os.path.join(pdir, write[-1]))
    if f is None or w is None:
        return None, [fetch[-1], write[-1]]
    return int((2.0 * f + w) * 1024), [fetch[-1], write[-1]]


def free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` with no torchrun environment: launch the N rank processes here, one per GPU (LOCAL_RANK = RANK = i),
    rendezvous on 127.0.0.1, backend nccl (= RCCL).  Refuses when fewer than N GPUs are visible -- S2V_BENCH_ONE_DEVICE=1 (all ranks
    on cuda:0; RCCL rejects two ranks on one device, so it needs S2V_BENCH_BACKEND=gloo) is a functional check of the rank logic on
    a one-GPU box and its line reports the devices actually used.  Returns the exit code (non-zero if any rank failed)."""
    import subprocess

    one_dev = os.environ.get("S2V_BENCH_ONE_DEVICE", "0") == "1"
    selftest = "--dist-selftest" in argv   # CPU / gloo exercise of the launcher, the watchdog and the broadcast: no GPU involved
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if selftest:
        ndev = n
    if ndev < 1:
        raise SystemExit(f"bench.py --gpus {n}: no GPU visible (torch.cuda.is_available() is False); nothing is printed for GPUs that do not exist")
    if ndev < n and not one_dev:
        raise SystemExit(f"bench.py --gpus {n}: only {ndev} GPU(s) visible; refusing to report n_gpus={n} "
                         f"(S2V_BENCH_ONE_DEVICE=1 S2V_BENCH_BACKEND=gloo runs the {n} ranks on cuda:0 as a functional check)")
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(n), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), S2V_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
        env.setdefault("NCCL_DEBUG", "WARN")               # RCCL states why a collective failed instead of just failing
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    return supervise(procs)


def supervise(procs, grace_s=5.0):
    """wait for the rank processes; the FIRST non-zero exit (a crash, a kill -- negative codes --, the watchdog's EXIT_WATCHDOG, a
    non-finite output) ends the others: they would sit in a collective waiting for the dead rank.  terminate, then kill after grace_s
    (exact PIDs).  Returns that first non-zero code, 0 when every rank exited 0."""
    rc = 0
    try:
        pending = list(procs)
        deadline = None
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    sys.stderr.write(f"[bench] rank process {p.pid} exited with code {code}; ending the other {len(pending)} rank(s)\n")
                    for q in pending:
                        q.terminate()
                    deadline = time.time() + grace_s
            if deadline is not None and time.time() > deadline:
                for q in pending:
                    q.kill()
                deadline = None
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def dist_selftest(args):
    """`bench.py --gpus N --dist-selftest` (no GPU): N gloo ranks rendezvous exactly as the benchmark's ranks do and replicate a synthetic CPU
    "arena" with dist.broadcast_arena under the same Watchdog; S2V_SELFTEST_KILL_RANK=r makes rank r die in the middle of the broadcast,
    S2V_SELFTEST_STALL_RANK=r makes it stop taking part (the peers' watchdog has to fire).  Prints one JSON line on success."""
    s2v = importlib.import_module("disentangled-subject-to-vid_amd")
    import torch.distributed as dist

    rank, world, _ = s2v.dist.init_from_env("gloo")
    nbytes = int(os.environ.get("S2V_SELFTEST_BYTES", str(64 << 20)))
    arena = torch.full((nbytes,), 7 if rank == 0 else 0, dtype=torch.uint8)
    kill = int(os.environ.get("S2V_SELFTEST_KILL_RANK", "-1"))
    stall = int(os.environ.get("S2V_SELFTEST_STALL_RANK", "-1"))
    done = [0]
    with s2v.dist.Watchdog("weight broadcast (self-test)", float(os.environ.get("S2V_BENCH_BCAST_TIMEOUT_S", "300")), lambda: f"{done[0]} of {nbytes} bytes enqueued"):
        if world > 1:
            dist.barrier()
        chunk = max(nbytes // 8, 1)
        flat = arena.view(-1)
        for off in range(0, nbytes, chunk):
            if rank == kill and off >= nbytes // 2:
                os._exit(9)                       # a rank that dies mid-broadcast
            if rank == stall and off >= nbytes // 2:
                time.sleep(3600)                   # a rank that stops taking part
            dist.broadcast(flat[off:off + chunk], src=0)
            done[0] = off + chunk
        if world > 1:
            dist.barrier()
    ok = bool((arena == 7).all().item())
    if rank == 0:
        print(json.dumps({"dist_selftest": "ok" if ok else "corrupt", "ranks": world, "bytes": nbytes}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit(5)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cogvideox-5b-49x480x720", choices=sorted(WORKLOADS))
    ap.add_argument("--graph", type=int, default=1, help="timed region replays the step from a captured hipGraph (north_star); 0 = eager launches")
    ap.add_argument("--single-mode", action="store_true", help="skip the pass in the other launch mode (rocprofv3 runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-vae", action="store_true", help="skip the one-off VAE decode timing")
    ap.add_argument("--native-bcast", action="store_true", help="N > 1: replicate the weights with the library's own RCCL communicator "
                    "(s2v_bcast_weights) instead of torch.distributed.broadcast")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "f16"], help="model dtype of the engine, VAE and T5: bf16 = the headline (and every "
                    "5B configuration); f32 = configs[0] as BASELINE names it (the CPU-reference-parity mode, on the fp32 matrix pipe); f16 = what the "
                    "reference loads non-5B checkpoints in (src/inference.py:191).  Non-bf16 lines skip the attention format A/B passes; every dtype times the CPU oracle beside it")
    ap.add_argument("--batch", type=int, default=2, choices=[1, 2], help="2 = the CFG pair on one GPU (the metric); 1 = ONE sample of the pair: what each of the two "
                    "GPUs of a CFG-parallel pair runs per step (s2v_denoise_split_begin + the CFG / scheduler step; the peer's half is a local copy) -- "
                    "reported as projected_cfg_parallel_ms, the step time of one video on two GPUs less the 2.2 MB all-gather")
    ap.add_argument("--cfg-parallel", action="store_true", help="--gpus N (even): ranks 2p, 2p+1 run video p TOGETHER (dist.CfgPair: B = 1 engines, one all-gather "
                    "of noise_pred per step, CFG + scheduler step on both); value = video steps/s over all pairs")
    ap.add_argument("--dist-selftest", action="store_true", help="no GPU: exercise the rank launcher, the rendezvous, the chunked broadcast and its "
                    "watchdog with gloo on CPU tensors (tests/test_dist_gloo.py)")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.cfg_parallel and (args.gpus < 2 or args.gpus % 2):
        raise SystemExit("--cfg-parallel pairs the ranks up: --gpus must be even and >= 2 (one GPU: --batch 1 gives the projection)")
    if args.cfg_parallel:
        args.batch = 1
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        rc = spawn_ranks(args.gpus, argv)
        if rc != 0:
            raise SystemExit(rc if isinstance(rc, int) and rc > 0 else 1)
        return
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={env_world}: the line reports the ranks that exist, not the flag")
    if args.dist_selftest:
        return dist_selftest(args)

    s2v = importlib.import_module("disentangled-subject-to-vid_amd")
    # S2V_BENCH_BACKEND=gloo + S2V_BENCH_ONE_DEVICE=1: a functional check of the multi-rank path on a one-GPU box (all ranks on cuda:0);
    # the driver's multi-GPU runs use neither
    backend = os.environ.get("S2V_BENCH_BACKEND", "nccl")
    one_dev = os.environ.get("S2V_BENCH_ONE_DEVICE", "0") == "1"
    if one_dev:
        os.environ["LOCAL_RANK"] = "0"
    import torch.distributed as dist

    if not one_dev and int(os.environ.get("LOCAL_RANK", "0")) >= torch.cuda.device_count():
        raise SystemExit(f"LOCAL_RANK={os.environ.get('LOCAL_RANK')} but {torch.cuda.device_count()} GPU(s) visible")
    rank, world, local = s2v.dist.init_from_env(backend if int(os.environ.get("WORLD_SIZE", "1")) > 1 else None)
    # from here on the number of ranks is what the process group says, never the flag
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the process group has {world} rank(s)")
    dev = f"cuda:{local}"
    torch.cuda.set_device(dev)

    preset, F, H, W, T = WORKLOADS[args.workload]
    cfg = s2v.config.PRESETS[preset]()
    if os.environ.get("S2V_ATTN_P"):   # same-box A/B of attn_p_format ("bf16" / "f16"); the line reports it in config.attn_p_format
        cfg.attn_p_format = os.environ["S2V_ATTN_P"]
    fp8 = cfg.weight_format in ("fp8", "fp8-qk", "fp8-auto")
    dt = {"bf16": torch.bfloat16, "f32": torch.float32, "f16": torch.float16}[args.dtype]
    if fp8 and args.dtype != "bf16":
        raise SystemExit("fp8 workloads are bf16 engines (weight_format needs the bf16 MFMA path)")
    if args.dtype != "bf16":  # the A/B passes are about the bf16 asm kernels
        os.environ["S2V_BENCH_SKIP_PFMT"] = os.environ["S2V_BENCH_SKIP_PARITY_PASS"] = "1"
    if args.batch == 1:  # the half-step lines carry the step and its per-kernel table; formats, video and CPU baseline belong to the metric's own line
        os.environ["S2V_BENCH_SKIP_PFMT"] = os.environ["S2V_BENCH_SKIP_PARITY_PASS"] = "1"
        args.no_vae = args.no_cpu_baseline = True
    PEAK_DT = {"bf16": PEAK_BF16_TFLOPS, "f16": PEAK_BF16_TFLOPS, "f32": PEAK_F32_TFLOPS}[args.dtype]
    vae = None
    if rank == 0 and not args.no_vae:
        # The decoder is created, and its first tiled decode run, BEFORE the transformer engine exists: the runtime binds a stream to a hardware
        # queue at its first use, and the six streams of the tiled decode (six tiles in flight) share the queues best when they are bound while
        # no other stream of the process has been used (HISTORY.md section 3, tools/vae_inproc_probe.py: 0.525 s against 0.57-0.59 s for a
        # decoder created after a loaded engine; the untiled decode is the same either way).  A caller gets the same by constructing
        # HipAutoencoderKLCogVideoX before the transformer; nothing here is inside the timed region of the metric.
        vcfg = s2v.VAEConfig(scaling_factor=cfg.vae_scaling_factor)
        vae = s2v.HipAutoencoderKLCogVideoX(vcfg, dt, dev)
        vae.load_state_dict(s2v.weights.synthetic_vae_state_dict(vcfg, seed=7, device=dev))
        vae.use_tiling = True
        vae.decode_latents(torch.randn(1, F, cfg.in_channels, H, W, generator=torch.Generator().manual_seed(3)).to(dev, dt))
        torch.cuda.synchronize()
    model = s2v.HipCogVideoXTransformer3DModel(cfg, dt, dev)  # the drop-in object; the pipeline run below goes through it
    eng = model.engine
    t_load = time.time()
    n_lora = 0
    if rank == 0:
        n_lora = load_synthetic(s2v, eng, cfg, 1234)
    bcast_s = bcast_bytes = None
    native_bcast = args.native_bcast
    bcast_note = None
    if world > 1:
        if native_bcast:  # the decision is taken by ALL ranks together: one rank without librccl sends everybody to torch.distributed
            ok, bad = s2v.dist.rccl_available_everywhere()
            if not ok:
                native_bcast = False
                bcast_note = f"--native-bcast requested but librccl could not be bound on rank(s) {[r for r, _ in bad]}: {bad[0][1]}; fell back to torch.distributed.broadcast"
                if rank == 0:
                    sys.stderr.write(f"[bench] WARNING: {bcast_note}\n")
        done = [0]
        with s2v.dist.Watchdog("weight broadcast rank 0 -> all", float(os.environ.get("S2V_BENCH_BCAST_TIMEOUT_S", "300")),
                               lambda: f"{done[0]} bytes enqueued on {dev}, backend {dist.get_backend()}, native={native_bcast}"):
            torch.cuda.synchronize()
            dist.barrier()  # also the communicator's first collective: its lazy set-up stays out of the broadcast time
            tb = time.time()
            if native_bcast:
                comm = s2v.dist.RcclComm()
                torch.cuda.synchronize()
                dist.barrier()
                tb = time.time()  # the communicator's set-up stays out of the broadcast time, as on the other path
                comm.broadcast_weights(eng, 0)  # receivers are marked loaded by the library
                bcast_bytes = eng.weight_arena().numel()
            else:
                bcast_bytes = s2v.dist.broadcast_arena(eng.weight_arena(), 0, progress=done)
            torch.cuda.synchronize()
            dist.barrier()
        bcast_s = time.time() - tb
        if rank != 0 and not native_bcast:
            eng.mark_weights_loaded()
    t_load = time.time() - t_load

    # one prompt per rank (different seeds), same shapes: weak scaling
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    text = torch.randn(2, T, cfg.text_embed_dim, generator=g, device=dev)
    ref = torch.randn(1, 1, cfg.in_channels, H, W, generator=g, device=dev) * 0.7
    latents = torch.randn(1, F, cfg.in_channels, H, W, generator=g, device=dev).to(dt).contiguous()
    cfg_pair = None
    if args.cfg_parallel:
        # video p on ranks 2p, 2p + 1: the pair shares ONE prompt, reference latent and start latents (drawn from the pair's seed); rank slot holds half slot
        cfg_pair = s2v.dist.CfgPair(native=native_bcast)   # --native-bcast also selects s2v_rccl_allgather for the per-step exchange
        g = torch.Generator(device=dev).manual_seed(100 + cfg_pair.pair)
        text = torch.randn(2, T, cfg.text_embed_dim, generator=g, device=dev)
        ref = torch.randn(1, 1, cfg.in_channels, H, W, generator=g, device=dev) * 0.7
        latents = torch.randn(1, F, cfg.in_channels, H, W, generator=g, device=dev).to(dt).contiguous()
    sch = s2v.CogVideoXDDIMScheduler(snr_shift_scale=cfg.snr_shift_scale)
    sch.set_timesteps(50)
    coefs = [sch.coef(t, dt, 6.0) for t in sch.timesteps]

    def set_batch(nb, slot=1):
        """geometry + tables + conditioning of the CFG pair on this GPU (nb = 2) or of ONE sample of it (nb = 1: half `slot` of [negative | positive])"""
        eng.set_geometry(nb, T, F, H, W)
        eng.prepare_tables(H * 8, W * 8)
        eng.set_conditioning(text if nb == 2 else text[slot:slot + 1], ref)

    def step_b1(i, graph, x=None):
        """what ONE GPU of a CFG-parallel pair does per step, on one GPU: the B = 1 forward (graph replay) into its half of the pair buffer, the peer's
        half stood in for by a device copy of its own (the all-gather's local traffic; guidance on two equal halves is the identity), CFG + scheduler step"""
        x = latents if x is None else x
        eng.denoise_split_begin(x, float(sch.timesteps[i % 50]), coefs[i % 50], 1, use_graph=graph)
        pair = eng.cfg_pair()
        pair[0].copy_(pair[1], non_blocking=True)
        eng.denoise_split_end(x)

    set_batch(args.batch, cfg_pair.slot if cfg_pair else 1)

    def step(i, graph):
        if cfg_pair is not None:
            cfg_pair.step(eng, latents, float(sch.timesteps[i % 50]), coefs[i % 50], use_graph=graph)
        elif args.batch == 1:
            step_b1(i, graph)
        else:
            eng.denoise_step(latents, float(sch.timesteps[i % 50]), coefs[i % 50], use_graph=graph)

    def timed(graph, nsteps, nwarm):
        """W untimed + K timed steps between barrier + synchronize on both sides; max over ranks"""
        for i in range(nwarm):
            step(i, graph)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for i in range(nsteps):
            step(nwarm + i, graph)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        total = nsteps
        if world > 1:
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = tt.item()
            ts = torch.tensor([nsteps], device=dev, dtype=torch.int64)  # the steps every rank actually ran, summed
            dist.all_reduce(ts, op=dist.ReduceOp.SUM)
            total = int(ts.item())
        return el, total

    # attn_p_format "auto" (opt-in, S2V_ATTN_P=auto; the default is bf16 P): the engine runs its first denoise step eagerly, reads the attention kernel's slow-path census and settles
    # on fp16 or bf16 P for good (engine.py).  That step is taken here, before anything is timed, on a copy of the latents.
    if getattr(eng, "_attn_auto_pending", False):
        keep = latents.clone()
        step(0, False)
        latents.copy_(keep)
    # ---- the timed region of the contract: no event recording, no profiling inside it
    elapsed, total_steps = timed(bool(args.graph), args.steps, args.warmup)
    finite = bool(torch.isfinite(latents.float()).all().item())
    # who ran: one record per rank (device index, name, bus id), gathered on rank 0 -- n_gpus is the number of DISTINCT devices
    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "device": dev, "name": props.name, "pci_bus_id": getattr(props, "pci_bus_id", None),
          "uuid": str(getattr(props, "uuid", "")), "steps": args.steps, "outputs_finite": finite}
    ranks_info = [me]
    if world > 1:
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, me)
    n_devices = len({(r["device"], r["uuid"]) for r in ranks_info})
    finite = all(r["outputs_finite"] for r in ranks_info)
    if not finite:  # a number measured on NaN latents is not a measurement: no metric line, every rank exits non-zero (they all hold ranks_info)
        badr = [r["rank"] for r in ranks_info if not r["outputs_finite"]]
        if rank == 0:
            sys.stderr.write(f"[bench] FAILED: non-finite latents after the timed steps on rank(s) {badr}; ranks: {json.dumps(ranks_info)}\n")
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        raise SystemExit(4)
    # ---- the other launch mode, timed the same way (beside the metric, not in it)
    other_steps = min(args.steps, 5)
    other = None if args.single_mode else timed(not bool(args.graph), other_steps, 1)[0]
    this_ms, other_ms = elapsed / args.steps * 1e3, None if other is None else other / other_steps * 1e3
    graph_ms, eager_ms = (this_ms, other_ms) if args.graph else (other_ms, this_ms)

    # ---- per-kernel durations: a SEPARATE eager pass with HIP events around every launch on the launch stream
    roofline = None
    if not args.no_roofline:
        import ctypes

        prof_steps = min(args.steps, 3)

        def profile_pass(engine, nsteps, stepfn):
            """separate eager pass: HIP events + shader-clock stamps around every launch of a class -> (ms, launches, MHz per class, wall s)"""
            s2v._lib.check(s2v.lib().s2v_profile_enable(engine._h, 1))
            torch.cuda.synchronize()
            tp = time.perf_counter()
            for i in range(nsteps):
                stepfn(i)
            torch.cuda.synchronize()
            wall = time.perf_counter() - tp
            mhz_ = (ctypes.c_float * 8)()
            s2v._lib.check(s2v.lib().s2v_profile_read_clocks(engine._h, mhz_, 8))
            ms_ = (ctypes.c_float * 8)()
            cnt_ = (ctypes.c_int32 * 8)()  # eight classes: CLASSES
            s2v._lib.check(s2v.lib().s2v_profile_read(engine._h, ms_, cnt_, 8))
            s2v._lib.check(s2v.lib().s2v_profile_enable(engine._h, 0))
            return list(ms_), list(cnt_), list(mhz_), wall

        ms, cnt, mhz, prof_elapsed = profile_pass(eng, prof_steps, lambda i: step(i, False))
        N = T + (F + 1) * (H // 2) * (W // 2)
        D = cfg.inner_dim
        NB = args.batch  # samples per forward on this GPU: 2 = the CFG pair, 1 = one sample of it (--batch 1 / --cfg-parallel)
        flops = {"gemm_qkv": 2 * NB * N * D * 3 * D, "attention": 4 * NB * N * N * D, "gemm_out": 2 * NB * N * D * D,
                 "gemm_ff1_gelu": 2 * NB * N * D * 4 * D, "gemm_ff2": 2 * NB * N * D * 4 * D}
        # HBM-bound kernels: algorithmic bytes per launch: LayerNorm + modulate reads and writes the residual
        # stream once; the V transpose reads V and writes V^T; the modulation GEMV streams every norm linear's weights once per step
        E = 4 if args.dtype == "f32" else 2
        mod_rows = 2 * cfg.num_layers * 6 * D + 2 * D
        hbm_bytes = {"ln_modulate": 2 * (NB * N * D * E), "qknorm_rope_vt": 2 * (NB * N * D * E), "mod_gemv": mod_rows * cfg.time_embed_dim * E}
        per_kernel = {}
        for k, name in enumerate(CLASSES):
            if cnt[k] == 0:
                continue
            avg = ms[k] / cnt[k]
            e = {"avg_ms": round(avg, 4), "launches": int(cnt[k]), "share_of_step": round(ms[k] / (prof_elapsed * 1e3), 4)}
            if mhz[k] > 0:
                e["shader_clock_mhz"] = round(mhz[k], 0)
            if name in flops:
                e["tflops"] = round(flops[name] / avg / 1e9, 1)
                e["peak_tflops"] = PEAK_FP8_TFLOPS if (fp8 and name.startswith("gemm_")) else PEAK_DT
                e["frac_of_peak"] = round(e["tflops"] / e["peak_tflops"], 4)
                if mhz[k] > 0:  # the datasheet peak is quoted at 2400 MHz; at the clock the part actually ran this launch at, the pipe's ceiling was
                    e["peak_at_measured_clock_tflops"] = round(e["peak_tflops"] * mhz[k] / PEAK_CLOCK_MHZ, 1)
                    e["frac_of_peak_at_measured_clock"] = round(e["tflops"] / e["peak_at_measured_clock_tflops"], 4)
            if name in hbm_bytes:
                e["bound"] = "hbm"
                e["gb_per_s"] = round(hbm_bytes[name] / avg / 1e6, 1)
                e["frac_of_hbm_peak"] = round(e["gb_per_s"] / PEAK_HBM_GBS, 4)
            per_kernel[name] = e
        dom = max((n for n in per_kernel if n in flops), key=lambda n: per_kernel[n]["avg_ms"] * per_kernel[n]["launches"])
        ach = per_kernel[dom]["tflops"]
        peak = per_kernel[dom]["peak_tflops"]  # the dominant kernel's own matrix-core peak (fp8 GEMMs: 5 PF, everything else 2.5 PF)
        traffic, traffic_files = pmc_traffic_bytes(dom, args.workload) if args.dtype == "bf16" else (None, [])  # the committed PMC passes are bf16 runs
        # unique operand bytes of one launch (bf16 activations; fp8 engines read 1-byte weights / activations on the GEMMs):
        # attention reads Q, K, V^T and writes O; a GEMM reads A [M,K] and W [N,K] and writes C [M,N]
        Eg = 1 if fp8 else 2
        Mr = NB * N
        algo_bytes = {"attention": 4 * Mr * D * 2, "gemm_qkv": Mr * D * Eg + 3 * D * D * Eg + Mr * 3 * D * 2,
                      "gemm_out": Mr * D * Eg + D * D * Eg + Mr * D * 2, "gemm_ff1_gelu": Mr * D * Eg + 4 * D * D * Eg + Mr * 4 * D * Eg,
                      "gemm_ff2": Mr * 4 * D * Eg + 4 * D * D * Eg + Mr * D * 2}
        if args.dtype == "f32":
            algo_bytes = {k_: 2 * v_ for k_, v_ in algo_bytes.items()}
        roofline = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": traffic,
                    "traffic_unit": "bytes/launch = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH, HBM section) from "
                                    + (" + ".join("profiles/" + f for f in traffic_files) if traffic_files else "no PMC pass committed for this workload"),
                    "algorithmic_bytes_per_launch": algo_bytes.get(dom),
                    "traffic_over_algorithmic": round(traffic / algo_bytes[dom], 3) if traffic and algo_bytes.get(dom) else None,
                    "algorithmic_flops_per_launch": flops[dom], "avg_launch_ms": per_kernel[dom]["avg_ms"],
                    "measured": f"HIP events on the launch stream in a separate eager pass of {prof_steps} steps after the timed region",
                    "per_kernel": per_kernel}
        if rank == 0:
            # (a) CALIBRATED peak: what the vendor library reaches in this process, on this box, now, on the FF1 shape -- the part is
            # power-managed (shader_clock_mhz above), so the datasheet's 2.5 PF at 2.4 GHz is not what the matrix pipe can deliver under load
            Mr2, Kc, Nc = 2 * N, D, 4 * D   # always the B = 2 FF1 shape: the calibration is about the box, not the batch
            xa = torch.randn(Mr2, Kc, device=dev, dtype=dt)
            xw = torch.randn(Kc, Nc, device=dev, dtype=dt)
            for _ in range(3):
                torch.matmul(xa, xw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                torch.matmul(xa, xw)
            e1.record()
            torch.cuda.synchronize()
            cal = 2.0 * Mr2 * Kc * Nc / (e0.elapsed_time(e1) / 10) / 1e9
            del xa, xw
            roofline["calibrated_peak"] = round(cal, 1)
            roofline["calibrated_peak_note"] = (f"torch.matmul (hipBLASLt) {args.dtype} [{Mr2} x {Kc}] x [{Kc} x {Nc}] (the FF1 shape, no epilogue), 10 launches, same process / box, "
                                                "right after the profile pass; fp8 GEMM classes are held against twice this figure")
            for name, e in per_kernel.items():
                if "tflops" in e:
                    e["frac_of_calibrated"] = round(e["tflops"] / (cal * e["peak_tflops"] / PEAK_DT), 4)
            roofline["frac_of_calibrated"] = per_kernel[dom].get("frac_of_calibrated")
            roofline["shader_clock_mhz"] = per_kernel[dom].get("shader_clock_mhz")
            roofline["frac_of_peak_at_measured_clock"] = per_kernel[dom].get("frac_of_peak_at_measured_clock")
            roofline["shader_clock_note"] = ("s_memtime / s_memrealtime stamps on the launch stream around every profiled launch (s2v_profile_read_clocks); "
                                             f"datasheet peaks are quoted at {PEAK_CLOCK_MHZ:.0f} MHz")
            # (c) the attention kernel in BOTH probability formats (the timed region ran config.attn_p_format), same engine, same data
            if eng.attn_p_format in ("bf16", "f16") and "attention" in per_kernel and not os.environ.get("S2V_BENCH_SKIP_PFMT"):
                ran = eng.attn_p_format
                both = {ran: {k_: per_kernel["attention"].get(k_) for k_ in ("avg_ms", "tflops", "frac_of_peak", "shader_clock_mhz", "frac_of_calibrated")}}
                other_fmt = "f16" if ran == "bf16" else "bf16"
                eng.set_attn_p_format(other_fmt)
                ms2, cnt2, mhz2, _ = profile_pass(eng, 1, lambda i: step(i, False))
                eng.set_attn_p_format(ran)
                if cnt2[1]:
                    a2 = ms2[1] / cnt2[1]
                    tf2 = flops["attention"] / a2 / 1e9
                    both[other_fmt] = {"avg_ms": round(a2, 4), "tflops": round(tf2, 1), "frac_of_peak": round(tf2 / PEAK_BF16_TFLOPS, 4),
                                       "shader_clock_mhz": round(mhz2[1], 0), "frac_of_calibrated": round(tf2 / cal, 4)}
                roofline["attention_by_p_format"] = both
            # (c') the same kernel on data with SHARP attention: a second engine with the parity tests' larger-variance weights (no LoRA), one eager
            # step per format; the fp16-P kernel's slow-path census is then non-trivial and its price visible
            if not fp8 and not os.environ.get("S2V_BENCH_SKIP_PARITY_PASS"):
                import copy

                c2 = copy.copy(cfg)
                c2.attn_p_format = "f16"
                eng2 = s2v.S2VEngine(c2, dt, dev)
                load_synthetic(s2v, eng2, c2, 4321, parity=True)
                eng2.set_geometry(2, T, F, H, W)
                eng2.prepare_tables(H * 8, W * 8)
                eng2.set_conditioning(text, ref)
                lat2 = latents.clone()
                res2 = {}
                for fmt in ("f16", "bf16"):
                    eng2.set_attn_p_format(fmt)
                    eng2.attn_slow_stats(reset=True)
                    lat2.copy_(latents)
                    msp, cntp, mhzp, _ = profile_pass(eng2, 1, lambda i: eng2.denoise_step(lat2, float(sch.timesteps[25]), coefs[25], use_graph=False))
                    slow, tot = eng2.attn_slow_stats(reset=True)
                    ap = msp[1] / max(cntp[1], 1)
                    res2[fmt] = {"avg_ms": round(ap, 4), "tflops": round(flops["attention"] / ap / 1e9, 1), "frac_of_peak": round(flops["attention"] / ap / 1e9 / PEAK_BF16_TFLOPS, 4),
                                 "slow_path_fraction": round(slow / tot, 6) if tot else None, "shader_clock_mhz": round(mhzp[1], 0)}
                res2["note"] = "weights.synthetic_state_dict(parity=True)-style weights (std 0.7 / sqrt(fan_in), non-zero biases, LayerNorm affines), timestep 499; outputs finite: " \
                               + str(bool(torch.isfinite(lat2.float()).all().item()))
                roofline["attention_on_parity_weights"] = res2
                eng2.close()
                del eng2, lat2
                torch.cuda.empty_cache()

    # ---- CFG-parallel projection: the step ONE GPU of a pair runs, timed here on one GPU the same way as the metric
    cfgp = None
    if rank == 0 and world == 1 and args.batch == 2 and not args.single_mode and not os.environ.get("S2V_BENCH_SKIP_CFGP"):
        keep = latents.clone()
        set_batch(1)
        nb1 = min(args.steps, 10)
        for i in range(2):
            step_b1(i, True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(nb1):
            step_b1(2 + i, True)
        torch.cuda.synchronize()
        b1_ms = (time.perf_counter() - t1) / nb1 * 1e3
        cfgp = {"b1_step_ms": round(b1_ms, 2), "ratio_to_cfg_pair_step": round(b1_ms / (elapsed / args.steps * 1e3), 4),
                "projected_cfg_parallel_ms": round(b1_ms, 2), "projected_steps_per_s_per_video_on_2_gpus": round(1e3 / b1_ms, 4),
                "note": f"{nb1} hipGraph steps of ONE sample of the CFG pair (s2v_denoise_split_begin, the peer's half stood in for by a device copy, "
                        "s2v_denoise_split_end) on this GPU: what each GPU of a CFG-parallel pair runs per step, less the 2.2 MB all-gather over xGMI; "
                        "`python bench.py --batch 1` gives the per-kernel table of this step, `--gpus 2 --cfg-parallel` the real thing"}
        if not args.no_roofline:
            msb, cntb, _, _ = profile_pass(eng, 1, lambda i: step_b1(i, False))
            cfgp["per_kernel_avg_ms"] = {name: round(msb[k] / cntb[k], 4) for k, name in enumerate(CLASSES) if cntb[k]}
            cfgp["per_kernel_ratio_to_cfg_pair"] = {name: round(v / roofline["per_kernel"][name]["avg_ms"], 4)
                                                    for name, v in cfgp["per_kernel_avg_ms"].items() if name in roofline["per_kernel"]}
        set_batch(2)
        latents.copy_(keep)

    video = None
    if rank == 0 and not args.no_vae:
        # wall-clock per video = 50 denoise steps + VAE decode (BASELINE.json metric, second half); decode is timed once
        # outside the step timing, untiled (288 GB part) and tiled (what src/inference.py:204-207 enables)
        dec = {}
        for tiling in (False, True):
            vae.use_tiling = tiling
            vae.decode_latents(latents)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            frames = vae.decode_latents(latents)
            torch.cuda.synchronize()
            dec["tiled" if tiling else "untiled"] = time.perf_counter() - t1
        vae_sets, vae_set_bytes = vae.workspace_info()
        # (d) MEASURED wall-clock of one video: the real 50-step S2VPipeline call (hipGraph replay) through the drop-in transformer object + tiled VAE
        # decode + post-processing to numpy frames, as src/inference.py:204-207 / custom_cogvideox_pipe.py:237-316 run it (prompt embeddings given)
        vae.use_tiling = True
        pipe = s2v.S2VPipeline(model, s2v.CogVideoXDDIMScheduler(snr_shift_scale=cfg.snr_shift_scale), vae)
        lat_in = torch.randn(1, F, cfg.in_channels, H, W, generator=torch.Generator(device=dev).manual_seed(5), device=dev).to(dt)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        vid_np = pipe(prompt_embeds=text[1:2].to(dt), negative_prompt_embeds=text[0:1].to(dt), ref_img_states=ref.to(dt), height=H * 8, width=W * 8,
                      num_frames=(F - 1) * 4 + 1, num_inference_steps=50, guidance_scale=6.0, latents=lat_in, output_type="np", return_dict=False,
                      use_graph=True)[0]
        torch.cuda.synchronize()
        measured_video_s = time.perf_counter() - t1
        import numpy as _np

        measured_video_ok = bool(_np.isfinite(_np.asarray(vid_np)).all())
        measured_video_shape = list(_np.asarray(vid_np).shape)
        del pipe, vid_np
        # reference-image encode (src/video_generate.py:26-38), reported beside the metric, not inside it
        vae.load_state_dict(s2v.weights.synthetic_vae_encoder_state_dict(vcfg, seed=8, device=dev, dtype=dt))
        img = (torch.rand(1, 3, 1, H * 8, W * 8, generator=torch.Generator().manual_seed(9)) * 2 - 1).to(dev, dt)
        enc = {}
        for tiling in (False, True):
            vae.use_tiling = tiling
            vae.encode(img).latent_dist.sample(generator=torch.Generator().manual_seed(1))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ref_lat = vae.encode(img).latent_dist.sample(generator=torch.Generator().manual_seed(1)) * vcfg.scaling_factor
            torch.cuda.synchronize()
            enc["tiled" if tiling else "untiled"] = time.perf_counter() - t1
        vae.close()
        vae = None
        # prompt embeddings: T5-v1.1-XXL encoder, prompt + negative prompt, 226 tokens (pipeline_cogvideox.py:197-237)
        tcfg = s2v.T5Config()
        t5 = s2v.HipT5EncoderModel(tcfg, dt, dev)
        t5.load_state_dict(s2v.weights.synthetic_t5_state_dict(tcfg, seed=10, device=dev, dtype=dt, gain=0.5))
        ids = torch.randint(0, tcfg.vocab_size, (2, 226), generator=torch.Generator().manual_seed(11)).to(dev)
        t5(ids)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        emb = t5(ids)[0]
        torch.cuda.synchronize()
        t5_s = time.perf_counter() - t1
        t5_ok = bool(torch.isfinite(emb.float()).all().item())
        t5.close()
        step_s = elapsed / args.steps
        video = {"denoise_steps": 50, "denoise_s": round(50 * step_s, 2),
                 "text_encode_t5xxl_s": round(t5_s, 4), "text_embeds_finite": t5_ok,
                 "ref_image_encode_untiled_s": round(enc["untiled"], 4), "ref_image_encode_tiled_s": round(enc["tiled"], 4),
                 "ref_latent_finite": bool(torch.isfinite(ref_lat.float()).all().item()),
                 "vae_decode_untiled_s": round(dec["untiled"], 3), "vae_decode_tiled_s": round(dec["tiled"], 3),
                 "vae_tiles_in_flight": vae_sets, "vae_workspace_set_gb": round(vae_set_bytes / 1e9, 1),
                 "s_per_video_untiled": round(50 * step_s + dec["untiled"], 2),
                 "s_per_video_tiled": round(50 * step_s + dec["tiled"], 2),
                 "s_per_video_measured": round(measured_video_s, 2),
                 "s_per_video_measured_note": "ONE real S2VPipeline call: 50 DDIM steps (hipGraph replay) + tiled VAE decode + post-processing to numpy frames "
                                              f"{measured_video_shape}, finite {measured_video_ok}; the two figures above are 50 x the timed step + the separately timed decode",
                 "frames": list(frames.shape), "frames_finite": bool(torch.isfinite(frames.float()).all().item())}
    if rank == 0:
        out = {
            "metric": ("denoise steps/sec (CogVideoX-5B, 49f 720x480; one step = CFG-pair transformer forward + CFG + "
                       "scheduler step)" if "5b" in args.workload else f"denoise steps/sec ({args.workload})")
                      + ("" if args.batch == 2 else "; each step of a video runs on TWO GPUs (CFG-parallel), one sample of the pair per GPU"
                         if args.cfg_parallel else "; --batch 1: ONE sample of the CFG pair per step on this GPU = half a step (the CFG-parallel projection)"),
            # sum over ranks of the steps they ran / max over ranks of the elapsed time; a rank of a CFG-parallel pair (and a --batch 1 run) does HALF of every step
            "value": round(total_steps / elapsed / (1 if args.batch == 2 else 2), 4),
            "unit": "steps/s",
            "n_gpus": n_devices, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "strong" if (args.cfg_parallel and world == 2) else "weak", "vs_baseline": None,
            "dtype": ("fp8 (e4m3 W8A8 block linears, MX e4m3 QK^T) + bf16" if eng.fp8_qk_active else "fp8 (e4m3 W8A8 block linears) + bf16") if fp8 else args.dtype, "data": "synthetic (seeded N(0,0.02^2) weights, N(0,1) latents / prompt embeddings)",
            "config": {"workload": args.workload, "latent_frames": F, "latent_hw": [H, W], "tokens": T + (F + 1) * (H // 2) * (W // 2),
                       "cfg_pair": 2, "samples_per_gpu_per_step": args.batch, "scheduler": "ddim-trailing-50",
                       "projected_cfg_parallel_ms": round(elapsed / args.steps * 1e3, 2) if (args.batch == 1 and not args.cfg_parallel) else (cfgp or {}).get("projected_cfg_parallel_ms"),
                       "cfg_parallel_projection": cfgp,
                       "cfg_parallel": None if not args.cfg_parallel else {"pairs": world // 2, "exchange": "s2v_rccl_allgather (the library's own communicator)" if native_bcast
                                                                           else f"torch.distributed.all_gather in the pair's sub-group ({dist.get_backend()})",
                                                                           "bytes_per_rank_per_step": int(latents.numel() * latents.element_size())}, "weight_format": cfg.weight_format, "fp8_qk_active": eng.fp8_qk_active if fp8 else None, "attn_p_format": cfg.attn_p_format if cfg.attn_p_format != "auto" else f"auto -> {eng.attn_p_format}",
                       "attn_slow_path_fraction": eng.attn_slow_fraction, "parallelism": f"cfg-parallel pairs x{world // 2}" if args.cfg_parallel else f"replicas x{world}",
                       "rccl_ranks": world, "backend": (dist.get_backend() if dist.is_initialized() else None),
                       "launcher": "self-spawned" if os.environ.get("S2V_BENCH_SPAWNED") == "1" else ("torchrun env" if env_world is not None else "single process"),
                       "ranks": ranks_info,
                       "one_device_functional_check": True if (one_dev and world > 1) else None,
                       "hipgraph": bool(args.graph), "graph_ms_per_step": None if graph_ms is None else round(graph_ms, 2),
                       "eager_ms_per_step": None if eager_ms is None else round(eager_ms, 2),
                       "per_gpu_steps_per_s": round(total_steps / elapsed / n_devices / (1 if args.batch == 2 else 2), 4),
                       "lora_merged": f"rank-128 synthetic adapter on {n_lora} weights (alpha / r = 0.5)" if n_lora else None, "weight_load_s": round(t_load, 2), "weight_broadcast_s": None if bcast_s is None else round(bcast_s, 3),
                       "weight_broadcast_via": None if bcast_s is None else ("s2v_bcast_weights (the library's own RCCL communicator)" if native_bcast else f"torch.distributed.broadcast ({dist.get_backend()}), 256-MiB chunks"),
                       "weight_broadcast_note": bcast_note,
                       "weight_broadcast_gb": None if not bcast_bytes else round(bcast_bytes / 1e9, 3),
                       "weight_broadcast_gb_per_s": None if not bcast_s else round(bcast_bytes / bcast_s / 1e9, 1),
                       "xgmi_link_bound_gb_per_s": 153.0 if world > 1 else None,
                       "outputs_finite": finite},
            "roofline": roofline,
            "wall_clock_per_video": video,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(s2v, cfg, F, H, W, T, dev, dt)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
