"""CFG-parallel (round 6; include/s2v_hip.h s2v_denoise_split_*, dist.CfgPair): ONE video on TWO GPUs.

The CFG pair custom_cogvideox_pipe.py:255-265 batches is two independent forwards that meet in the guidance formula (:266-279).  A rank of a pair
holds a B = 1 geometry with its half of [negative | positive] and the un-duplicated reference tokens.  Claims held here, all BITWISE:

  * a B = 1 engine's forward equals the corresponding half of the B = 2 engine's forward (every kernel of the path treats rows / (sample, head)
    pairs independently and sums each dot product in the same order whatever tile the row sits in), at the tiny goldens' size, at a multi-tile
    size with a ragged row tail, and at the full 19 126 tokens of the headline configuration at 5B width;
  * begin (slot 0) + begin (slot 1) + end == s2v_denoise_step, DDIM and DPM, eager and hipGraph;
  * two PROCESSES sharing cuda:0 (gloo carrying device tensors: RCCL refuses two ranks on one device; the 2-GPU run takes the same code with
    backend nccl or the library's own communicator) run S2VPipeline(cfg_parallel=CfgPair) and both end with the latents of the one-process
    pipeline, bit for bit.
"""
import importlib
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


def _inputs(cfg, T, F, H, W, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    text = torch.randn(2, T, cfg.text_embed_dim, generator=g, device=DEV)
    ref = torch.randn(1, 1, cfg.in_channels, H, W, generator=g, device=DEV) * 0.7
    lat = torch.randn(1, F, cfg.in_channels, H, W, generator=g, device=DEV)
    return text, ref, lat


def _engine(s2v, cfg, dt, sd, B, text, ref, T, F, H, W):
    m = s2v.HipCogVideoXTransformer3DModel(cfg, dt, DEV)
    m.load_state_dict(sd)
    eng = m.engine
    eng.set_geometry(B, T, F, H, W)
    eng.prepare_tables(H * 8, W * 8)
    eng.set_conditioning(text, ref)
    return m, eng


CASES = {
    # name: (config factory, T, F, H, W): tiny = the goldens' size; mid = 3 x 34 x 46 -> N = 7 + 391 * 4 = 1571 tokens per sample, ragged against every
    # tile size (B = 2: 3142 rows = 12 row tiles + 70 rows); sincos = the 2B positional path
    "tiny-rope": (lambda s2v: s2v.tiny(use_rope=True, heads=2, layers=2, text_dim=64, temb=64), 5, 2, 8, 12),
    "tiny-sincos": (lambda s2v: s2v.tiny(use_rope=False, heads=2, layers=2, text_dim=64, temb=64), 5, 2, 8, 12),
    "mid-rope": (lambda s2v: s2v.tiny(use_rope=True, heads=6, layers=2, text_dim=128, temb=64), 7, 3, 34, 46),
}


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32, torch.float16], ids=["bf16", "f32", "f16"])
def test_b1_forward_is_the_half_of_the_b2_forward_bitwise(s2v, case, dt):
    mk, T, F, H, W = CASES[case]
    cfg = mk(s2v)
    sd = s2v.weights.synthetic_state_dict(cfg, seed=61, parity=True)
    text, ref, lat = _inputs(cfg, T, F, H, W, 62)
    lat = lat.to(dt).contiguous()
    m2, e2 = _engine(s2v, cfg, dt, sd, 2, text, ref, T, F, H, W)
    y2 = e2.forward(lat, torch.tensor([321.0, 321.0]), shared_latent=True).clone()
    for slot in (0, 1):
        m1, e1 = _engine(s2v, cfg, dt, sd, 1, text[slot:slot + 1], ref, T, F, H, W)
        y1 = e1.forward(lat, torch.tensor([321.0]), shared_latent=True)
        torch.cuda.synchronize()
        assert torch.isfinite(y1.float()).all()
        assert torch.equal(y1[0], y2[slot]), f"slot {slot}: the B = 1 forward differs from its half of the CFG pair"
        e1.close()
    assert not torch.equal(y2[0], y2[1])
    e2.close()


@pytest.mark.parametrize("kind", ["ddim", "dpm"])
@pytest.mark.parametrize("graph", [False, True], ids=["eager", "graph"])
def test_split_step_equals_the_fused_step_bitwise(s2v, kind, graph):
    """three steps: s2v_denoise_step on the B = 2 engine against begin(0) on one B = 1 engine, begin(1) on another, the halves copied across (what the
    exchange does), end on both: identical latents on both "ranks" and equal to the fused step's; DPM: same noise on both, x0 history per rank"""
    mk, T, F, H, W = CASES["mid-rope"]
    cfg = mk(s2v)
    dt = torch.bfloat16
    sd = s2v.weights.synthetic_state_dict(cfg, seed=63, parity=True)
    text, ref, lat0 = _inputs(cfg, T, F, H, W, 64)
    lat0 = lat0.to(dt).contiguous()
    m2, e2 = _engine(s2v, cfg, dt, sd, 2, text, ref, T, F, H, W)
    ranks = [_engine(s2v, cfg, dt, sd, 1, text[s:s + 1], ref, T, F, H, W) for s in (0, 1)]
    sch = (s2v.CogVideoXDDIMScheduler if kind == "ddim" else s2v.CogVideoXDPMScheduler)(snr_shift_scale=1.0)
    sch.set_timesteps(3)
    ts = sch.timesteps
    lat_f = lat0.clone()
    lat_r = [lat0.clone(), lat0.clone()]
    dpm = kind == "dpm"
    x0_f = torch.zeros(lat0.shape, dtype=torch.float32, device=DEV) if dpm else None
    x0_r = [torch.zeros(lat0.shape, dtype=torch.float32, device=DEV) if dpm else None for _ in (0, 1)]
    gen = torch.Generator(device=DEV).manual_seed(65)
    for i, t in enumerate(ts):
        noise = torch.randn(lat0.shape, generator=gen, device=DEV).to(dt) if dpm else None
        coef = sch.coef(t, ts[i - 1] if i > 0 else None, i == 0, dt, 6.0) if dpm else sch.coef(t, dt, 6.0)
        e2.denoise_step(lat_f, float(t), coef, x0_f, noise, use_graph=graph)
        for s, (_, e) in enumerate(ranks):
            e.denoise_split_begin(lat_r[s], float(t), coef, s, use_graph=graph)
        p0, p1 = ranks[0][1].cfg_pair(), ranks[1][1].cfg_pair()
        p0[1].copy_(p1[1])
        p1[0].copy_(p0[0])
        for s, (_, e) in enumerate(ranks):
            e.denoise_split_end(lat_r[s], x0_r[s], noise)
        torch.cuda.synchronize()
        assert torch.equal(p0, e2.last_noise_pred()), f"step {i}: the gathered pair differs from the B = 2 model output"
        assert torch.equal(lat_r[0], lat_r[1]), f"step {i}: the two ranks' latents differ"
        assert torch.equal(lat_r[0], lat_f), f"step {i}: split step differs from the fused step"
        if dpm:
            assert torch.equal(x0_r[0], x0_f) and torch.equal(x0_r[1], x0_f)
    assert torch.isfinite(lat_f.float()).all() and not torch.equal(lat_f, lat0)
    for _, e in ranks:
        e.close()
    e2.close()


def test_full_tokens_5b_width_b1_is_the_half_of_b2_bitwise(s2v):
    """the headline geometry (N = 19 126 per sample, 5B width, two layers): M = 19 126 rows against M = 38 252 take different tile walks, row-tail
    splits and persistent-launch schedules (75 x 12 = 900 tiles = 3.5 rounds against 150 x 12) -- per sample the bits must not care"""
    cfg = s2v.cogvideox_5b()
    cfg.num_layers = 2
    dt = torch.bfloat16
    sd = s2v.weights.synthetic_state_dict(cfg, seed=66, device=DEV, parity=True)
    T, F, H, W = 226, 13, 60, 90
    text, ref, lat = _inputs(cfg, T, F, H, W, 67)
    lat = lat.to(dt).contiguous()
    m2, e2 = _engine(s2v, cfg, dt, sd, 2, text, ref, T, F, H, W)
    y2 = e2.forward(lat, torch.tensor([500.0, 500.0]), shared_latent=True).clone()
    torch.cuda.synchronize()
    e2.close()
    del m2, e2
    for slot in (0, 1):
        m1, e1 = _engine(s2v, cfg, dt, sd, 1, text[slot:slot + 1], ref, T, F, H, W)
        y1 = e1.forward(lat, torch.tensor([500.0]), shared_latent=True)
        torch.cuda.synchronize()
        assert torch.isfinite(y1.float()).all()
        assert torch.equal(y1[0], y2[slot]), f"slot {slot} at the full token count"
        e1.close()
        del m1, e1


# ---- two processes on one device ------------------------------------------------------------------------------------------------------------
def _pipe_case(s2v):
    cfg = s2v.tiny(use_rope=True, heads=2, layers=2, text_dim=64, temb=64)
    sd = s2v.weights.synthetic_state_dict(cfg, seed=68, parity=True)
    g = torch.Generator().manual_seed(69)
    F, H, W, T = 3, 8, 12, 5
    kw = dict(prompt_embeds=torch.randn(1, T, 64, generator=g), negative_prompt_embeds=torch.randn(1, T, 64, generator=g),
              ref_img_states=torch.randn(1, 1, 16, H, W, generator=g) * 0.7, height=H * 8, width=W * 8, num_frames=(F - 1) * 4 + 1,
              num_inference_steps=4, guidance_scale=6.0, latents=torch.randn(1, F, 16, H, W, generator=g), output_type="latent", return_dict=False)
    return cfg, sd, kw


def _run_pipe(s2v, sched, cfg_parallel, use_graph):
    cfg, sd, kw = _pipe_case(s2v)
    m = s2v.HipCogVideoXTransformer3DModel(cfg, torch.bfloat16, DEV)
    m.load_state_dict(sd)
    sch = (s2v.CogVideoXDDIMScheduler if sched == "ddim" else s2v.CogVideoXDPMScheduler)(snr_shift_scale=1.0)
    pipe = s2v.S2VPipeline(m, sch)
    out = pipe(**kw, generator=torch.Generator().manual_seed(70), use_graph=use_graph, cfg_parallel=cfg_parallel)[0]
    torch.cuda.synchronize()
    res = out.float().cpu()
    m.engine.close()
    return res


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    s2v = importlib.import_module("disentangled-subject-to-vid_amd")
    torch.cuda.set_device(0)
    s2v.dist.init_from_env("gloo")
    cp = s2v.dist.CfgPair(native=False)
    res = {}
    for sched in ("ddim", "dpm"):
        for graph in (False, True):
            res[(sched, graph)] = _run_pipe(s2v, sched, cp, graph).numpy()
    q.put((rank, cp.slot, res))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_device_pipeline_bitwise(s2v):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=600) for _ in range(2)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [g[1] for g in got] == [0, 1]
    for sched in ("ddim", "dpm"):
        exp = _run_pipe(s2v, sched, None, True).numpy()
        for graph in (False, True):
            a, b = got[0][2][(sched, graph)], got[1][2][(sched, graph)]
            assert (a == b).all(), f"{sched} graph={graph}: the two ranks of the pair differ"
            assert (a == exp).all(), f"{sched} graph={graph}: CFG-parallel differs from the one-process pipeline"


def test_split_entry_points_reject_what_they_cannot_run(s2v):
    """argument validation of the CFG-parallel C ABI: a B = 2 geometry (the pair already lives on this GPU), a slot outside {0, 1}, null pointers,
    and attn_p_format 'auto' (settled per engine from its own census: the two ranks of a pair could diverge) fail loudly with a message"""
    import ctypes

    L = s2v._lib
    cfg = s2v.tiny(use_rope=True, heads=2, layers=1, text_dim=64, temb=64)
    sd = s2v.weights.synthetic_state_dict(cfg, seed=71, parity=True)
    T, F, H, W = 5, 2, 8, 12
    text, ref, lat = _inputs(cfg, T, F, H, W, 72)
    lat = lat.bfloat16().contiguous()
    sch = s2v.CogVideoXDDIMScheduler(snr_shift_scale=1.0)
    sch.set_timesteps(3)
    coef = sch.coef(sch.timesteps[0], torch.bfloat16, 6.0)
    m2, e2 = _engine(s2v, cfg, torch.bfloat16, sd, 2, text, ref, T, F, H, W)
    with pytest.raises(L.S2VError, match="B = 1"):
        e2.denoise_split_begin(lat, 999.0, coef, 0)
    with pytest.raises(L.S2VError, match="B = 1"):
        e2.denoise_split_end(lat)
    e2.close()
    m1, e1 = _engine(s2v, cfg, torch.bfloat16, sd, 1, text[1:2], ref, T, F, H, W)
    with pytest.raises(L.S2VError, match="slot"):
        e1.denoise_split_begin(lat, 999.0, coef, 2)
    with pytest.raises(L.S2VError, match="pending"):     # the step's coefficients are uploaded by begin: no end without its begin ...
        e1.denoise_split_end(lat)
    e1.denoise_split_begin(lat, 999.0, coef, 1)
    e1.denoise_split_end(lat)
    with pytest.raises(L.S2VError, match="pending"):     # ... and one end per begin
        e1.denoise_split_end(lat)
    dsch = s2v.CogVideoXDPMScheduler(snr_shift_scale=1.0)
    dsch.set_timesteps(3)
    e1.denoise_split_begin(lat, 999.0, dsch.coef(dsch.timesteps[0], None, True, torch.bfloat16, 6.0), 1)
    with pytest.raises(L.S2VError, match="DPM"):         # a DPM step brings its noise and x0 history to the end
        e1.denoise_split_end(lat)
    assert L.lib().s2v_denoise_split_begin(e1._h, None, 999.0, ctypes.byref(coef), 0, 0, L.stream_ptr()) != 0
    assert L.lib().s2v_cfg_pair(e1._h, None, None) != 0
    assert L.lib().s2v_rccl_allgather(None, None, None, 16, L.stream_ptr()) != 0 and b"s2v_rccl_allgather" in L.lib().s2v_last_error()
    assert L.lib().s2v_denoise_step_cfg_parallel(e1._h, None, 0, L.ptr(lat), 999.0, ctypes.byref(coef), None, None, 0, L.stream_ptr()) != 0
    pair = e1.cfg_pair()
    assert tuple(pair.shape) == (2, F, 16, H, W) and pair.dtype == torch.bfloat16
    e1.close()
    import copy
    ca = copy.copy(cfg)
    ca.attn_p_format = "auto"
    ma, ea = _engine(s2v, ca, torch.bfloat16, sd, 1, text[1:2], ref, T, F, H, W)
    with pytest.raises(L.S2VError, match="auto"):
        ea.denoise_split_begin(lat, 999.0, coef, 1)
    ea.close()
