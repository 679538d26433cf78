"""The replica hand-off (dist.run_replicas / broadcast_components) with REAL components on a one-GPU box: two
processes share cuda:0, the process group is gloo carrying device tensors (the same code path the 8-GPU run takes with backend
"nccl" = RCCL), rank 0 alone loads weights, rank 1 is filled ONLY through the arena broadcast.  Every rank then runs its prompts
through the transformer engine (hipGraph denoise steps), the VAE -- encode of a reference image AND decode, so the encoder half must
have travelled (it did not) -- and the T5 encoder.  The gathered results must equal a single-process run bit for bit.
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
PROMPTS = [3, 4, 5]  # uneven over two ranks: rank 0 runs prompts 0 and 2, rank 1 prompt 1


def _configs(s2v):
    tcfg = s2v.tiny(use_rope=True, heads=2, layers=2, text_dim=64, temb=64)
    vcfg = s2v.VAEConfig(block_out_channels=(16, 16, 32, 32), layers_per_block=1, norm_num_groups=4, latent_channels=16,
                         sample_height=96, sample_width=160, scaling_factor=0.7, temporal_compression_ratio=4)
    t5cfg = s2v.T5Config(vocab_size=100, d_model=64, d_kv=64, num_heads=2, d_ff=128, num_layers=2)
    return tcfg, vcfg, t5cfg


def _make(s2v):
    tcfg, vcfg, t5cfg = _configs(s2v)
    return (s2v.S2VEngine(tcfg, torch.bfloat16, DEV), s2v.HipAutoencoderKLCogVideoX(vcfg, torch.bfloat16, DEV),
            s2v.HipT5EncoderModel(t5cfg, torch.bfloat16, DEV))


def _load(s2v, parts):
    eng, vae, t5 = parts
    tcfg, vcfg, t5cfg = _configs(s2v)
    eng.load_state_dict(s2v.weights.synthetic_state_dict(tcfg, seed=1, parity=True))
    sd = dict(s2v.weights.synthetic_vae_state_dict(vcfg, seed=3))
    sd.update(s2v.weights.synthetic_vae_encoder_state_dict(vcfg, seed=4))
    vae.load_state_dict(sd)
    t5.load_state_dict(s2v.weights.synthetic_t5_state_dict(t5cfg, seed=6, gain=0.6))


def _run_prompt(s2v, parts, pid, prompt):
    """a miniature of video_generate.inference: T5 -> text embeddings, VAE encode of the reference image, three denoise steps
    on the captured graph, VAE decode; returns everything a wrong or missing weight would change"""
    eng, vae, t5 = parts
    g = torch.Generator().manual_seed(int(prompt))
    ids = torch.randint(1, 100, (2, 5), generator=g).to(DEV)
    text = t5(ids)[0]                                                    # [2, 5, 64]
    img = (torch.rand(1, 3, 1, 64, 96, generator=g) * 2 - 1).bfloat16().to(DEV)
    ref = vae.encode(img).latent_dist.sample(torch.Generator().manual_seed(int(prompt) + 100)) * vae.config.scaling_factor
    F, H, W = 2, 8, 12
    lat = torch.randn(1, F, 16, H, W, generator=g).bfloat16().to(DEV).contiguous()
    eng.set_geometry(2, 5, F, H, W)
    eng.prepare_tables(H * 8, W * 8)
    eng.set_conditioning(text, ref.permute(0, 2, 1, 3, 4).contiguous())
    sch = s2v.CogVideoXDDIMScheduler(snr_shift_scale=1.0)
    sch.set_timesteps(3)
    for t in sch.timesteps:
        eng.denoise_step(lat, float(t), sch.coef(t, torch.bfloat16, 6.0), use_graph=True)
    video = vae.decode_latents(lat)
    torch.cuda.synchronize()
    return torch.cat([text.float().flatten(), ref.float().flatten(), lat.float().flatten(), video.float().flatten()]).cpu()


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    s2v = importlib.import_module("disentangled-subject-to-vid_amd")
    torch.cuda.set_device(0)
    s2v.dist.init_from_env("gloo")
    moved = {}
    orig = s2v.dist.broadcast_components

    def counted(components, src=0):
        moved["bytes"] = orig(components, src)
        return moved["bytes"]

    s2v.dist.broadcast_components = counted
    res = s2v.dist.run_replicas(lambda: _make(s2v), lambda parts: _load(s2v, parts), PROMPTS,
                                lambda parts, pid, prompt: _run_prompt(s2v, parts, pid, prompt))
    if rank == 0:
        q.put(({k: v.numpy() for k, v in res.items()}, moved.get("bytes", 0)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_device_real_components_bitwise(s2v):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, moved = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    parts = _make(s2v)
    _load(s2v, parts)
    arena_bytes = sum(a.numel() for c in parts for a in c.weight_arenas())
    assert len(parts[1].weight_arenas()) == 2, "the VAE must hand over decoder AND encoder"
    assert moved == arena_bytes and moved > 1 << 16
    assert sorted(got) == [0, 1, 2]
    for pid, prompt in enumerate(PROMPTS):
        exp = _run_prompt(s2v, parts, pid, prompt).numpy()
        assert got[pid].shape == exp.shape
        assert (got[pid] == exp).all(), f"prompt {pid} (ran on rank {pid % 2}) differs from the single-process run"
