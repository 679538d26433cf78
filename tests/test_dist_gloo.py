"""world_size-2 CPU (gloo) coverage of the replica path (dist.py): weight-arena broadcast, prompt sharding, result
gather.  The engine is a stand-in whose "denoise" is the CPU oracle's scheduler arithmetic on the broadcast weights,
so a rank that did not receive the arena produces different latents."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeEngine:
    def __init__(self, nbytes):
        self.arena = torch.zeros(nbytes, dtype=torch.uint8)
        self.loaded = False

    def weight_arena(self):
        return self.arena

    def weight_arenas(self):
        return [self.arena]

    def mark_weights_loaded(self):
        self.loaded = True


def _run_prompt(eng, pid, prompt):
    from oracle import sched_ref

    assert eng.loaded
    w = eng.arena[:4096].float().reshape(1, 4, 16, 8, 8) / 255.0
    ac = sched_ref.alphas_cumprod(1.0)
    g = torch.Generator().manual_seed(int(prompt))
    lat = torch.randn(1, 4, 16, 8, 8, generator=g)
    for t in sched_ref.trailing_timesteps(3):
        lat, _ = sched_ref.ddim_step(ac, 3, (w - 0.5) * lat, int(t), lat)
        lat = lat.float()
    return lat


def _load_rank0(eng):
    g = torch.Generator().manual_seed(1)
    eng.arena.copy_(torch.randint(0, 256, eng.arena.shape, generator=g, dtype=torch.uint8))
    eng.loaded = True


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    d = importlib.import_module("disentangled-subject-to-vid_amd.dist")
    r, w, _ = d.init_from_env("gloo")
    assert (r, w) == (rank, world)
    prompts = [11, 22, 33, 44, 55]
    assert d.shard_prompts(5, rank, world) == [p for p in range(5) if p % world == rank]
    res = d.run_replicas(lambda: FakeEngine(3 * (1 << 20) + 17), _load_rank0, prompts, _run_prompt)
    if rank == 0:
        q.put({k: v.numpy() for k, v in res.items()})
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_replicas_match_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process expectation
    eng = FakeEngine(3 * (1 << 20) + 17)
    _load_rank0(eng)
    assert sorted(got) == [0, 1, 2, 3, 4]
    for pid, prompt in enumerate([11, 22, 33, 44, 55]):
        np.testing.assert_array_equal(got[pid], _run_prompt(eng, pid, prompt).numpy())


class FakeVAE:
    """same replica protocol as HipAutoencoderKLCogVideoX: built with the encoder half by default, weight_arenas() /
    mark_weights_loaded() called WITHOUT arguments (as dist.broadcast_components calls them) cover the halves the object has"""

    def __init__(self, with_encoder=True):
        self.dec = torch.zeros(70001, dtype=torch.uint8)
        self.enc = torch.zeros(513, dtype=torch.uint8) if with_encoder else None
        self.loaded, self.enc_loaded = False, False

    def weight_arenas(self, with_encoder=None):
        if with_encoder is None:
            with_encoder = self.enc is not None
        return [self.dec] + ([self.enc] if with_encoder else [])

    def mark_weights_loaded(self, with_encoder=None):
        if with_encoder is None:
            with_encoder = self.enc is not None
        self.loaded = True
        self.enc_loaded = self.enc_loaded or with_encoder


def _worker_multi(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    d = importlib.import_module("disentangled-subject-to-vid_amd.dist")
    d.init_from_env("gloo")

    def load0(parts):
        eng, vae, t5 = parts
        _load_rank0(eng)
        g = torch.Generator().manual_seed(2)
        for a in vae.weight_arenas() + t5.weight_arenas():
            a.copy_(torch.randint(0, 256, a.shape, generator=g, dtype=torch.uint8))
        vae.loaded = t5.loaded = True

    def run(parts, pid, prompt):
        eng, vae, t5 = parts
        assert eng.loaded and vae.loaded and t5.loaded
        lat = _run_prompt(eng, pid, prompt)
        # "decode" and "text encode" stand-ins read the other components' arenas: a rank that missed one broadcast differs
        return lat * (1.0 + vae.dec[:64].float().mean() / 255.0) + vae.enc[:8].float().sum() + t5.arena[-16:].float().sum()

    prompts = [7, 8, 9]  # uneven over 2 ranks: rank 0 runs prompts 0 and 2, rank 1 runs prompt 1
    res = d.run_replicas(lambda: (FakeEngine(1 << 20), FakeVAE(), FakeEngine(300007)), load0, prompts, run)
    if rank == 0:
        q.put({k: v.numpy() for k, v in res.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_multi_component_broadcast_and_uneven_prompts():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_multi, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    eng, vae, t5 = FakeEngine(1 << 20), FakeVAE(), FakeEngine(300007)
    _load_rank0(eng)
    g = torch.Generator().manual_seed(2)
    for a in vae.weight_arenas() + t5.weight_arenas():
        a.copy_(torch.randint(0, 256, a.shape, generator=g, dtype=torch.uint8))
    assert sorted(got) == [0, 1, 2]
    for pid, prompt in enumerate([7, 8, 9]):
        lat = _run_prompt(eng, pid, prompt)
        exp = lat * (1.0 + vae.dec[:64].float().mean() / 255.0) + vae.enc[:8].float().sum() + t5.arena[-16:].float().sum()
        np.testing.assert_array_equal(got[pid], exp.numpy())


def test_single_process_paths():
    d = importlib.import_module("disentangled-subject-to-vid_amd.dist")
    assert d.broadcast_arena(torch.zeros(10, dtype=torch.uint8)) == 0
    assert d.broadcast_components([FakeEngine(10)]) == 0
    assert d.gather_results({0: torch.ones(2)})[0].sum() == 2
    assert d.shard_prompts(8, 3, 8) == [3]


class FakeVAEFlags(FakeVAE):
    """the protocol of HipAutoencoderKLCogVideoX since round 4: the sender's per-arena `loaded` flags travel with the
    hand-off, so a half the sender never filled is neither sent nor marked on the receiver"""

    def arenas_loaded(self):
        return [self.loaded] + ([self.enc_loaded] if self.enc is not None else [])

    def mark_weights_loaded(self, with_encoder=None, loaded=None):
        if loaded is None:
            return super().mark_weights_loaded(with_encoder)
        self.loaded = self.loaded or loaded[0]
        self.enc_loaded = self.enc_loaded or (len(loaded) > 1 and loaded[1])


def _worker_flags(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    d = importlib.import_module("disentangled-subject-to-vid_amd.dist")
    d.init_from_env("gloo")
    vae = FakeVAEFlags()
    if rank == 0:  # the sender loaded `decoder.*` only
        vae.dec.fill_(7)
        vae.loaded = True
    moved = d.broadcast_components([vae], 0)
    q.put((rank, moved, bool(vae.loaded), bool(vae.enc_loaded), int(vae.dec[0]), int(vae.enc.sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_receiver_marks_only_the_halves_the_sender_had_loaded():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_flags, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, moved, loaded, enc_loaded, d0, esum in got:
        assert moved == 70001, "only the decoder arena travels"
        assert loaded and not enc_loaded and d0 == 7 and esum == 0
