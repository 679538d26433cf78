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


def test_single_process_paths():
    d = importlib.import_module("disentangled-subject-to-vid_amd.dist")
    assert d.broadcast_arena(torch.zeros(10, dtype=torch.uint8)) == 0
    assert d.gather_results({0: torch.ones(2)})[0].sum() == 2
    assert d.shard_prompts(8, 3, 8) == [3]
