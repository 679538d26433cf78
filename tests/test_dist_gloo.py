"""world_size-2 CPU (gloo) coverage of the replica path (dist.py): weight-arena broadcast, prompt sharding, result
gather.  The engine is a stand-in whose "denoise" is the CPU oracle's scheduler arithmetic on the broadcast weights,
so a rank that did not receive the arena produces different latents."""
import importlib
import json
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


# ---- bench.py under rank failures: what the first real 8-GPU run would hit -----------------------------------
def _bench(args, env_extra, timeout):
    import subprocess
    import sys
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    return r, time.time() - t0


def test_bench_dist_selftest_two_gloo_ranks_replicate_the_arena():
    r, _ = _bench(["--gpus", "2", "--dist-selftest"], {"S2V_SELFTEST_BYTES": str(8 << 20)}, 180)
    assert r.returncode == 0, r.stderr[-1500:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line == {"dist_selftest": "ok", "ranks": 2, "bytes": 8 << 20}


def test_bench_exits_nonzero_when_a_rank_dies_mid_broadcast():
    """rank 1 is killed half-way through the chunked broadcast: the launcher sees its exit code, ends rank 0 (which sits in the next
    collective) and bench.py returns non-zero long before any collective timeout"""
    r, dt = _bench(["--gpus", "2", "--dist-selftest"], {"S2V_SELFTEST_KILL_RANK": "1", "S2V_SELFTEST_BYTES": str(8 << 20),
                                                        "S2V_BENCH_BCAST_TIMEOUT_S": "120"}, 180)
    assert r.returncode == 9, (r.returncode, r.stderr[-1500:])
    assert "exited with code 9; ending the other 1 rank(s)" in r.stderr
    assert "dist_selftest" not in r.stdout and dt < 60, dt


def test_bench_watchdog_ends_a_broadcast_whose_peer_stopped_taking_part():
    """rank 1 stays alive but never enters the remaining collectives: nothing dies by itself, so the per-rank Watchdog has to fire -- it
    reports the phase, the progress and the environment, and exits with dist.EXIT_WATCHDOG"""
    r, dt = _bench(["--gpus", "2", "--dist-selftest"], {"S2V_SELFTEST_STALL_RANK": "1", "S2V_SELFTEST_BYTES": str(8 << 20),
                                                        "S2V_BENCH_BCAST_TIMEOUT_S": "6"}, 180)
    assert r.returncode == 86, (r.returncode, r.stderr[-1500:])
    assert "[s2v watchdog] rank" in r.stderr and "still running after 6 s" in r.stderr and "bytes enqueued" in r.stderr
    assert dt < 60, dt


def test_supervise_ends_the_survivors_of_a_failed_rank():
    import subprocess
    import sys
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    try:
        import importlib

        bench = importlib.import_module("bench")
    finally:
        sys.path.remove(root)
    procs = [subprocess.Popen([sys.executable, "-c", "import time; time.sleep(600)"]),
             subprocess.Popen([sys.executable, "-c", "import sys, time; time.sleep(0.5); sys.exit(4)"])]   # 4 = bench.py's "non-finite outputs" exit
    t0 = time.time()
    assert bench.supervise(procs, grace_s=2.0) == 4
    assert time.time() - t0 < 30 and all(p.poll() is not None for p in procs)


def test_replica_mismatch_is_raised_on_every_rank():
    """a rank whose component has a different number of weight arenas than the sender's used to raise alone and leave the others in the
    arena broadcast; the verdict is now agreed on collectively -- both ranks raise, nothing is sent"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mismatch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert got == {0: "ok", 1: "ok"}, got


def _mismatch_worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        s2v = importlib.import_module("disentangled-subject-to-vid_amd")
        s2v.dist.init_from_env("gloo")

        class Comp:
            def __init__(self, n):
                self.a = [torch.zeros(1024, dtype=torch.uint8) for _ in range(n)]

            def weight_arenas(self):
                return self.a

            def arenas_loaded(self):
                return [True] * len(self.a)

            def mark_weights_loaded(self, loaded=None):
                raise AssertionError("nothing must be marked loaded after a mismatch")

        comp = Comp(2 if rank == 0 else 3)
        try:
            s2v.dist.broadcast_components([comp], 0)
            q.put((rank, "no error"))
        except RuntimeError as e:
            q.put((rank, "ok" if "replica mismatch" in str(e) and "every rank raises" in str(e) else f"wrong error: {e}"))
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        q.put((rank, f"exception: {e!r}"))


# ---- CFG-parallel host logic (dist.cfg_pair_layout / dist.CfgPair) -----------------------------------------------
class FakeCfgEngine:
    """stand-in for a B = 1 S2VEngine on CPU tensors: "forward" = a deterministic function of (latents, this rank's text half); the CFG combine and
    the DDIM step are the oracle's (oracle.sched_ref), i.e. what custom_cogvideox_pipe.py:266-296 computes on the pair"""

    def __init__(self, text_half):
        self.text = text_half
        self.pair = torch.zeros(2, 2, 4, 6, 6)
        self.coef = None

    def cfg_pair(self):
        return self.pair

    def denoise_split_begin(self, latents, timestep, coef, slot, use_graph=False):
        self.coef = (timestep, coef)
        self.pair[slot] = torch.tanh(latents[0] * self.text.mean() + self.text.std() * 0.1 * timestep / 1000.0)

    def denoise_split_end(self, latents, x0_hist=None, noise=None):
        from oracle import sched_ref

        t, (ac, n, g) = self.coef
        v = sched_ref.cfg_combine(self.pair.clone(), g)
        out, _ = sched_ref.ddim_step(ac, n, v, int(t), latents)
        latents.copy_(out.float())


def _cfg_run(pairs_text, slot_engines_step, steps=3):
    from oracle import sched_ref

    ac = sched_ref.alphas_cumprod(1.0)
    lat = torch.randn(1, 2, 4, 6, 6, generator=torch.Generator().manual_seed(7))
    for t in sched_ref.trailing_timesteps(steps):
        slot_engines_step(lat, float(t), (ac, steps, 6.0))
    return lat


def _cfg_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    d = importlib.import_module("disentangled-subject-to-vid_amd.dist")
    d.init_from_env("gloo")
    cp = d.CfgPair(native=False)
    assert (cp.pair, cp.slot, cp.peer, cp.pairs) == (rank // 2, rank % 2, rank ^ 1, world // 2)
    # video p of pair p: embeddings [negative | positive] seeded by p; this rank holds half `slot`
    text = torch.randn(2, 5, 8, generator=torch.Generator().manual_seed(100 + cp.pair))
    eng = FakeCfgEngine(text[cp.slot:cp.slot + 1])
    lat = _cfg_run(None, lambda x, t, coef: cp.step(eng, x, t, coef))
    q.put((rank, cp.pair, lat.numpy(), eng.pair.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_cfg_parallel_pairs_exchange_halves_and_agree_bitwise(world):
    """world 2: one video on two ranks; world 4: two videos, two sub-groups (composes with replicas).  Both ranks of a pair end with identical
    latents, equal to ONE process computing both halves itself"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cfg_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=120) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for pair in range(world // 2):
        text = torch.randn(2, 5, 8, generator=torch.Generator().manual_seed(100 + pair))
        e0, e1 = FakeCfgEngine(text[0:1]), FakeCfgEngine(text[1:2])

        def both(x, t, coef):
            e0.denoise_split_begin(x, t, coef, 0)
            e1.denoise_split_begin(x, t, coef, 1)
            e0.pair[1] = e1.pair[1]
            e0.denoise_split_end(x)

        exp = _cfg_run(None, both).numpy()
        a, b = got[2 * pair], got[2 * pair + 1]
        assert a[1] == b[1] == pair
        assert (a[2] == b[2]).all() and (a[3] == b[3]).all(), "the two ranks of a pair must hold identical latents and pair buffers"
        assert (a[2] == exp).all(), "pair result differs from one process computing both halves"
    if world == 4:
        assert not (got[0][2] == got[2][2]).all(), "the two videos must differ"


def test_cfg_pair_layout_rejects_odd_worlds():
    d = importlib.import_module("disentangled-subject-to-vid_amd.dist")
    assert d.cfg_pair_layout(5, 8) == (2, 1, 4, 4)
    for bad in (1, 3, 7):
        with pytest.raises(ValueError):
            d.cfg_pair_layout(0, bad)
    with pytest.raises(ValueError):
        d.cfg_pair_layout(2, 2)


def _cfgp_replica_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    d = importlib.import_module("disentangled-subject-to-vid_amd.dist")
    d.init_from_env("gloo")

    def run_prompt(eng, pair, pid, prompt):
        assert eng.loaded
        text = torch.randn(2, 5, 8, generator=torch.Generator().manual_seed(int(prompt)))
        fe = FakeCfgEngine(text[pair.slot:pair.slot + 1] + eng.arena[:8].float().mean() / 255.0)
        return _cfg_run(None, lambda x, t, coef: pair.step(fe, x, t, coef))

    res = d.run_cfg_parallel(lambda: FakeEngine(1 << 16), _load_rank0, [11, 22, 33], run_prompt)
    if rank == 0:
        q.put({k: v.numpy() for k, v in res.items()})
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_cfg_parallel_pairs_compose_with_replicas():
    """dist.run_cfg_parallel over four ranks = two pairs: three prompts (pair 0 runs prompts 0 and 2, pair 1 prompt 1), weights from rank 0 only, results
    gathered from the slot-0 ranks; every result equals one process computing both halves"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cfgp_replica_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(got) == [0, 1, 2]
    eng = FakeEngine(1 << 16)
    _load_rank0(eng)
    shift = eng.arena[:8].float().mean() / 255.0
    for pid, prompt in enumerate([11, 22, 33]):
        text = torch.randn(2, 5, 8, generator=torch.Generator().manual_seed(prompt)) + shift
        e0, e1 = FakeCfgEngine(text[0:1]), FakeCfgEngine(text[1:2])

        def both(x, t, coef):
            e0.denoise_split_begin(x, t, coef, 0)
            e1.denoise_split_begin(x, t, coef, 1)
            e0.pair[1] = e1.pair[1]
            e0.denoise_split_end(x)

        assert (got[pid] == _cfg_run(None, both).numpy()).all(), pid


def _same_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    d = importlib.import_module("disentangled-subject-to-vid_amd.dist")
    d.init_from_env("gloo")
    cp = d.CfgPair(native=False)
    lat = torch.randn(1, 2, 4, 6, 6, generator=torch.Generator().manual_seed(3))
    cp.assert_same(latents=lat, ref=None, text=torch.ones(2, 3))          # identical: passes
    verdict = "no error"
    try:
        cp.assert_same(latents=lat + (1e-3 if rank == 1 else 0.0), text=torch.ones(2, 3))   # rank 1 drew its own latents
    except RuntimeError as e:
        verdict = str(e)
    q.put((rank, verdict))
    dist.barrier()
    dist.destroy_process_group()


def test_cfg_pair_refuses_ranks_that_start_from_different_latents():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_same_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):   # BOTH ranks raise, naming what differs
        assert "different ['latents']" in got[r], got[r]
