"""Data-parallel replicas of the denoise path over the GPUs of one node (the path does not
shard inside a video, every prompt is an independent unit).  One process per GPU; the only collective is the one-off
broadcast of the packed weight arena rank0 -> all (RCCL over xGMI when the backend is "nccl"), plus an optional
gather of results.  No per-step communication -- unless the caller asks for CFG-parallel (CfgPair, round 6): the two
samples of a video's CFG pair on two GPUs, one 2.2 MB all-gather per step, for single-video latency.  The host logic is backend-agnostic and covered with gloo on CPU
(tests/test_dist_gloo.py)."""
import os

import torch
import torch.distributed as dist


EXIT_WATCHDOG = 86  # a rank whose collective exceeded its deadline ends itself with this code


class Watchdog:
    """Deadline around a collective phase (the weight broadcast is the only one this path has).  A rank that is still inside the phase after
    `timeout_s` seconds writes what it was doing to stderr (rank, phase, progress callback, the RCCL / torch.distributed settings in force,
    every thread's Python stack) and ends the PROCESS with EXIT_WATCHDOG -- a peer that died or never arrived otherwise leaves the others
    blocked in the collective until an outer limit kills the job without a word.  The launcher (bench.py spawn_ranks, torchrun) sees the
    non-zero exit and takes the remaining ranks down.  No signals: a timer thread and os._exit, so it works under any launcher."""

    def __init__(self, what, timeout_s, progress=None):
        self.what, self.timeout_s, self.progress = what, float(timeout_s), progress
        self._timer = None

    def _fire(self):
        import faulthandler
        import sys

        rank = os.environ.get("RANK", "0")
        prog = ""
        try:
            prog = f"; progress: {self.progress()}" if self.progress else ""
        except Exception as e:  # the progress callback must never mask the report
            prog = f"; progress callback failed: {e!r}"
        env = {k: os.environ.get(k) for k in ("WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "NCCL_DEBUG", "NCCL_DEBUG_SUBSYS",
                                              "TORCH_NCCL_ASYNC_ERROR_HANDLING", "HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_SOCKET_IFNAME")}
        sys.stderr.write(f"[s2v watchdog] rank {rank}: '{self.what}' still running after {self.timeout_s:.0f} s{prog}\n"
                         f"[s2v watchdog] rank {rank}: environment {env}; RCCL prints its own diagnosis under NCCL_DEBUG=WARN (bench.py sets it for N > 1)\n"
                         f"[s2v watchdog] rank {rank}: exiting with code {EXIT_WATCHDOG}; Python stacks follow\n")
        sys.stderr.flush()
        try:
            faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
        finally:
            os._exit(EXIT_WATCHDOG)

    def __enter__(self):
        import threading

        if self.timeout_s > 0:
            self._timer = threading.Timer(self.timeout_s, self._fire)
            self._timer.daemon = True
            self._timer.start()
        return self

    def __exit__(self, *exc):
        if self._timer is not None:
            self._timer.cancel()
        return False


def init_from_env(backend=None, timeout_s=None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).  timeout_s (default S2V_DIST_TIMEOUT_S or 600): the process
    group's own collective timeout -- torch's NCCL watchdog then aborts a collective whose peer vanished instead of waiting for ever."""
    import datetime

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        if timeout_s is None:
            timeout_s = float(os.environ.get("S2V_DIST_TIMEOUT_S", "600"))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
    return rank, world, local


def rccl_available_everywhere():
    """(ok, bad): ok is True when EVERY rank can bind librccl through the library; bad lists (rank, reason) of those that cannot.  The
    native-broadcast decision has to be the same on all ranks, or some would enter ncclBroadcast and the others torch.distributed.broadcast.
    The probe is s2v_rccl_available: dlopen + symbol lookup only -- no unique id is drawn, so no bootstrap thread or listening socket is left
    behind on the ranks whose id would never be used."""
    from . import _lib

    ok, why = 1, ""
    try:
        _lib.check(_lib.lib().s2v_rccl_available())
    except Exception as e:
        ok, why = 0, str(e)
    if dist.is_initialized() and dist.get_world_size() > 1:
        verdicts = [None] * dist.get_world_size()
        dist.all_gather_object(verdicts, (dist.get_rank(), ok, why))
        bad = [(r, w) for r, o, w in verdicts if not o]
        return (not bad), bad
    return bool(ok), ([] if ok else [(0, why)])


def shard_prompts(num_prompts, rank, world):
    """prompt p runs on rank p mod world"""
    return [p for p in range(num_prompts) if p % world == rank]


def broadcast_arena(arena, src=0, chunk_bytes=256 << 20, progress=None):
    """replicate a finalized model: `arena` is the uint8 view of the packed weights (S2VEngine.weight_arena()).
    Chunked so that each collective is large enough to saturate an xGMI link (>= 64 MB) without needing a second
    full-size staging buffer.  progress: an optional one-element list that receives the bytes enqueued so far (a Watchdog reports it)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    flat = arena.view(-1)
    n = flat.numel()
    for off in range(0, n, chunk_bytes):
        dist.broadcast(flat[off : min(off + chunk_bytes, n)], src=src)
        if progress is not None:
            progress[0] = min(off + chunk_bytes, n)
    return n


class RcclComm:
    """the library's OWN RCCL communicator (csrc/rccl.hip: s2v_rccl_*): the weight broadcast without torch.distributed in the data
    path -- what a C / C++ host of libs2v_hip.so would do.  torch.distributed (any backend, gloo included) only carries the 128-byte
    unique id from rank 0 to the others.  One process per GPU; the communicator binds to the CURRENT device."""

    def __init__(self, rank=None, world=None, device=None, group=None):
        """group: a torch.distributed sub-group (the two ranks of a CFG pair): rank / world are then the group's, and the id travels inside it"""
        import ctypes

        from . import _lib

        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        if device is not None:
            torch.cuda.set_device(device)
        buf = ctypes.create_string_buffer(128)
        if self.rank == 0:
            _lib.check(_lib.lib().s2v_rccl_unique_id(buf))
        box = [buf.raw if self.rank == 0 else None]
        if self.world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self._h = ctypes.c_void_p()
        _lib.check(_lib.lib().s2v_rccl_comm_create(ctypes.create_string_buffer(box[0], 128), self.rank, self.world, ctypes.byref(self._h)))

    def broadcast(self, arena, src=0):
        """in place on torch's current stream; returns the bytes moved"""
        from . import _lib

        flat = arena.view(-1)
        _lib.check(_lib.lib().s2v_rccl_bcast(self._h, _lib.ptr(flat), flat.numel() * flat.element_size(), src, _lib.stream_ptr()))
        return flat.numel() * flat.element_size()

    def allgather(self, recv, rank_bytes):
        """in place on torch's current stream: rank r contributed bytes [r * rank_bytes, (r + 1) * rank_bytes) of `recv`"""
        from . import _lib

        flat = recv.view(-1).view(torch.uint8)
        send = flat[self.rank * rank_bytes:(self.rank + 1) * rank_bytes]
        _lib.check(_lib.lib().s2v_rccl_allgather(self._h, _lib.ptr(send), _lib.ptr(flat), int(rank_bytes), _lib.stream_ptr()))

    def broadcast_weights(self, engine, src=0):
        """s2v_bcast_weights: the transformer engine's arena, receivers marked loaded by the library itself"""
        from . import _lib

        _lib.check(_lib.lib().s2v_bcast_weights(engine._h, self._h, src, _lib.stream_ptr()))

    def close(self):
        from . import _lib

        if self._h:
            _lib.lib().s2v_rccl_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def cfg_pair_layout(rank, world):
    """CFG-parallel over `world` ranks (even): ranks 2p and 2p + 1 run video p together -> (pair index p, slot, peer rank, number of pairs).
    slot 0 = the unconditional half (negative prompt), slot 1 = the conditional one: the order custom_cogvideox_pipe.py:196 concatenates and
    :266 chunks.  Composes with replicas: prompt q runs on pair q mod (world // 2)."""
    if world < 2 or world % 2:
        raise ValueError(f"CFG-parallel needs an even number of ranks >= 2, got {world}")
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside 0..{world - 1}")
    return rank // 2, rank % 2, rank ^ 1, world // 2


class CfgPair:
    """The two ranks that run ONE video together: each holds a B = 1 engine with its half of the prompt embeddings; per step they
    exchange their halves of noise_pred (2.2 MB bf16 at 49 x 480 x 720) and both run CFG + the scheduler step, so their latents stay bit-identical
    with ONE collective per step.  The exchange is s2v_rccl_allgather through the library's own communicator (native=True; RCCL, one process per
    GPU) or torch.distributed.all_gather in the pair's sub-group (any backend: what the gloo tests run); neither is ever inside the captured graph."""

    def __init__(self, native=False):
        if not dist.is_initialized():
            raise RuntimeError("CfgPair needs an initialised process group")
        rank, world = dist.get_rank(), dist.get_world_size()
        self.pair, self.slot, self.peer, self.pairs = cfg_pair_layout(rank, world)
        # new_group is collective over ALL ranks, for every group, in the same order
        self.group = None
        for p in range(self.pairs):
            g = dist.new_group([2 * p, 2 * p + 1])
            if p == self.pair:
                self.group = g
        self.comm = RcclComm(group=self.group) if native else None
        self._dev = None    # torch-allocated all-gather target (nccl / CPU path of exchange)
        self._host = None   # pinned staging of the pair buffer for backends that do not move device memory themselves (exchange)

    def exchange(self, engine):
        """after denoise_split_begin: fill the peer's half of the engine's pair buffer.
        native: s2v_rccl_allgather, stream-ordered.  Backend nccl (= RCCL): torch's all_gather on the device halves, stream-ordered.  Any other
        backend (gloo: the tests, one-GPU boxes): the half travels through pinned host memory AFTER the stream has drained -- gloo's own handling of
        device tensors waits for pending GPU work by polling and took 0.8-1.4 s per call behind a 6 ms forward (tools/cfgp_one_device_probe.py);
        host-staged it is 1-2 ms."""
        pair = engine.cfg_pair()
        if self.comm is not None:
            self.comm.allgather(pair, pair[0].numel() * pair.element_size())
            return
        if pair.is_cuda and dist.get_backend(self.group) != "nccl":
            if self._host is None or self._host.shape != pair.shape or self._host.dtype != pair.dtype:
                self._host = torch.empty(pair.shape, dtype=pair.dtype).pin_memory()
            host = self._host
            host[self.slot].copy_(pair[self.slot], non_blocking=True)
            torch.cuda.current_stream().synchronize()
            dist.all_gather([host[0], host[1]], host[self.slot].clone(), group=self.group)
            pair[1 - self.slot].copy_(host[1 - self.slot], non_blocking=True)
            return
        # nccl (= RCCL), or CPU tensors on any backend: one all-gather into a torch-allocated buffer, then one copy into the context-owned pair
        # (the pair buffer is library memory wrapped through __cuda_array_interface__: it is kept out of c10d's allocator bookkeeping -- two 2.2 MB
        # device copies per step are the price, stream-ordered like the collective)
        mine = pair[self.slot].clone()
        if self._dev is None or self._dev.shape != pair.shape or self._dev.dtype != pair.dtype or self._dev.device != pair.device:
            self._dev = torch.empty(pair.shape, dtype=pair.dtype, device=pair.device)
        dist.all_gather_into_tensor(self._dev.view(-1), mine.view(-1), group=self.group)
        pair[1 - self.slot].copy_(self._dev[1 - self.slot], non_blocking=True)

    def assert_same(self, **tensors):
        """once per video: the two ranks of a pair must hold the SAME start latents, reference latent and embeddings (each computes its half of every
        step from them and both apply the scheduler step): a rank that drew its own latents would diverge silently.  Compares a checksum of every
        named tensor across the pair (one small object all-gather) and raises on BOTH ranks when they differ."""
        sums = {}
        for k, t in tensors.items():
            if t is None:
                sums[k] = None
                continue
            b = t.detach().contiguous().view(torch.uint8).to(torch.int64)
            w = torch.arange(1, b.numel() + 1, device=b.device, dtype=torch.int64) % 65521
            sums[k] = (tuple(t.shape), str(t.dtype), int((b.view(-1) * w).sum().item()))
        both = [None, None]
        dist.all_gather_object(both, sums, group=self.group)
        bad = [k for k in sums if both[0][k] != both[1][k]]
        if bad:
            raise RuntimeError(f"CFG-parallel pair {self.pair}: the two ranks were given different {bad} (slot 0: {[both[0][k] for k in bad]}, "
                               f"slot 1: {[both[1][k] for k in bad]}); pass the same tensors, or generators seeded alike, to both ranks")

    def step(self, engine, latents, timestep, coef, x0_hist=None, noise=None, use_graph=False):
        """one denoise step of the pair's video; latents (identical on both ranks) updated in place on both"""
        if self.comm is not None:
            engine.denoise_step_cfg_parallel(self.comm, self.slot, latents, timestep, coef, x0_hist, noise, use_graph)
            return
        engine.denoise_split_begin(latents, timestep, coef, self.slot, use_graph)
        self.exchange(engine)
        engine.denoise_split_end(latents, x0_hist, noise)

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None


def run_cfg_parallel(make_engine, load_weights_rank0, prompts, run_prompt, native=False):
    """run_replicas for CFG-parallel pairs: rank 0 ingests the checkpoints, every other rank receives the arenas; ranks 2p, 2p + 1
    then run prompt q TOGETHER for every q with q mod pairs == p.  run_prompt(engine, pair: CfgPair, prompt_id, prompt) -> tensor is called on BOTH
    ranks of the pair with the same arguments (it hands `pair` to S2VPipeline(cfg_parallel=pair) or drives pair.step itself) and returns the same
    tensor on both; the slot-0 rank's copy is gathered on rank 0.  An odd world size is refused (cfg_pair_layout)."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    pair = CfgPair(native=native)
    eng = make_engine()
    if rank == 0:
        load_weights_rank0(eng)
    broadcast_components(eng if isinstance(eng, (tuple, list)) else [eng], 0)
    mine = {}
    for q in range(len(prompts)):
        if q % pair.pairs == pair.pair:
            out = run_prompt(eng, pair, q, prompts[q])
            if pair.slot == 0:
                mine[q] = out
    res = gather_results(mine, 0)
    pair.close()
    return res


def broadcast_components(components, src=0, comm=None):
    """replicate every component of the path -- transformer engine, VAE (decoder [+ encoder]), T5 encoder: anything with
    .weight_arenas() -> [uint8 tensors] and .mark_weights_loaded() -- rank `src` -> all.  Returns the bytes moved.
    A component may also expose .arenas_loaded() -> [bool per arena] (the VAE: its encoder half exists on every replica but is
    only filled when the sender loaded `encoder.*` weights): the sender's flags travel first, an arena the sender never filled is
    not sent, and the receiver is told exactly which arenas now hold weights (mark_weights_loaded(loaded=flags)).
    comm: an RcclComm -- the arenas then travel through the library's own RCCL communicator (s2v_rccl_bcast) instead of
    torch.distributed.broadcast; the small flag objects still use torch.distributed."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    total = 0
    me = dist.get_rank()
    for c in components:
        arenas = c.weight_arenas()
        flags = [c.arenas_loaded() if (me == src and hasattr(c, "arenas_loaded")) else None]
        dist.broadcast_object_list(flags, src=src)
        flags = flags[0]
        # a replica built differently from the sender must stop EVERY rank before the first arena collective (a rank that raised alone left
        # the others waiting in the broadcast): the verdict is agreed on collectively and raised everywhere
        bad = flags is not None and len(flags) != len(arenas)
        verdicts = [None] * dist.get_world_size()
        dist.all_gather_object(verdicts, (me, len(arenas), bool(bad)))
        if any(v[2] for v in verdicts):
            detail = ", ".join(f"rank {r} built {n}" for r, n, b in verdicts if b)
            raise RuntimeError(f"replica mismatch: sender has {len(flags) if flags is not None else '?'} weight arenas; {detail} "
                               f"(this is rank {me}; every rank raises, no arena was sent)")
        for i, arena in enumerate(arenas):
            if flags is None or flags[i]:
                total += comm.broadcast(arena, src) if comm is not None else broadcast_arena(arena, src)
        if me != src:
            if flags is None:
                c.mark_weights_loaded()
            else:
                c.mark_weights_loaded(loaded=list(flags))
    return total


def gather_results(local_results, dst=0):
    """collect {prompt_id: tensor} from every rank on `dst` (CPU tensors; result latents are a few MB)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(local_results)
    payload = {k: v.detach().cpu() for k, v in local_results.items()}
    gathered = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(payload, gathered, dst=dst)
    if dist.get_rank() != dst:
        return None
    out = {}
    for g in gathered:
        out.update(g)
    return out


def run_replicas(make_engine, load_weights_rank0, prompts, run_prompt):
    """generic driver: rank 0 ingests the checkpoints, everybody else receives the arenas; then each rank runs its
    prompts (prompt p on rank p mod world: an uneven count leaves the last ranks one prompt short, nobody waits for anybody).
    make_engine() -> one component or a tuple / list of components (transformer engine, VAE, T5 ...), each with
    .weight_arenas() / .mark_weights_loaded(); load_weights_rank0(engine) loads + finalizes on rank 0;
    run_prompt(engine, prompt_id, prompt) -> tensor."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    eng = make_engine()
    if rank == 0:
        load_weights_rank0(eng)
    if world > 1:
        broadcast_components(eng if isinstance(eng, (tuple, list)) else [eng], 0)
    mine = {p: run_prompt(eng, p, prompts[p]) for p in shard_prompts(len(prompts), rank, world)}
    return gather_results(mine, 0)
