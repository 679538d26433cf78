"""The library's own RCCL entry points (include/s2v_hip.h `s2v_rccl_*`, `s2v_bcast_weights`; csrc/rccl.hip): SURVEY.md section 8b
proposed `s2v_bcast_weights(rccl_comm)` so that the C ABI alone can replicate a model (smaller 13).  No reference code
(single-process reference).  CPU: binding and argument checks; GPU: a one-rank communicator carries the real
ncclBroadcast call path (RCCL refuses two ranks on one device, and the build environment has one GPU per call)."""
import ctypes
import importlib

import pytest
import torch


def test_unique_id_and_argument_checks(s2v):
    l = s2v.lib()
    a, b = ctypes.create_string_buffer(128), ctypes.create_string_buffer(128)
    assert l.s2v_rccl_unique_id(a) == 0 and l.s2v_rccl_unique_id(b) == 0
    assert a.raw != b.raw and any(a.raw)
    assert l.s2v_rccl_unique_id(None) != 0
    h = ctypes.c_void_p()
    assert l.s2v_rccl_comm_create(a, 2, 2, ctypes.byref(h)) != 0  # rank outside the world
    assert b"bad argument" in l.s2v_last_error()
    assert l.s2v_rccl_bcast(None, None, 0, 0, None) != 0
    assert l.s2v_bcast_weights(None, None, 0, None) != 0
    l.s2v_rccl_comm_destroy(None)  # no-op


@pytest.mark.gpu
def test_one_rank_communicator_broadcasts_in_place(s2v):
    dist = importlib.import_module("disentangled-subject-to-vid_amd.dist")
    dev = "cuda:0"
    cfg = s2v.tiny(use_rope=True, heads=2, layers=2, text_dim=64, temb=64)
    eng = s2v.S2VEngine(cfg, torch.bfloat16, dev)
    eng.load_state_dict(s2v.weights.synthetic_state_dict(cfg, seed=1))
    before = eng.weight_arena().clone()
    comm = dist.RcclComm(rank=0, world=1, device=dev)
    comm.broadcast_weights(eng, 0)
    assert comm.broadcast(eng.weight_arena(), 0) == before.numel()
    big = torch.arange(300 * (1 << 20) // 8 + 3, dtype=torch.int64, device=dev)  # more than one 256-MiB chunk, ragged tail
    ref = big.clone()
    assert comm.broadcast(big.view(torch.uint8), 0) == big.numel() * 8
    torch.cuda.synchronize()
    assert torch.equal(eng.weight_arena(), before) and torch.equal(big, ref)
    l = s2v.lib()
    assert l.s2v_rccl_bcast(comm._h, s2v._lib.ptr(big), 8, 1, s2v._lib.stream_ptr()) != 0  # root outside the world
    comm.close()
