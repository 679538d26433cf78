"""What the per-step exchange of a CFG-parallel pair costs on ONE GPU over gloo (two processes on cuda:0; the functional path of tests / one-GPU boxes --
the product path is s2v_rccl_allgather or the nccl backend on two GPUs): dist.all_gather on device tensors against a host-staged all_gather.
    python tools/cfgp_exchange_probe.py [bytes_per_rank]"""
import os, socket, sys, time
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, port, nbytes):
    os.environ.update(RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    pair = torch.zeros(2, nbytes // 2, dtype=torch.bfloat16, device="cuda:0")
    pair[rank].fill_(rank + 1)
    host = torch.zeros(2, nbytes // 2, dtype=torch.bfloat16).pin_memory()
    for name in ("device tensors", "host-staged"):
        for it in range(13):
            if it == 3:
                torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
            if name == "device tensors":
                mine = pair[rank].clone()
                dist.all_gather([pair[0], pair[1]], mine)
            else:
                host[rank].copy_(pair[rank], non_blocking=True)
                torch.cuda.current_stream().synchronize()
                dist.all_gather([host[0], host[1]], host[rank].clone())
                pair.copy_(host, non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        ok = bool((pair[0] == 1).all() and (pair[1] == 2).all())
        if rank == 0:
            print(f"gloo all_gather of {nbytes} bytes per rank, {name:15s}: {dt * 1e3:8.2f} ms per exchange  (correct {ok})", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    nbytes = int(sys.argv[1]) if len(sys.argv) > 1 else 2246400
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(worker, args=(port, nbytes), nprocs=2, join=True)
