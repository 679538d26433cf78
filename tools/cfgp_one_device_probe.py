"""Where the time of a CFG-parallel step goes when BOTH ranks of a pair share one GPU (gloo): begin / exchange / end timed separately, against the
same loop with the exchange replaced by a local copy.  2B at 9 x 256 x 256 (a 6 ms half step)."""
import importlib, os, socket, sys, time
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, port):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    s2v = importlib.import_module("disentangled-subject-to-vid_amd")
    torch.cuda.set_device(0)
    s2v.dist.init_from_env("gloo")
    cp = s2v.dist.CfgPair(native=False)
    cfg = s2v.cogvideox_2b()
    dev, dt = "cuda:0", torch.bfloat16
    eng = s2v.S2VEngine(cfg, dt, dev)
    eng.load_state_dict(s2v.weights.synthetic_state_dict(cfg, seed=1, device=dev))
    F, H, W, T = 3, 32, 32, 226
    g = torch.Generator(device=dev).manual_seed(5)
    text = torch.randn(2, T, 4096, generator=g, device=dev)
    ref = torch.randn(1, 1, 16, H, W, generator=g, device=dev)
    lat = torch.randn(1, F, 16, H, W, generator=g, device=dev).to(dt).contiguous()
    eng.set_geometry(1, T, F, H, W); eng.prepare_tables(H * 8, W * 8); eng.set_conditioning(text[cp.slot:cp.slot + 1], ref)
    sch = s2v.CogVideoXDDIMScheduler(snr_shift_scale=cfg.snr_shift_scale); sch.set_timesteps(50)
    coefs = [sch.coef(t, dt, 6.0) for t in sch.timesteps]
    for mode in ("local copy", "gloo exchange", "gloo exchange, sync before"):
        for graph in (True, False):
            tb = tx = te = 0.0
            for i in range(8):
                if i == 3:
                    torch.cuda.synchronize(); dist.barrier(); t_all = time.perf_counter(); tb = tx = te = 0.0
                t0 = time.perf_counter()
                eng.denoise_split_begin(lat, float(sch.timesteps[i]), coefs[i], cp.slot, use_graph=graph)
                if mode.endswith("sync before"):
                    torch.cuda.synchronize()
                t1 = time.perf_counter()
                if mode == "local copy":
                    pair = eng.cfg_pair(); pair[1 - cp.slot].copy_(pair[cp.slot])
                else:
                    cp.exchange(eng)
                t2 = time.perf_counter()
                eng.denoise_split_end(lat)
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                tb += t1 - t0; tx += t2 - t1; te += t3 - t2
            tot = (time.perf_counter() - t_all) / 5
            if rank == 0:
                print(f"{mode:28s} graph={graph!s:5s}: step {tot * 1e3:8.2f} ms  (begin {tb / 5 * 1e3:7.2f}  exchange {tx / 5 * 1e3:7.2f}  end+sync {te / 5 * 1e3:7.2f})", flush=True)
    dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(worker, args=(port,), nprocs=2, join=True)
