"""RCCL smoke on one GPU (world size 1): process-group init over nccl (= RCCL), the chunked arena broadcast, barrier, all_reduce."""
import os, importlib, torch, torch.distributed as dist, sys
sys.path.insert(0, os.getcwd())
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ["RANK"] = "0"; os.environ["WORLD_SIZE"] = "1"; os.environ["LOCAL_RANK"] = "0"
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
rank, world = 0, 1
cfg = s2v.tiny(use_rope=True, heads=2, layers=2, text_dim=64, temb=64)
eng = s2v.S2VEngine(cfg, torch.bfloat16, "cuda:0")
eng.load_state_dict(s2v.weights.synthetic_state_dict(cfg, seed=1))
n = s2v.dist.broadcast_arena(eng.weight_arena(), 0)
dist.barrier()
t = torch.tensor([1.5], device="cuda:0", dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("rccl ok", rank, world, n, t.item())
dist.destroy_process_group()
