"""Would running the two samples of the CFG pair on two HIP streams (tails of one sample's kernels filled by the other's, HBM-bound
kernels beside MFMA-bound ones) beat the single B = 2 launch sequence?  Two engines with B = 1 each (weights duplicated for the
probe) against one engine with B = 2, same C3 geometry, forward only."""
import importlib, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
s2v = importlib.import_module("disentangled-subject-to-vid_amd")
DEV = "cuda:0"
cfg = s2v.cogvideox_5b()
cfg.num_layers = int(os.environ.get("LAYERS", "8"))
F, H, W, T = 13, 60, 90, 226
sd = s2v.weights.synthetic_state_dict(cfg, seed=1, device=DEV)


def make(B):
    e = s2v.S2VEngine(cfg, torch.bfloat16, DEV)
    e.load_state_dict(sd)
    e.set_geometry(B, T, F, H, W)
    e.prepare_tables(480, 720)
    g = torch.Generator(device=DEV).manual_seed(2)
    e.set_conditioning(torch.randn(B, T, 4096, generator=g, device=DEV), torch.randn(1, 1, 16, H, W, generator=g, device=DEV))
    return e


lat = torch.randn(1, F, 16, H, W, device=DEV).bfloat16()
e2 = make(2)
ts2 = torch.tensor([500.0, 500.0])
ts1 = torch.tensor([500.0])


def timed(fn, n=4):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


t_b2 = timed(lambda: e2.forward(lat, ts2, shared_latent=True))
print(f"one engine, B = 2, one stream: {t_b2:.1f} ms per forward ({cfg.num_layers} layers)")
del e2
ea, eb = make(1), make(1)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def pair():
    sa.wait_stream(torch.cuda.current_stream()); sb.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(sa):
        ea.forward(lat, ts1)
    with torch.cuda.stream(sb):
        eb.forward(lat, ts1)
    torch.cuda.current_stream().wait_stream(sa); torch.cuda.current_stream().wait_stream(sb)


t_one = timed(lambda: ea.forward(lat, ts1))
t_pair = timed(pair)
print(f"one engine, B = 1: {t_one:.1f} ms; two engines B = 1 on two streams: {t_pair:.1f} ms per pair  (B = 2 single stream {t_b2:.1f})")
