"""CPU tests of the product's host side: tables vs the reference's golden vectors, scheduler scalars (the kernel
arithmetic is emulated here in numpy with the product's coefficients and must reproduce the reference bit for bit),
the C ABI surface, argument validation.  No compute call reaches the GPU library here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden


# ------------------------------------------------------------------------------------------------ tables
def test_tables_match_reference(s2v):
    g = load_golden("tables.npz")
    for snr in (1.0, 3.0):
        np.testing.assert_array_equal(s2v.tables.alphas_cumprod(snr), g[f"alphas_{snr}"])
    for n in (3, 10, 50):
        np.testing.assert_array_equal(s2v.tables.trailing_timesteps(n), g[f"timesteps_{n}"])
    c, s = s2v.tables.rope_tables(256, 256, 3)
    np.testing.assert_array_equal(c, g["rope_cos_256x256"])
    np.testing.assert_array_equal(s, g["rope_sin_256x256"])
    for hw in ((480, 720), (720, 1280)):
        c, s = s2v.tables.rope_tables(hw[0], hw[1], 13)
        np.testing.assert_array_equal(c[::61], g[f"rope_cos_{hw[0]}x{hw[1]}_rows61"])
        np.testing.assert_array_equal(s[::61], g[f"rope_sin_{hw[0]}x{hw[1]}_rows61"])
        assert c.shape == ((hw[0] // 16) * (hw[1] // 16) * 14, 64)
    np.testing.assert_array_equal(s2v.tables.sincos_table(128, 4, 4, 2), g["sincos_128_4x4x2"])
    np.testing.assert_array_equal(s2v.tables.sincos_table(192, 4, 6, 3), g["sincos_192_6x4x3"])
    np.testing.assert_array_equal(s2v.tables.sincos_table(1920, 16, 16, 3)[::7], g["sincos_1920_16x16x3_rows7"])
    assert s2v.tables.crop_region(45, 80) == ((2, 0), (27, 45))   # 720x1280: fractional h grid
    assert s2v.tables.crop_region(16, 16) == ((0, 8), (30, 38))   # 256x256


# ------------------------------------------------------------------------------------------------ schedulers
def _rnd(x, bf16):
    """ET<T>::rnd of the kernels: identity for fp32, round-trip through bf16 otherwise (numpy emulation)"""
    if not bf16:
        return x.astype(np.float32)
    return torch.from_numpy(x.astype(np.float32)).to(torch.bfloat16).float().numpy()


def _emulate_kernel(c, npred, x, x0_old, noise, bf16):
    """numpy mirror of sched_step_k (csrc/elementwise.hip) -- every op a separately rounded fp32 op"""
    f = np.float32
    u, cc = npred[0:1], npred[1:2]
    v = (u + f(c.guidance) * (cc - u).astype(f)).astype(f)
    x0 = (_rnd(x * f(c.c_x0_x), bf16) - (f(c.c_x0_v) * v).astype(f)).astype(f)
    if c.kind == 0:
        prev = (_rnd(f(c.a_t) * x, bf16) + (f(c.b_t) * x0).astype(f)).astype(f)
    else:
        d = x0
        if c.kind == 2:
            d = ((f(c.m3) * x0).astype(f) - (f(c.m4) * x0_old).astype(f)).astype(f)
        prev = ((_rnd(f(c.m1) * x, bf16) - (f(c.m2) * d).astype(f)).astype(f) + _rnd(f(c.mn) * noise, bf16)).astype(f)
    return _rnd(prev, bf16), x0


@pytest.mark.parametrize("kind", ["ddim", "dpm"])
@pytest.mark.parametrize("dt_name", ["f32", "bf16"])
@pytest.mark.parametrize("n_steps", [10, 50])
def test_scheduler_coefficients_reproduce_reference_bits(s2v, kind, dt_name, n_steps):
    g = load_golden(f"sched_{kind}_{dt_name}_{n_steps}.npz")
    bf16 = dt_name == "bf16"
    dt = torch.bfloat16 if bf16 else torch.float32
    cls = s2v.CogVideoXDDIMScheduler if kind == "ddim" else s2v.CogVideoXDPMScheduler
    sch = cls(snr_shift_scale=float(g["snr"]))
    sch.set_timesteps(n_steps)
    np.testing.assert_array_equal(sch.timesteps.numpy(), g["timesteps"])
    ids = list(g["step_ids"])
    with np.errstate(all="ignore"):
        for i in ids:
            t = sch.timesteps[i]
            if kind == "ddim":
                c = sch.coef(t, dt, 6.0)
                x0_old = noise = None
            else:
                first = i == 0
                if not first and (i - 1) not in ids:
                    continue
                c = sch.coef(t, sch.timesteps[i - 1] if i > 0 else None, first, dt, 6.0)
                x0_old = g[f"x0_{i-1}"] if not first else None
                noise = g[f"n2_{i}"] if c.kind == 2 else g[f"n1_{i}"]
            prev, x0 = _emulate_kernel(c, g[f"noise_pred_{i}"], g[f"lat_in_{i}"], x0_old, noise, bf16)
            np.testing.assert_array_equal(x0, g[f"x0_{i}"], err_msg=f"x0 step {i}")
            np.testing.assert_array_equal(prev, g[f"lat_out_{i}"], err_msg=f"latents step {i}")


def test_scheduler_scalar_semantics_switch(s2v):
    """"cpu" (default): the scalars that multiply bf16 tensors are rounded to bf16 first, as CPU torch materialises a 0-dim scalar
    in the tensor dtype (what the goldens above pin); "cuda": they stay fp32, as torch's CUDA kernels fetch them (opmath)."""
    cpu = s2v.CogVideoXDDIMScheduler(snr_shift_scale=1.0)
    gpu = s2v.CogVideoXDDIMScheduler(snr_shift_scale=1.0, scalar_semantics="cuda")
    for s in (cpu, gpu):
        s.set_timesteps(50)
    differs = 0
    for i in (0, 7, 23, 49):
        t = cpu.timesteps[i]
        a, b = cpu.coef(t, torch.bfloat16, 6.0), gpu.coef(t, torch.bfloat16, 6.0)
        for f in ("c_x0_x", "a_t"):
            va, vb = getattr(a, f), getattr(b, f)
            assert va == float(torch.tensor(vb).bfloat16().float())      # cpu = cuda rounded to bf16
            assert vb == float(np.float32(vb))                            # cuda = an fp32 value
            differs += va != vb
        assert (a.c_x0_v, a.b_t) == (b.c_x0_v, b.b_t)                   # the other two were fp32 in both
        f32a, f32b = cpu.coef(t, torch.float32, 6.0), gpu.coef(t, torch.float32, 6.0)
        assert (f32a.c_x0_x, f32a.a_t) == (f32b.c_x0_x, f32b.a_t)       # no difference for an fp32 model
    assert differs > 0
    with pytest.raises(ValueError):
        s2v.CogVideoXDPMScheduler(scalar_semantics="tpu")


def test_scheduler_protocol_surface(s2v):
    s = s2v.CogVideoXDDIMScheduler(snr_shift_scale=1.0)
    assert s.order == 1 and s.init_noise_sigma == 1.0
    x = torch.zeros(3)
    assert s.scale_model_input(x, 5) is x
    with pytest.raises(ValueError):
        s.set_timesteps(2000)
    with pytest.raises(ValueError):
        s.coef(999, torch.float32)  # set_timesteps not called
    with pytest.raises(NotImplementedError):
        s2v.CogVideoXDDIMScheduler(prediction_type="epsilon")


# ------------------------------------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol(s2v):
    hdr = open(os.path.join(ROOT, "include", "s2v_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(s2v_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = ctypes.CDLL(s2v._lib.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    # ... and nothing else: the library is built with -fvisibility=hidden, diagnostics live in libs2v_hip_diag.so
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", s2v._lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))
    # everything the Python binding uses is declared in the header
    bound = set(s2v._lib._SIGS)
    assert bound <= declared | {"s2v_mark_weights_loaded"}, bound - declared
    assert s2v.lib().s2v_version().startswith(b"s2v_hip")


def test_missing_library_fails_loudly(s2v, monkeypatch):
    monkeypatch.setattr(s2v._lib, "_lib", None)
    monkeypatch.setattr(s2v._lib, "LIB_PATH", "/nonexistent/libs2v_hip.so")
    with pytest.raises(s2v.S2VError, match="no CPU fallback"):
        s2v._lib.lib()


def test_cpu_tensor_is_rejected(s2v):
    with pytest.raises(s2v.S2VError):
        s2v._lib.ptr(torch.zeros(4))


def test_weight_key_inventory(s2v):
    cfg5 = s2v.cogvideox_5b()
    shapes = s2v.weights.state_dict_shapes(cfg5)
    n = sum(int(np.prod(s)) for s in shapes.values())
    assert abs(n - 5.570e9) < 0.01e9  # 5.570 B parameters
    cfg2 = s2v.cogvideox_2b()
    n2 = sum(int(np.prod(s)) for s in s2v.weights.state_dict_shapes(cfg2).values())
    assert abs(n2 - 1.694e9) < 0.01e9
    keys = s2v.weights.lora_target_keys(s2v.tiny())
    assert "patch_embed.proj.weight" in keys and "transformer_blocks.0.ff.net.0.proj.weight" in keys
    assert "transformer_blocks.0.norm1.linear.weight" in keys and "norm_out.linear.weight" not in keys
    # the golden tiny transformer has exactly these keys
    g = load_golden("transformer_tiny_rope.npz")
    gk = {k[2:] for k in g if k.startswith("w:")}
    assert gk == set(s2v.weights.state_dict_shapes(s2v.tiny()))


def test_seam_cache_keys_on_identity_version_and_engine_epoch():
    """transformer._Cache: a new tensor (even at a recycled address), an in-place edit, or a write to the engine from outside
    the cache (epoch bump) all invalidate; the cached tensors are kept alive so that their addresses cannot be recycled"""
    import importlib
    import weakref

    tm = importlib.import_module("disentangled-subject-to-vid_amd.transformer")

    class FakeEngine:
        def __init__(self):
            self.epoch = {"rope": 0, "cond": 0}

    e, c = FakeEngine(), tm._Cache("cond")
    a, b = torch.zeros(3), torch.zeros(3)
    assert c.changed(e, a, b)
    c.store(e, a, b)
    assert not c.changed(e, a, b)
    a.add_(1)
    assert c.changed(e, a, b)           # in-place edit
    c.store(e, a, b)
    assert c.changed(e, torch.zeros(3), b)  # another tensor object with equal shape / dtype
    e.epoch["cond"] += 1
    assert c.changed(e, a, b)           # engine state was rewritten behind the cache
    c.store(e, a, None)
    assert not c.changed(e, a, None) and c.changed(e, a, b)
    t = torch.zeros(4)
    w = weakref.ref(t)
    c.store(e, t)
    del t
    assert w() is not None              # strong reference held by the key


def test_export_to_video_writes_a_readable_file_without_imageio(s2v, tmp_path):
    """utils/export_utils.py:143-186: the frames handed to export_to_video come back out of the file (frame count, fps, size,
    content of the first frame up to JPEG loss); without imageio-ffmpeg the container is Motion-JPEG AVI"""
    vg = s2v.video_generate
    yy, xx = np.mgrid[0:48, 0:80]
    frames = np.stack([np.stack([(xx * 3 + 10 * f) % 256, (yy * 5) % 256, np.full_like(xx, 40 * f % 256)], -1) for f in range(9)]).astype(np.uint8)
    out = vg.export_to_video(frames, str(tmp_path / "clip.mp4"), fps=8)
    assert os.path.exists(out) and os.path.getsize(out) > 1000
    if out.endswith(".avi"):
        n, fps, w, h, first = vg.read_avi_info(out)
        assert (n, w, h) == (9, 80, 48) and abs(fps - 8.0) < 1e-6
        assert np.abs(first.astype(int) - frames[0].astype(int)).mean() < 6.0
    with pytest.raises(ValueError):
        vg.export_to_video(frames.astype(np.float32) / 255.0, str(tmp_path / "bad.mp4"))


def test_block_object_is_assignable_into_the_reference_module_list(s2v):
    """`transformer.transformer_blocks[i] = HipCogVideoXBlock(engine, i)`: the reference keeps its blocks in an nn.ModuleList
    (cogvideox_transformer_3d.py:315-330), which only accepts nn.Module entries; the drop-in has no parameters of its own"""
    blocks = torch.nn.ModuleList([torch.nn.Identity(), torch.nn.Identity()])
    blocks[1] = s2v.HipCogVideoXBlock(None, 1)
    assert isinstance(blocks[1], torch.nn.Module) and blocks[1].layer == 1 and not list(blocks[1].parameters())
    assert "forward" in type(blocks[1]).__dict__


# ------------------------------------------------------------------------------------------------ build recipe
def _builder():
    import importlib.util

    spec = importlib.util.spec_from_file_location("s2v_build", os.path.join(ROOT, "disentangled-subject-to-vid_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_every_included_generated_file_is_listed_in_the_build_table():
    """a generated file a translation unit includes but build.GENERATORS does not list would never re-run its generator when it is
    missing or stale (gemm_g4t_body_qknorm.inc was such a file)"""
    b = _builder()
    listed = {o for _, outs in b.GENERATORS for o in outs}
    csrc = b.CSRC
    included = set()
    for fn in os.listdir(csrc):
        if fn.endswith((".hip", ".h")):
            for m in re.finditer(r'#include\s+"([^"]+)"', open(os.path.join(csrc, fn)).read()):
                if m.group(1).endswith(".inc") or m.group(1).endswith("_regs.h"):
                    included.add(m.group(1))
    assert included, "no generated include found: the scan is broken"
    assert included <= listed, f"included but not in build.GENERATORS: {sorted(included - listed)}"
    on_disk = {fn for fn in os.listdir(csrc) if fn.endswith(".inc") or fn.endswith("_regs.h")}
    assert on_disk == listed, f"csrc/ holds {sorted(on_disk - listed)} unlisted and lacks {sorted(listed - on_disk)}"
    for g, _ in b.GENERATORS:
        assert os.path.exists(os.path.join(csrc, g))


@pytest.mark.parametrize("gen", ["gen_attn_q4.py", "gen_gemm_g4.py", "gen_gemm_g4t.py", "gen_gemm_g4f.py"])
def test_generator_reproduces_the_committed_outputs(gen, tmp_path):
    """each asm generator, re-run into a scratch directory with a clean environment, writes exactly the files the build table lists
    and exactly the committed bytes: the .inc files in the tree are what the generator in the tree produces"""
    import subprocess
    import sys

    b = _builder()
    outs = dict(b.GENERATORS)[gen]
    env = {k: v for k, v in os.environ.items() if not k.startswith(("Q4_", "G4_", "G4T_", "G4F_"))}  # the experiment knobs of the generators
    env["S2V_GEN_OUT"] = str(tmp_path)
    subprocess.check_call([sys.executable, os.path.join(b.CSRC, gen)], env=env, stdout=subprocess.DEVNULL)
    assert sorted(os.listdir(tmp_path)) == sorted(outs)
    for o in outs:
        assert open(os.path.join(tmp_path, o), "rb").read() == open(os.path.join(b.CSRC, o), "rb").read(), f"{o} differs from what {gen} writes"
