"""Build recipe for libs2v_hip.so (hipcc, gfx950 only).  `python build.py` or `build_library()`."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libs2v_hip.so")
SOURCES = ["api.hip", "gemm.hip", "gemm_g4.hip", "gemm_g4t.hip", "gemm_g4f.hip", "gemm_f32m.hip", "attention.hip", "attention_f32m.hip", "attention_q4.hip", "elementwise.hip", "vae.hip", "vae_api.hip", "t5.hip", "rccl.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-result", "-Wno-unused-value", "-Wno-inline-asm"]
# the HBM-bound kernels mirror the reference's separately-rounded elementwise ops: no fma contraction there
# (hipcc defaults to -ffp-contract=fast); the scheduler step is bit-exact against the CPU reference because of it
EXTRA = {"attention_q4.hip": ["-fno-slp-vectorize"], "gemm_g4.hip": ["-fno-slp-vectorize"], "gemm_g4t.hip": ["-fno-slp-vectorize"], "gemm_g4f.hip": ["-fno-slp-vectorize"], "elementwise.hip": ["-ffp-contract=off", "-DS2V_TU_FP_CONTRACT_OFF"], "vae.hip": ["-ffp-contract=off", "-DS2V_TU_FP_CONTRACT_OFF"],
         "t5.hip": ["-ffp-contract=off", "-DS2V_TU_FP_CONTRACT_OFF"]}


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _digest(paths, extra=()):
    """content hash of the inputs of a build step: file bytes + the command-line pieces.  Staleness is decided by CONTENT, not by
    modification time: a fresh checkout (new mtimes everywhere, or old ones) next to a binary that travelled from another tree
    rebuilds exactly when the sources differ from the ones the binary was made from."""
    h = hashlib.sha256()
    for e in extra:
        h.update(str(e).encode() + b"\0")
    for p in sorted(paths):
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


DIAG_LIB = os.path.join(HERE, "libs2v_hip_diag.so")
# generated asm (outputs are committed): generator -> EVERY file it writes (a missing one of them re-runs the generator).
# tests/test_host_cpu.py holds this table to the truth twice: every `#include "*.inc"` / `*_regs.h` under csrc/ must appear here, and each
# generator, re-run into a scratch directory (S2V_GEN_OUT), must write exactly these files with exactly the committed bytes.
GENERATORS = [("gen_attn_q4.py", ["attn_q4_body.inc", "attn_q4h_body.inc", "attn_q4f_body.inc", "attn_q4fh_body.inc", "attn_q4hh_body.inc", "attn_q8_body.inc", "attn_q4_regs.h"]),
              ("gen_gemm_g4.py", ["gemm_g4_body.inc", "gemm_g4_body_f16.inc", "gemm_g4_sk_sum.inc", "gemm_g4_regs.h"]),
              ("gen_gemm_g4t.py", ["gemm_g4t_body_gelu.inc", "gemm_g4t_body_bias.inc", "gemm_g4t_body_qknorm.inc", "gemm_g4t_regs.h"]),
              ("gen_gemm_g4f.py", ["gemm_g4f_body_a3.inc", "gemm_g4f_body_mx.inc", "gemm_g4f_regs.h"])]


def build_library(force=False, verbose=True, diag=False):
    """diag=False: the product library (exports exactly what include/s2v_hip.h declares).
    diag=True: libs2v_hip_diag.so = the same sources with -DS2V_DIAG (A/B reference kernels, stall accounting, ablations and
    their knobs); only tools/ and the race-screen test load it.
    force (or S2V_FORCE_BUILD=1 in the environment): recompile every translation unit and relink.  Otherwise a step runs when the
    sha256 of its inputs (sources, every header / generated .inc, flags) differs from the stamp written beside its output:
    `<lib>.stamp` travels with the .so (git-ignored like it), the per-object stamps live in build/."""
    force = force or os.environ.get("S2V_FORCE_BUILD", "0") == "1"
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build_diag" if diag else "build")
    LIB = DIAG_LIB if diag else globals()["LIB"]
    FLAGS = globals()["FLAGS"] + (["-DS2V_DIAG"] if diag else [])
    gens = GENERATORS
    headers = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith((".h", ".inc"))]
    headers.append(os.path.join(HERE, "..", "include", "s2v_hip.h"))
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    gen_srcs = [os.path.join(CSRC, g) for g, _ in gens]
    whole = _digest(srcs + headers + gen_srcs, FLAGS + [f"{k}:{v}" for k, v in sorted(EXTRA.items())])
    if not force and os.path.exists(LIB) and _read(LIB + ".stamp") == whole:
        if verbose:
            print(f"up to date (content hash {whole[:12]}): {LIB}", flush=True)
        return LIB
    os.makedirs(objdir, exist_ok=True)
    for g, outs in gens:
        gp, ops = os.path.join(CSRC, g), [os.path.join(CSRC, o) for o in outs]
        stamp = os.path.join(objdir, g + ".stamp")
        d = _digest([gp])
        missing = [o for o in ops if not os.path.exists(o)]
        if force or missing or (_read(stamp) != d and any(_stale(o, [gp]) for o in ops)):
            if verbose:
                print(f"{sys.executable} {gp}", flush=True)
            subprocess.check_call([sys.executable, gp])
            with open(stamp, "w") as f:  # written only after a regeneration: a stamp next to outputs it did not produce proved nothing
                f.write(d)
    headers = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith((".h", ".inc"))]
    headers.append(os.path.join(HERE, "..", "include", "s2v_hip.h"))
    whole = _digest(srcs + headers + gen_srcs, FLAGS + [f"{k}:{v}" for k, v in sorted(EXTRA.items())])
    objs = []
    procs = []
    for sp in srcs:
        src = os.path.basename(sp)
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc] + FLAGS + EXTRA.get(src, []) + ["-c", sp, "-o", obj]
        d = _digest([sp] + headers, cmd[1:-3])
        if force or not os.path.exists(obj) or _read(obj + ".stamp") != d:
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, obj, d, subprocess.Popen(cmd)))
    for src, obj, d, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
        with open(obj + ".stamp", "w") as f:
            f.write(d)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(LIB + ".stamp", "w") as f:
        f.write(whole)
    if verbose:
        print(f"built {len(procs)} of {len(objs)} translation units, linked {LIB} (content hash {whole[:12]})", flush=True)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, diag="--diag" in sys.argv))
