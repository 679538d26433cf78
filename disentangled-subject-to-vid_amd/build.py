"""Build recipe for libs2v_hip.so (hipcc, gfx950 only).  `python build.py` or `build_library()`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libs2v_hip.so")
SOURCES = ["api.hip", "gemm.hip", "gemm_g4.hip", "attention.hip", "attention_q4.hip", "elementwise.hip", "vae.hip", "vae_api.hip", "t5.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-result", "-Wno-unused-value", "-Wno-inline-asm"]
# the HBM-bound kernels mirror the reference's separately-rounded elementwise ops: no fma contraction there
# (hipcc defaults to -ffp-contract=fast); the scheduler step is bit-exact against the CPU reference because of it
EXTRA = {"attention_q4.hip": ["-fno-slp-vectorize"], "gemm_g4.hip": ["-fno-slp-vectorize"], "elementwise.hip": ["-ffp-contract=off"], "vae.hip": ["-ffp-contract=off"], "t5.hip": ["-ffp-contract=off"]}


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


DIAG_LIB = os.path.join(HERE, "libs2v_hip_diag.so")


def build_library(force=False, verbose=True, diag=False):
    """diag=False: the product library (exports exactly what include/s2v_hip.h declares).
    diag=True: libs2v_hip_diag.so = the same sources with -DS2V_DIAG (A/B reference kernels, stall accounting, ablations and
    their knobs); only tools/ and the race-screen test load it."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build_diag" if diag else "build")
    LIB = DIAG_LIB if diag else globals()["LIB"]
    FLAGS = globals()["FLAGS"] + (["-DS2V_DIAG"] if diag else [])
    os.makedirs(objdir, exist_ok=True)
    gen = os.path.join(CSRC, "gen_attn_q4.py")  # generated asm of attention_q4.hip (outputs are committed; regenerated when stale)
    if _stale(os.path.join(CSRC, "attn_q4_body.inc"), [gen]):
        subprocess.check_call([sys.executable, gen])
    gen = os.path.join(CSRC, "gen_gemm_g4.py")
    if _stale(os.path.join(CSRC, "gemm_g4_body.inc"), [gen]):
        subprocess.check_call([sys.executable, gen])
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith((".h", ".inc"))]
    headers.append(os.path.join(HERE, "..", "include", "s2v_hip.h"))
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [sp] + headers):
            cmd = [hipcc] + FLAGS + EXTRA.get(src, []) + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, diag="--diag" in sys.argv))
