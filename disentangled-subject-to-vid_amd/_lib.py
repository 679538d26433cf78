"""ctypes binding of libs2v_hip.so (C ABI: include/s2v_hip.h).  There is NO fallback: if the HIP library is
missing or a call fails, this module raises."""
import ctypes
import os

import torch  # noqa: F401  (must be imported first so the process-wide HIP runtime is torch's libamdhip64)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("S2V_LIB") or os.path.join(_HERE, "libs2v_hip.so")  # S2V_LIB: an experiment build of the SAME library (tools/ab_step.sh)

DTYPE_F32, DTYPE_BF16, DTYPE_F16 = 0, 1, 2
TORCH_DTYPE = {DTYPE_F32: torch.float32, DTYPE_BF16: torch.bfloat16, DTYPE_F16: torch.float16}
DTYPE_OF = {torch.float32: DTYPE_F32, torch.bfloat16: DTYPE_BF16, torch.float16: DTYPE_F16}


class ModelConfigC(ctypes.Structure):
    _fields_ = [
        ("num_layers", ctypes.c_int32), ("num_heads", ctypes.c_int32), ("in_channels", ctypes.c_int32),
        ("out_channels", ctypes.c_int32), ("patch_size", ctypes.c_int32), ("time_embed_dim", ctypes.c_int32),
        ("text_embed_dim", ctypes.c_int32), ("use_rope", ctypes.c_int32), ("dtype", ctypes.c_int32),
        ("norm_eps", ctypes.c_float), ("force_simple", ctypes.c_int32), ("weight_format", ctypes.c_int32),
        ("lora_adaln_scope", ctypes.c_int32), ("attn_p_format", ctypes.c_int32), ("reserved", ctypes.c_int32 * 2),
    ]


class SchedCoefC(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("guidance", ctypes.c_float)] + [
        (n, ctypes.c_float) for n in ("c_x0_x", "c_x0_v", "a_t", "b_t", "m1", "m2", "m3", "m4", "mn", "pad")
    ]


class S2VError(RuntimeError):
    pass


_lib = None

_P, _I32, _I64, _F = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
_SIGS = {
    "s2v_create": [ctypes.POINTER(ModelConfigC), ctypes.POINTER(_P)],
    "s2v_load_weight": [_P, ctypes.c_char_p, _P, ctypes.POINTER(_I64), _I32, _I32, _P],
    "s2v_merge_lora": [_P, ctypes.c_char_p, _P, _P, _I32, _F, _P],
    "s2v_finalize_weights": [_P, _P],
    "s2v_weight_arena": [_P, ctypes.POINTER(_P), ctypes.POINTER(_I64)],
    "s2v_weight_slot": [_P, ctypes.c_char_p, ctypes.POINTER(_I64), ctypes.POINTER(_I64), ctypes.POINTER(_I64), ctypes.POINTER(_I64)],
    "s2v_mark_weights_loaded": [_P],
    "s2v_set_geometry": [_P, _I32, _I32, _I32, _I32, _I32],
    "s2v_fp8_qk_active": [_P, ctypes.POINTER(_I32)],
    "s2v_set_rope": [_P, _P, _P, _P],
    "s2v_set_pos_embed": [_P, _P, _P],
    "s2v_set_conditioning": [_P, _P, _P, _P],
    "s2v_transformer_forward": [_P, _P, _I64, _P, _P, _P],
    "s2v_block_forward": [_P, _I32, _P, _P, _P, _P, _P, _P, _P, _P],
    "s2v_attn_forward": [_P, _I32, _P, _P, _P, _P, _P],
    "s2v_sched_step": [_P, ctypes.POINTER(SchedCoefC), _P, _I32, _P, _P, _P, _P, _I64, _I32, _P],
    "s2v_denoise_step": [_P, _P, _F, ctypes.POINTER(SchedCoefC), _P, _P, _I32, _P],
    "s2v_last_noise_pred": [_P, ctypes.POINTER(_P)],
    "s2v_denoise_split_begin": [_P, _P, _F, ctypes.POINTER(SchedCoefC), _I32, _I32, _P],
    "s2v_cfg_pair": [_P, ctypes.POINTER(_P), ctypes.POINTER(_I64)],
    "s2v_denoise_split_end": [_P, _P, _P, _P, _P],
    "s2v_denoise_step_cfg_parallel": [_P, _P, _I32, _P, _F, ctypes.POINTER(SchedCoefC), _P, _P, _I32, _P],
    "s2v_profile_enable": [_P, _I32],
    "s2v_profile_read": [_P, ctypes.POINTER(_F), ctypes.POINTER(_I32), _I32],
    "s2v_profile_read_clocks": [_P, ctypes.POINTER(_F), _I32],
    "s2v_op_linear": [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P],
    "s2v_op_ff_fp8": [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P],
    "s2v_op_mod_gemv": [_P, _P, _P, _P, _I32, _I32, _I64, _I32, _I32, _P],
    "s2v_op_attention": [_P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P],
    "s2v_attn_slow_stats": [_P, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), _I32],
    "s2v_set_attn_p_format": [_P, _I32],
    "s2v_op_attention_fp8qk": [_P, _P, _P, _I64, _P, _I32, _I32, _I32, _P],
    "s2v_op_linear_fp8": [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _I64, _P],
}
# replicas over RCCL behind the C ABI (csrc/rccl.hip); RCCL itself is bound at first use
_SIGS.update({
    "s2v_rccl_available": [],
    "s2v_rccl_unique_id": [_P],
    "s2v_rccl_comm_create": [_P, _I32, _I32, ctypes.POINTER(_P)],
    "s2v_rccl_bcast": [_P, _P, _I64, _I32, _P],
    "s2v_rccl_allgather": [_P, _P, _P, _I64, _P],
    "s2v_bcast_weights": [_P, _P, _I32, _P],
})
# VAE entry points are registered by vae.py through register_sigs()


def register_sigs(sigs):
    _SIGS.update(sigs)
    if _lib is not None:
        _apply_sigs(_lib, sigs)


def _apply_sigs(lib, sigs):
    for name, argtypes in sigs.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int


def lib():
    """Load the shared library (once).  Raises S2VError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise S2VError(f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` (hipcc, gfx950). "
                           "There is no CPU fallback.")
        l = ctypes.CDLL(LIB_PATH)
        _apply_sigs(l, _SIGS)
        l.s2v_last_error.restype = ctypes.c_char_p
        l.s2v_version.restype = ctypes.c_char_p
        l.s2v_destroy.argtypes = [_P]
        l.s2v_destroy.restype = None
        l.s2v_rccl_comm_destroy.argtypes = [_P]
        l.s2v_rccl_comm_destroy.restype = None
        _lib = l
    return _lib


_diag = None
DIAG_LIB_PATH = os.path.join(_HERE, "libs2v_hip_diag.so")


def diag_lib():
    """libs2v_hip_diag.so: the same sources built with -DS2V_DIAG (A/B reference kernels, stall accounting, ablations and their
    knobs s2v_set_gemm_impl / s2v_set_attn_variant / s2v_debug_read / s2v_attn_debug_read).  Only tools/ and the race-screen
    test load it (`python disentangled-subject-to-vid_amd/build.py --diag`); the product never does."""
    global _diag
    if _diag is None:
        if not os.path.exists(DIAG_LIB_PATH):
            raise S2VError(f"{DIAG_LIB_PATH} is missing: build it with `python {os.path.join(_HERE, 'build.py')} --diag`")
        l = ctypes.CDLL(os.environ.get("S2V_DIAG_LIB", DIAG_LIB_PATH))  # S2V_DIAG_LIB: an experiment build (tools/g4_ablate.sh)
        _apply_sigs(l, {k: v for k, v in _SIGS.items() if k.startswith("s2v_op_")})
        l.s2v_last_error.restype = ctypes.c_char_p
        for name in ("s2v_set_gemm_impl", "s2v_set_attn_variant"):
            getattr(l, name).argtypes = [ctypes.c_int]
        _diag = l
    return _diag


def check(rc):
    if rc != 0:
        raise S2VError(f"s2v call failed ({rc}): {lib().s2v_last_error().decode()}")


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise S2VError("tensor must live on the GPU")
    if not t.is_contiguous():
        raise S2VError("tensor must be contiguous")
    return ctypes.c_void_p(t.data_ptr())
