"""Drop-in object for the text-encoder seam: `text_encoder(input_ids)[0]` and `.dtype` as used by
pipelines/cogvideo/pipeline_cogvideox.py:197-237; built like src/inference.py:183-189 builds `T5EncoderModel`.  The stack runs
in libs2v_hip.so (csrc/t5.hip); the relative-position bucket table is evaluated here with the same torch ops transformers
uses (models/t5/modeling_t5.py `_relative_position_bucket`), like the RoPE tables of tables.py."""
import ctypes
import math
from dataclasses import dataclass
from types import SimpleNamespace

import torch

from . import _lib

_P, _I32, _I64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64


@dataclass
class T5Config:
    """T5 v1.1 XXL encoder (the text_encoder/config.json that ships with CogVideoX)"""
    vocab_size: int = 32128
    d_model: int = 4096
    d_kv: int = 64
    num_heads: int = 64
    d_ff: int = 10240
    num_layers: int = 24
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128
    layer_norm_epsilon: float = 1e-6


class T5ConfigC(ctypes.Structure):
    _fields_ = [("vocab_size", _I32), ("d_model", _I32), ("d_kv", _I32), ("num_heads", _I32), ("d_ff", _I32),
                ("num_layers", _I32), ("relative_attention_num_buckets", _I32), ("relative_attention_max_distance", _I32),
                ("dtype", _I32), ("force_simple", _I32), ("layer_norm_epsilon", ctypes.c_float), ("reserved", _I32 * 5)]


_lib.register_sigs({
    "s2v_t5_create": [ctypes.POINTER(T5ConfigC), ctypes.POINTER(_P)],
    "s2v_t5_load_weight": [_P, ctypes.c_char_p, _P, ctypes.POINTER(_I64), _I32, _I32, _P],
    "s2v_t5_finalize": [_P],
    "s2v_t5_weight_arena": [_P, ctypes.POINTER(_P), ctypes.POINTER(_I64)],
    "s2v_t5_mark_weights_loaded": [_P],
    "s2v_t5_rel_table": [_P, ctypes.POINTER(_P)],
    "s2v_t5_set_position_bias": [_P, _P, _I32, _I32, _P],
    "s2v_t5_encode": [_P, _P, _I32, _I32, _P, _P],
})


def relative_position_bucket(relative_position, num_buckets=32, max_distance=128):
    """T5Attention._relative_position_bucket, bidirectional (the encoder's) form, same op sequence"""
    num_buckets //= 2
    relative_buckets = (relative_position > 0).to(torch.long) * num_buckets
    relative_position = torch.abs(relative_position)
    max_exact = num_buckets // 2
    is_small = relative_position < max_exact
    relative_position_if_large = max_exact + (
        torch.log(relative_position.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)
    ).to(torch.long)
    relative_position_if_large = torch.min(relative_position_if_large,
                                           torch.full_like(relative_position_if_large, num_buckets - 1))
    return relative_buckets + torch.where(is_small, relative_position, relative_position_if_large)


def position_buckets(T, num_buckets=32, max_distance=128):
    ctx = torch.arange(T, dtype=torch.long)[:, None]
    mem = torch.arange(T, dtype=torch.long)[None, :]
    return relative_position_bucket(mem - ctx, num_buckets, max_distance)  # [T(query), T(key)]


class HipT5EncoderModel:
    def __init__(self, cfg: T5Config = None, dtype=torch.bfloat16, device="cuda:0", force_simple=False):
        cfg = cfg or T5Config()
        if dtype not in _lib.DTYPE_OF:
            raise _lib.S2VError(f"unsupported T5 dtype {dtype}")
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        self.config = SimpleNamespace(**cfg.__dict__)
        torch.cuda.set_device(self.device)
        c = T5ConfigC()
        for f in ("vocab_size", "d_model", "d_kv", "num_heads", "d_ff", "num_layers", "relative_attention_num_buckets",
                  "relative_attention_max_distance"):
            setattr(c, f, getattr(cfg, f))
        c.dtype, c.force_simple, c.layer_norm_epsilon = _lib.DTYPE_OF[dtype], int(force_simple), cfg.layer_norm_epsilon
        self._h = ctypes.c_void_p()
        _lib.check(_lib.lib().s2v_t5_create(ctypes.byref(c), ctypes.byref(self._h)))
        self._bias_for = None

    def close(self):
        if self._h:
            lib = _lib.lib()
            lib.s2v_t5_destroy.argtypes = [_P]
            lib.s2v_t5_destroy.restype = None
            lib.s2v_t5_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def load_state_dict(self, sd, strict=True):
        keep = []
        for k, t in sd.items():
            if k == "encoder.embed_tokens.weight" and "shared.weight" in sd:
                continue  # tied copy
            t = t.to(self.device)
            if t.dtype not in _lib.DTYPE_OF:
                t = t.float()
            t = t.contiguous()
            shape = (_I64 * t.ndim)(*t.shape)
            _lib.check(_lib.lib().s2v_t5_load_weight(self._h, k.encode(), _lib.ptr(t), shape, t.ndim,
                                                     _lib.DTYPE_OF[t.dtype], _lib.stream_ptr()))
            keep.append(t)
        torch.cuda.synchronize(self.device)
        _lib.check(_lib.lib().s2v_t5_finalize(self._h))
        self._bias_for = None

    def weight_arenas(self):
        """[uint8 CUDA tensor aliasing the packed weights] (dist.broadcast_components)"""
        from .engine import _ArenaView

        p, n = ctypes.c_void_p(), ctypes.c_int64()
        _lib.check(_lib.lib().s2v_t5_weight_arena(self._h, ctypes.byref(p), ctypes.byref(n)))
        return [torch.as_tensor(_ArenaView(p.value, n.value), device=self.device)]

    def mark_weights_loaded(self):
        _lib.check(_lib.lib().s2v_t5_mark_weights_loaded(self._h))

    def _set_bias(self, B, T):
        if self._bias_for == (B, T):
            return
        p = _P()
        _lib.check(_lib.lib().s2v_t5_rel_table(self._h, ctypes.byref(p)))
        nb, H = self.cfg.relative_attention_num_buckets, self.cfg.num_heads
        table = torch.empty((nb, H), dtype=self.dtype, device=self.device)
        # device -> device copy out of the handle-owned table
        src = _ArenaView(p.value, nb * H * table.element_size())
        table.view(torch.uint8).view(-1).copy_(torch.as_tensor(src, device=self.device))
        torch.cuda.current_stream(self.device).synchronize()
        buckets = position_buckets(T, nb, self.cfg.relative_attention_max_distance).to(self.device)
        bias = table[buckets].permute(2, 0, 1).contiguous()  # compute_bias: [H, T, T]
        _lib.check(_lib.lib().s2v_t5_set_position_bias(self._h, _lib.ptr(bias), B, T, _lib.stream_ptr()))
        torch.cuda.synchronize(self.device)
        self._bias_for = (B, T)

    def __call__(self, input_ids, attention_mask=None, **kw):
        if attention_mask is not None:
            raise NotImplementedError("the pipeline passes no attention mask (pipeline_cogvideox.py:227)")
        ids = input_ids.to(self.device, torch.long).contiguous()
        B, T = ids.shape
        self._set_bias(B, T)
        out = torch.empty((B, T, self.cfg.d_model), dtype=self.dtype, device=self.device)
        _lib.check(_lib.lib().s2v_t5_encode(self._h, _lib.ptr(ids), B, T, _lib.ptr(out), _lib.stream_ptr()))
        return (out,)


class _ArenaView:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
