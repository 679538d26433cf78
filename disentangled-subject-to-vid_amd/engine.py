"""Thin object wrapper over the C ABI context (include/s2v_hip.h).  torch is used for device memory and streams
only; every arithmetic step of the path runs in libs2v_hip.so."""
import ctypes

import numpy as np
import torch

from . import _lib, tables
from .config import TransformerConfig


class _ArenaView:
    """exposes a raw device range through __cuda_array_interface__ so torch.distributed can broadcast it"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class S2VEngine:
    def __init__(self, cfg: TransformerConfig, dtype=torch.bfloat16, device="cuda:0", force_simple=False):
        if dtype not in _lib.DTYPE_OF:
            raise _lib.S2VError(f"unsupported model dtype {dtype} (float32, bfloat16 and float16 are implemented)")
        if cfg.attention_head_dim != 64:
            raise _lib.S2VError("attention_head_dim must be 64")
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        self.D = cfg.inner_dim
        torch.cuda.set_device(self.device)
        c = _lib.ModelConfigC()
        c.num_layers, c.num_heads = cfg.num_layers, cfg.num_attention_heads
        c.in_channels, c.out_channels, c.patch_size = cfg.in_channels, cfg.out_channels, cfg.patch_size
        c.time_embed_dim, c.text_embed_dim = cfg.time_embed_dim, cfg.text_embed_dim
        c.use_rope = int(cfg.use_rotary_positional_embeddings)
        c.dtype = _lib.DTYPE_OF[dtype]
        c.norm_eps = cfg.norm_eps
        c.force_simple = int(force_simple)
        if cfg.weight_format not in (None, "fp8", "fp8-qk", "fp8-auto"):
            raise _lib.S2VError(f"unknown weight_format {cfg.weight_format!r} (None, 'fp8', 'fp8-qk' or 'fp8-auto')")
        c.weight_format = {None: 0, "fp8": 1, "fp8-qk": 2, "fp8-auto": 3}[cfg.weight_format]
        if cfg.lora_adaln_scope not in ("shipped", "intended"):
            raise _lib.S2VError(f"unknown lora_adaln_scope {cfg.lora_adaln_scope!r} ('shipped' or 'intended')")
        c.lora_adaln_scope = 1 if cfg.lora_adaln_scope == "intended" else 0
        if cfg.attn_p_format not in ("bf16", "f16", "auto"):
            raise _lib.S2VError(f"unknown attn_p_format {cfg.attn_p_format!r} ('bf16', 'f16' or 'auto')")
        c.attn_p_format = 0 if cfg.attn_p_format == "bf16" else 1
        # "auto" (opt-in): start with fp16 P (faster on smooth score distributions) and look at the slow-path census after the first denoise
        # step of a geometry (re-armed by set_geometry); more than AUTO_SLOW_FRACTION of the (wave, KV tile) pairs on the slow path -> bf16 P
        # (threshold 2^64).  The census read synchronises the device (s2v_attn_slow_stats): the deciding step runs eagerly, never inside a
        # caller's stream capture; forward() / the seam objects take no census and keep the format they find.
        self.attn_p_format = "f16" if c.attn_p_format else "bf16"
        self._attn_auto_pending = cfg.attn_p_format == "auto"
        self.attn_slow_fraction = None
        self._h = ctypes.c_void_p()
        _lib.check(_lib.lib().s2v_create(ctypes.byref(c), ctypes.byref(self._h)))
        self.geometry = None
        self._keep = []
        # bumped by every write of the engine-side RoPE / positional table ("rope") and conditioning ("cond"): the seam
        # adapters' caches (transformer._Cache) are valid only for the epoch they were stored at
        self.epoch = {"rope": 0, "cond": 0}
        self._rope_key, self.have_rope = None, None  # ensure_rope / ensure_no_rope

    def close(self):
        if self._h:
            torch.cuda.synchronize(self.device)
            _lib.lib().s2v_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights -----------------------------------------------------------------------------------------
    def load_weight(self, name, tensor):
        t = tensor.to(self.device)
        if t.dtype not in _lib.DTYPE_OF:
            t = t.float()
        t = t.contiguous()
        shape = (ctypes.c_int64 * t.ndim)(*t.shape)
        _lib.check(_lib.lib().s2v_load_weight(self._h, name.encode(), _lib.ptr(t), shape, t.ndim,
                                              _lib.DTYPE_OF[t.dtype], _lib.stream_ptr()))
        self._keep.append(t)  # alive until the stream has consumed it

    def load_state_dict(self, sd, lora=None, lora_scale=0.5):
        """sd: {reference state-dict key: tensor}; lora: {weight key: (A [r,in..], B [out,r])} merged as
        W + lora_scale * B A (src/inference.py:218-229: alpha/r = 64/128)."""
        for k, v in sd.items():
            if "pos_embedding" in k:
                continue  # non-persistent buffer, rebuilt by tables.sincos_table
            self.load_weight(k, v)
        known = None
        self.unexpected_lora_keys = []
        for k, (A, B) in (lora or {}).items():
            if known is None:
                from .weights import state_dict_shapes
                known = set(state_dict_shapes(self.cfg))
            if k not in known:
                # the reference only reports adapter keys it cannot place (src/inference.py:96-105) and carries on
                self.unexpected_lora_keys.append(k)
                continue
            A2 = A.to(self.device).float().reshape(A.shape[0], -1).contiguous()
            B2 = B.to(self.device).float().contiguous()
            _lib.check(_lib.lib().s2v_merge_lora(self._h, k.encode(), _lib.ptr(A2), _lib.ptr(B2), A2.shape[0],
                                                 float(lora_scale), _lib.stream_ptr()))
            self._keep += [A2, B2]  # the merge is stream-ordered
        if self.unexpected_lora_keys:
            print(f"Loading adapter weights led to unexpected keys not found in the model: {self.unexpected_lora_keys}")
        _lib.check(_lib.lib().s2v_finalize_weights(self._h, _lib.stream_ptr()))
        torch.cuda.synchronize(self.device)
        self._keep.clear()

    def weight_arena(self):
        """uint8 CUDA tensor aliasing the packed, context-owned weights (one RCCL broadcast replicates a model)."""
        p, n = ctypes.c_void_p(), ctypes.c_int64()
        _lib.check(_lib.lib().s2v_weight_arena(self._h, ctypes.byref(p), ctypes.byref(n)))
        return torch.as_tensor(_ArenaView(p.value, n.value), device=self.device)

    def weight_arenas(self):
        return [self.weight_arena()]

    def read_weight(self, name):
        """one state-dict tensor as it sits in the arena (after load / LoRA merge), [rows, cols] in the model dtype: a strided view
        of weight_arena() located by s2v_weight_slot (biases / norm vectors come back as [1, n]; conv weights flattened to
        [out, in * kh * kw])"""
        off, rows, cols, ld = (ctypes.c_int64() for _ in range(4))
        _lib.check(_lib.lib().s2v_weight_slot(self._h, name.encode(), ctypes.byref(off), ctypes.byref(rows), ctypes.byref(cols), ctypes.byref(ld)))
        esz = torch.empty((), dtype=self.dtype).element_size()
        flat = self.weight_arena()[off.value : off.value + ((rows.value - 1) * ld.value + cols.value) * esz].view(self.dtype)
        return flat.as_strided((rows.value, cols.value), (ld.value, 1))

    def mark_weights_loaded(self):
        _lib.check(_lib.lib().s2v_mark_weights_loaded(self._h))

    # ---- geometry / tables / conditioning ----------------------------------------------------------------------
    def _bump(self, *kinds):
        for k in kinds:
            self.epoch[k] += 1

    @property
    def fp8_qk_active(self):
        """whether QK^T of the attention runs in fp8 at the current geometry ("fp8-qk": always; "fp8-auto": from the library's token threshold on).
        The library took the decision at s2v_set_geometry and is asked for it (s2v_fp8_qk_active): nothing is re-derived here."""
        if self.cfg.weight_format not in ("fp8-qk", "fp8-auto"):
            return False
        on = ctypes.c_int32()
        _lib.check(_lib.lib().s2v_fp8_qk_active(self._h, ctypes.byref(on)))
        return bool(on.value)

    def set_geometry(self, B, T, F, H, W):
        new = self.geometry != (B, T, F, H, W)
        _lib.check(_lib.lib().s2v_set_geometry(self._h, B, T, F, H, W))
        self.geometry = (B, T, F, H, W)
        if new and self.cfg.attn_p_format == "auto":  # a new geometry is new data to the kernel: decide again, starting from fp16
            if self.attn_p_format != "f16":
                self.set_attn_p_format("f16")
            self._attn_auto_pending = True
        self._rope_key, self.have_rope = None, None
        self._bump("rope", "cond")

    def clear_rope(self):
        _lib.check(_lib.lib().s2v_set_rope(self._h, None, None, _lib.stream_ptr()))
        self._rope_key, self.have_rope = None, False
        self._bump("rope")

    # One rotary-table cache PER ENGINE: the transformer object, its 42 block objects and an AttnProcessor all write the same device
    # table, so the "did the caller's tensors change" key lives here (a per-object key made every block re-upload the tables --
    # a stream sync, two D2H copies and a host scan each -- because its neighbour's upload had invalidated it).
    def ensure_rope(self, key_tensors, build):
        """key_tensors: the caller's table tensors (or None entries), compared by identity + in-place version, held by STRONG reference
        so their storage cannot be recycled under the key; build() -> (cos, sin) for set_rope when they changed"""
        k = self._rope_key
        same = (k is not None and len(k) == len(key_tensors)
                and all((a is None and t is None) or (a is not None and t is not None and a[0] is t and a[1] == t._version)
                        for a, t in zip(k, key_tensors)))
        if not same:
            self.set_rope(*build())
            self._rope_key = tuple(None if t is None else (t, t._version) for t in key_tensors)

    def ensure_no_rope(self):
        if self.have_rope is not False:
            self.clear_rope()

    def set_rope(self, cos, sin):
        cos = cos.to(self.device, torch.float32).contiguous()
        sin = sin.to(self.device, torch.float32).contiguous()
        B, T, F, H, W = self.geometry
        n = (H // 2) * (W // 2) * (F + 1)
        if tuple(cos.shape) != (n, 64) or tuple(sin.shape) != (n, 64):
            raise _lib.S2VError(f"RoPE tables must be [{n}, 64] ([ref | video] rows)")
        _lib.check(_lib.lib().s2v_set_rope(self._h, _lib.ptr(cos), _lib.ptr(sin), _lib.stream_ptr()))
        torch.cuda.current_stream().synchronize()
        self._rope_key, self.have_rope = None, True
        self._bump("rope")

    def set_pos_embed(self, table):
        t = table.to(self.device, self.dtype).contiguous()
        _lib.check(_lib.lib().s2v_set_pos_embed(self._h, _lib.ptr(t), _lib.stream_ptr()))
        torch.cuda.current_stream().synchronize()
        self._rope_key = None
        self._bump("rope")

    def prepare_tables(self, height, width):
        """build + upload the step-invariant positional tables for a video of height x width pixels"""
        B, T, F, H, W = self.geometry
        if self.cfg.use_rotary_positional_embeddings:
            cos, sin = tables.rope_tables(height, width, F)
            self.set_rope(torch.from_numpy(cos), torch.from_numpy(sin))
        else:
            pe = tables.sincos_table(self.D, H // 2, W // 2, F, self.cfg.spatial_interpolation_scale,
                                     self.cfg.temporal_interpolation_scale)
            self.set_pos_embed(torch.from_numpy(pe))

    def set_conditioning(self, text, ref_latent):
        text = text.to(self.device, self.dtype).contiguous()
        ref = ref_latent.to(self.device, self.dtype).contiguous()
        B, T, F, H, W = self.geometry
        if tuple(text.shape) != (B, T, self.cfg.text_embed_dim):
            raise _lib.S2VError(f"text embeddings must be [{B},{T},{self.cfg.text_embed_dim}], got {tuple(text.shape)}")
        if ref.numel() != self.cfg.in_channels * H * W:
            raise _lib.S2VError("ref_img_states must be [1,1,C,H,W] with the geometry's H, W")
        _lib.check(_lib.lib().s2v_set_conditioning(self._h, _lib.ptr(text), _lib.ptr(ref), _lib.stream_ptr()))
        torch.cuda.current_stream().synchronize()
        self._bump("cond")

    # ---- compute -------------------------------------------------------------------------------------------
    def forward(self, latents, timesteps, shared_latent=False):
        """CogVideoXTransformer3DModel.forward seam: latents [B,F,C,H,W] (or [1,F,C,H,W] with shared_latent)."""
        B, T, F, H, W = self.geometry
        lat = latents.to(self.device, self.dtype).contiguous()
        t = timesteps.to(self.device, torch.float32).contiguous()
        if t.numel() != B:
            t = t.reshape(-1)[:1].expand(B).contiguous()
        out = torch.empty((B, F, self.cfg.out_channels, H, W), dtype=self.dtype, device=self.device)
        stride = 0 if shared_latent else F * self.cfg.in_channels * H * W
        _lib.check(_lib.lib().s2v_transformer_forward(self._h, _lib.ptr(lat), stride, _lib.ptr(t), _lib.ptr(out),
                                                      _lib.stream_ptr()))
        return out

    def block_forward(self, layer, hidden, enc0, enc1, temb):
        hidden, enc0, enc1, temb = (x.to(self.device, self.dtype).contiguous() for x in (hidden, enc0, enc1, temb))
        oh, o0, o1 = torch.empty_like(hidden), torch.empty_like(enc0), torch.empty_like(enc1)
        _lib.check(_lib.lib().s2v_block_forward(self._h, layer, _lib.ptr(hidden), _lib.ptr(enc0), _lib.ptr(enc1),
                                                _lib.ptr(temb), _lib.ptr(oh), _lib.ptr(o0), _lib.ptr(o1),
                                                _lib.stream_ptr()))
        return oh, o0, o1

    def attn_forward(self, layer, hidden, encoder):
        hidden, encoder = (x.to(self.device, self.dtype).contiguous() for x in (hidden, encoder))
        oh, oe = torch.empty_like(hidden), torch.empty_like(encoder)
        _lib.check(_lib.lib().s2v_attn_forward(self._h, layer, _lib.ptr(hidden), _lib.ptr(encoder), _lib.ptr(oh),
                                               _lib.ptr(oe), _lib.stream_ptr()))
        return oh, oe

    AUTO_SLOW_FRACTION = 5e-3   # a slow path costs ~2.5 KV tiles of time, fp16 P saves ~4 %: break-even near 1.7 % of the pairs

    def attn_slow_stats(self, reset=False):
        """(slow paths taken, (wave, KV tile) pairs run) by the four-wave attention kernels since the last reset; synchronises the device"""
        slow, total = ctypes.c_uint64(), ctypes.c_uint64()
        _lib.check(_lib.lib().s2v_attn_slow_stats(self._h, ctypes.byref(slow), ctypes.byref(total), int(reset)))
        return slow.value, total.value

    def set_attn_p_format(self, fmt):
        if fmt not in ("bf16", "f16"):
            raise _lib.S2VError(f"unknown attn_p_format {fmt!r} ('bf16' or 'f16')")
        _lib.check(_lib.lib().s2v_set_attn_p_format(self._h, 1 if fmt == "f16" else 0))
        self.attn_p_format = fmt

    def denoise_step(self, latents, timestep, coef, x0_hist=None, noise=None, use_graph=False):
        """one iteration of the denoise loop, latents [1,F,C,H,W] (model dtype) updated in place"""
        if latents.dtype != self.dtype or not latents.is_contiguous():
            raise _lib.S2VError("latents must be a contiguous model-dtype tensor (it is updated in place)")
        auto = self._attn_auto_pending
        if auto:
            self.attn_slow_stats(reset=True)
        _lib.check(_lib.lib().s2v_denoise_step(self._h, _lib.ptr(latents), float(timestep), ctypes.byref(coef),
                                               _lib.ptr(x0_hist), _lib.ptr(noise), int(use_graph and not auto), _lib.stream_ptr()))
        if auto:  # the first step of an "auto" engine runs eagerly; its census decides the format of every later step
            slow, total = self.attn_slow_stats()
            self.attn_slow_fraction = slow / total if total else 0.0
            if self.attn_slow_fraction > self.AUTO_SLOW_FRACTION:
                self.set_attn_p_format("bf16")
            self._attn_auto_pending = False

    # ---- CFG-parallel: this engine holds ONE sample of the CFG pair (geometry B = 1); include/s2v_hip.h, dist.CfgPair ---------------------------
    def cfg_pair(self):
        """the context-owned pair buffer [2, F, C, H, W] (model dtype): half `slot` is written by denoise_split_begin, the other by the exchange"""
        B, T, F, H, W = self.geometry
        p, n = ctypes.c_void_p(), ctypes.c_int64()
        _lib.check(_lib.lib().s2v_cfg_pair(self._h, ctypes.byref(p), ctypes.byref(n)))
        raw = torch.as_tensor(_ArenaView(p.value, 2 * n.value), device=self.device)
        return raw.view(self.dtype).view(2, F, self.cfg.out_channels, H, W)

    def _no_auto_in_a_pair(self):
        if self.cfg.attn_p_format == "auto":  # "auto" settles per engine from its own census: the two ranks of a pair could settle differently
            raise _lib.S2VError("attn_p_format 'auto' is per-engine; the two ranks of a CFG-parallel pair must run the same arithmetic: use 'bf16' or 'f16'")

    def denoise_split_begin(self, latents, timestep, coef, slot, use_graph=False):
        self._no_auto_in_a_pair()
        if latents.dtype != self.dtype or not latents.is_contiguous():
            raise _lib.S2VError("latents must be a contiguous model-dtype tensor")
        _lib.check(_lib.lib().s2v_denoise_split_begin(self._h, _lib.ptr(latents), float(timestep), ctypes.byref(coef), int(slot), int(use_graph),
                                                      _lib.stream_ptr()))

    def denoise_split_end(self, latents, x0_hist=None, noise=None):
        _lib.check(_lib.lib().s2v_denoise_split_end(self._h, _lib.ptr(latents), _lib.ptr(x0_hist), _lib.ptr(noise), _lib.stream_ptr()))

    def denoise_step_cfg_parallel(self, comm, slot, latents, timestep, coef, x0_hist=None, noise=None, use_graph=False):
        """begin + s2v_rccl_allgather + end in ONE library call over an RcclComm of the two ranks of the pair"""
        self._no_auto_in_a_pair()
        if latents.dtype != self.dtype or not latents.is_contiguous():
            raise _lib.S2VError("latents must be a contiguous model-dtype tensor (it is updated in place)")
        _lib.check(_lib.lib().s2v_denoise_step_cfg_parallel(self._h, comm._h, int(slot), _lib.ptr(latents), float(timestep), ctypes.byref(coef),
                                                            _lib.ptr(x0_hist), _lib.ptr(noise), int(use_graph), _lib.stream_ptr()))

    def last_noise_pred(self):
        B, T, F, H, W = self.geometry
        p = ctypes.c_void_p()
        _lib.check(_lib.lib().s2v_last_noise_pred(self._h, ctypes.byref(p)))
        n = B * F * self.cfg.out_channels * H * W * (4 if self.dtype == torch.float32 else 2)
        raw = torch.as_tensor(_ArenaView(p.value, n), device=self.device)
        return raw.view(self.dtype).view(B, F, self.cfg.out_channels, H, W).clone()
