"""Drop-in object for the VAE seam of the pipeline (SURVEY.md section 8b "VAE object"):
`.config.{block_out_channels, temporal_compression_ratio, scaling_factor}`, `.decode(z).sample`,
`enable_tiling / disable_tiling / enable_slicing`, `.eval()` as used by
pipelines/cogvideo/pipeline_cogvideox.py:185-193,346-351 and src/inference.py:201-207.  Decode runs in
libs2v_hip.so (csrc/vae.hip, vae_api.hip).  `encode` is a caller-side step outside this path and
raises."""
import ctypes
from types import SimpleNamespace

import torch

from . import _lib
from .config import VAEConfig

_P, _I32, _I64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64


class VaeConfigC(ctypes.Structure):
    _fields_ = [("latent_channels", _I32), ("out_channels", _I32), ("num_blocks", _I32),
                ("block_out_channels", _I32 * 8), ("layers_per_block", _I32), ("norm_num_groups", _I32),
                ("temporal_compression_ratio", _I32), ("sample_height", _I32), ("sample_width", _I32),
                ("dtype", _I32), ("force_simple", _I32), ("scaling_factor", ctypes.c_float),
                ("norm_eps", ctypes.c_float), ("reserved", _I32 * 4)]


_lib.register_sigs({
    "s2v_vae_create": [ctypes.POINTER(VaeConfigC), ctypes.POINTER(_P)],
    "s2v_vae_load_weight": [_P, ctypes.c_char_p, _P, ctypes.POINTER(_I64), _I32, _I32, _P],
    "s2v_vae_finalize": [_P],
    "s2v_vae_out_shape": [_P, _I32, _I32, _I32, _I32, ctypes.POINTER(_I32), ctypes.POINTER(_I32), ctypes.POINTER(_I32)],
    "s2v_vae_decode": [_P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P],
    "s2v_vae_postprocess": [_P, _I32, _I32, _I32, _I32, _P, _I32, _P],
})


class HipAutoencoderKLCogVideoX:
    def __init__(self, cfg: VAEConfig = None, dtype=torch.bfloat16, device="cuda:0", force_simple=False):
        cfg = cfg or VAEConfig()
        if dtype not in _lib.DTYPE_OF:
            raise _lib.S2VError(f"unsupported VAE dtype {dtype}")
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        self.config = SimpleNamespace(block_out_channels=tuple(cfg.block_out_channels),
                                      temporal_compression_ratio=cfg.temporal_compression_ratio,
                                      scaling_factor=cfg.scaling_factor, latent_channels=cfg.latent_channels,
                                      sample_height=cfg.sample_height, sample_width=cfg.sample_width)
        self.use_tiling = False
        self.use_slicing = False
        torch.cuda.set_device(self.device)
        c = VaeConfigC()
        c.latent_channels, c.out_channels = cfg.latent_channels, cfg.out_channels
        c.num_blocks = len(cfg.block_out_channels)
        for i, ch in enumerate(cfg.block_out_channels):
            c.block_out_channels[i] = ch
        c.layers_per_block, c.norm_num_groups = cfg.layers_per_block, cfg.norm_num_groups
        c.temporal_compression_ratio = cfg.temporal_compression_ratio
        c.sample_height, c.sample_width = cfg.sample_height, cfg.sample_width
        c.dtype, c.force_simple = _lib.DTYPE_OF[dtype], int(force_simple)
        c.scaling_factor, c.norm_eps = cfg.scaling_factor, 1e-6
        self._h = ctypes.c_void_p()
        _lib.check(_lib.lib().s2v_vae_create(ctypes.byref(c), ctypes.byref(self._h)))

    def close(self):
        if self._h:
            lib = _lib.lib()
            lib.s2v_vae_destroy.argtypes = [_P]
            lib.s2v_vae_destroy.restype = None
            lib.s2v_vae_destroy(self._h)
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

    def enable_tiling(self, *a, **k):
        if a or k:
            raise NotImplementedError("custom tile sizes / overlap factors (enable_tiling is called without "
                                      "arguments in src/inference.py:206-207)")
        self.use_tiling = True

    def disable_tiling(self):
        self.use_tiling = False

    def enable_slicing(self):
        self.use_slicing = True  # batch slicing is a no-op for the single-video path (decode takes B = 1)

    def disable_slicing(self):
        self.use_slicing = False

    def load_state_dict(self, sd, strict=True):
        """sd: reference state dict; only `decoder.*` tensors are consumed (the encoder is outside this path)."""
        keep = []
        for k, t in sd.items():
            if not k.startswith("decoder."):
                continue
            t = t.to(self.device)
            if t.dtype not in _lib.DTYPE_OF:
                t = t.float()
            t = t.contiguous()
            shape = (_I64 * t.ndim)(*t.shape)
            _lib.check(_lib.lib().s2v_vae_load_weight(self._h, k.encode(), _lib.ptr(t), shape, t.ndim,
                                                      _lib.DTYPE_OF[t.dtype], _lib.stream_ptr()))
            keep.append(t)
        torch.cuda.synchronize(self.device)
        _lib.check(_lib.lib().s2v_vae_finalize(self._h))

    def encode(self, x):
        raise NotImplementedError("VAE encode of the reference image is a caller-side step (src/video_generate.py:26-38)")

    def _out_shape(self, F, h, w):
        fo, ho, wo = _I32(), _I32(), _I32()
        _lib.check(_lib.lib().s2v_vae_out_shape(self._h, F, h, w, int(self.use_tiling), ctypes.byref(fo), ctypes.byref(ho),
                                                ctypes.byref(wo)))
        return fo.value, ho.value, wo.value

    def _decode(self, lat, scaled):
        if lat.shape[0] != 1:
            raise NotImplementedError("one video per decode call")
        lat = lat.to(self.device, self.dtype).contiguous()
        _, F, C, h, w = lat.shape
        fo, ho, wo = self._out_shape(F, h, w)
        out = torch.empty((1, self.cfg.out_channels, fo, ho, wo), dtype=self.dtype, device=self.device)
        _lib.check(_lib.lib().s2v_vae_decode(self._h, _lib.ptr(lat), F, h, w, int(self.use_tiling), int(scaled),
                                             _lib.ptr(out), _lib.stream_ptr()))
        return out

    def decode_latents(self, latents):
        """CogVideoXPipeline.decode_latents (pipeline_cogvideox.py:346-351): latents [1,F,C,h,w] in the pipeline's
        layout -> frames [1,3,F',H,W]; the 1/scaling_factor product is folded into the first kernel."""
        return self._decode(latents, True)

    def decode(self, z, return_dict=True):
        """AutoencoderKLCogVideoX.decode (:1259-1282): z [1,C,F,h,w] (already divided by scaling_factor)."""
        dec = self._decode(z.permute(0, 2, 1, 3, 4), False)
        if not return_dict:
            return (dec,)
        return SimpleNamespace(sample=dec)

    def postprocess_video(self, video, output_type="np"):
        """VideoProcessor.postprocess_video: [B,3,F,H,W] -> np float32 [B,F,H,W,3] (or 'pt' [B,F,3,H,W])"""
        if output_type not in ("np", "pt"):
            raise ValueError(f"{output_type} does not exist. Please choose one of ['np', 'pt']")
        B, C, F, H, W = video.shape
        outs = []
        for b in range(B):
            v = video[b].contiguous()
            o = torch.empty((F, H, W, C), dtype=torch.float32, device=v.device)
            _lib.check(_lib.lib().s2v_vae_postprocess(_lib.ptr(v), C, F, H, W, _lib.ptr(o), _lib.DTYPE_OF[v.dtype],
                                                      _lib.stream_ptr()))
            outs.append(o)
        res = torch.stack(outs)
        if output_type == "np":
            return res.cpu().numpy()
        return res.permute(0, 1, 4, 2, 3)
