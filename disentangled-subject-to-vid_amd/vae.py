"""Drop-in object for the VAE seam of the pipeline (SURVEY.md section 8b "VAE object"):
`.config.{block_out_channels, temporal_compression_ratio, scaling_factor}`, `.decode(z).sample`,
`enable_tiling / disable_tiling / enable_slicing`, `.eval()` as used by
pipelines/cogvideo/pipeline_cogvideox.py:185-193,346-351 and src/inference.py:201-207.  Decode runs in
libs2v_hip.so (csrc/vae.hip, vae_api.hip).  `encode(x).latent_dist.sample()` of ONE reference frame
(src/video_generate.py:26-38) runs there too once `encoder.*` weights were loaded; video encode raises."""
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
    "s2v_vae_weight_arena": [_P, ctypes.POINTER(_P), ctypes.POINTER(_I64)],
    "s2v_vae_mark_weights_loaded": [_P],
    "s2v_vae_out_shape": [_P, _I32, _I32, _I32, _I32, ctypes.POINTER(_I32), ctypes.POINTER(_I32), ctypes.POINTER(_I32)],
    "s2v_vae_workspace_info": [_P, ctypes.POINTER(_I32), ctypes.POINTER(_I64)],
    "s2v_vae_decode": [_P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P],
    "s2v_vae_postprocess": [_P, _I32, _I32, _I32, _I32, _P, _I32, _P],
    "s2v_vae_postprocess_u8": [_P, _I32, _I32, _I32, _I32, _P, _I32, _P],
    "s2v_vae_enc_create": [ctypes.POINTER(VaeConfigC), ctypes.POINTER(_P)],
    "s2v_vae_encode_shape": [_P, _I32, _I32, _I32, ctypes.POINTER(_I32), ctypes.POINTER(_I32)],
    "s2v_vae_encode": [_P, _P, _I32, _I32, _I32, _P, _P],
    "s2v_vae_gaussian_sample": [_P, _P, _I32, _I64, _P, _I32, _P],
})


class HipDiagonalGaussianDistribution:
    """DiagonalGaussianDistribution (autoencoders/vae.py:767-790) over device-resident moments [1,2C,1,h,w]."""

    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)

    def sample(self, generator=None):
        # randn_tensor (utils/torch_utils.py): a CPU generator draws on the CPU and the draw is moved to the device
        p = self.parameters
        shape = self.mean.shape
        gdev = generator.device if generator is not None else p.device
        noise = torch.randn(shape, generator=generator, device=gdev, dtype=p.dtype).to(p.device).contiguous()
        out = torch.empty(shape, dtype=p.dtype, device=p.device)
        n_sp = shape[2] * shape[3] * shape[4]
        _lib.check(_lib.lib().s2v_vae_gaussian_sample(_lib.ptr(p), _lib.ptr(noise), shape[1], n_sp, _lib.ptr(out),
                                                      _lib.DTYPE_OF[p.dtype], _lib.stream_ptr()))
        return out

    def mode(self):
        return self.mean


class HipAutoencoderKLCogVideoX:
    def __init__(self, cfg: VAEConfig = None, dtype=torch.bfloat16, device="cuda:0", force_simple=False, with_encoder=True):
        """with_encoder: also build the encoder half (AutoencoderKLCogVideoX always has one; src/video_generate.py:26-38 encodes the
        reference image with it).  Every replica must be built the same way: the arenas of BOTH halves are what
        dist.broadcast_components replicates, so a receiving rank can run `encode` without ever seeing a checkpoint."""
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
        self._cfg_c = c
        self._enc = ctypes.c_void_p()  # encoder handle (also created lazily when `encoder.*` weights arrive)
        self._enc_loaded = False       # weights present: load_state_dict saw `encoder.*` keys, or a replica was marked loaded
        self._dec_loaded = False
        if with_encoder:
            _lib.check(_lib.lib().s2v_vae_enc_create(ctypes.byref(c), ctypes.byref(self._enc)))

    def close(self):
        lib = _lib.lib()
        lib.s2v_vae_destroy.argtypes = [_P]
        lib.s2v_vae_destroy.restype = None
        for name in ("_h", "_enc"):
            h = getattr(self, name, None)
            if h:
                lib.s2v_vae_destroy(h)
                setattr(self, name, ctypes.c_void_p())

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
        """AutoencoderKLCogVideoX.enable_slicing (:1070-1075) splits a BATCH of videos into single-video decodes; this object's
        decode / encode take B = 1 (the pipeline's only use, pipeline_cogvideox.py:346-351), so the flag is recorded and has no
        effect"""
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    def load_state_dict(self, sd, strict=True):
        """sd: reference state dict.  `decoder.*` tensors feed the decode path; `encoder.*` tensors, when present, the
        reference-image encode (either half may be loaded on its own)."""
        keep = []
        halves = {"decoder.": self._h}
        if any(k.startswith("encoder.") for k in sd):
            if not self._enc:
                _lib.check(_lib.lib().s2v_vae_enc_create(ctypes.byref(self._cfg_c), ctypes.byref(self._enc)))
            halves["encoder."] = self._enc
        loaded = set()
        for k, t in sd.items():
            h = next((hh for pre, hh in halves.items() if k.startswith(pre)), None)
            if h is None:
                continue
            t = t.to(self.device)
            if t.dtype not in _lib.DTYPE_OF:
                t = t.float()
            t = t.contiguous()
            shape = (_I64 * t.ndim)(*t.shape)
            _lib.check(_lib.lib().s2v_vae_load_weight(h, k.encode(), _lib.ptr(t), shape, t.ndim,
                                                      _lib.DTYPE_OF[t.dtype], _lib.stream_ptr()))
            keep.append(t)
            loaded.add(k.split(".", 1)[0] + ".")
        torch.cuda.synchronize(self.device)
        for pre in loaded:
            _lib.check(_lib.lib().s2v_vae_finalize(halves[pre]))
        if "encoder." in loaded:
            self._enc_loaded = True
        if "decoder." in loaded:
            self._dec_loaded = True

    # ---- replicas: the weights of each half are one device range (dist.broadcast_components) ------------------------
    def weight_arenas(self, with_encoder=None):
        """list of uint8 CUDA tensors aliasing the packed weights: [decoder, encoder], or [decoder] for an object built or asked
        without the encoder half (with_encoder: None = whatever this object has, True = create the half if missing)"""
        from .engine import _ArenaView

        if with_encoder and not self._enc:
            _lib.check(_lib.lib().s2v_vae_enc_create(ctypes.byref(self._cfg_c), ctypes.byref(self._enc)))
        if with_encoder is None:
            with_encoder = bool(self._enc)
        out = []
        for h in [self._h] + ([self._enc] if with_encoder else []):
            p, n = ctypes.c_void_p(), ctypes.c_int64()
            _lib.check(_lib.lib().s2v_vae_weight_arena(h, ctypes.byref(p), ctypes.byref(n)))
            out.append(torch.as_tensor(_ArenaView(p.value, n.value), device=self.device))
        return out

    def arenas_loaded(self):
        """one flag per arena of weight_arenas(): does it hold weights?  (the sender's side of dist.broadcast_components)"""
        return [self._dec_loaded] + ([self._enc_loaded] if self._enc else [])

    def mark_weights_loaded(self, with_encoder=None, loaded=None):
        """the receiving side of a replica hand-off.  `loaded` = the SENDER's arenas_loaded(): only the halves the sender had
        filled are marked (a sender that never saw `encoder.*` weights leaves this replica's encode() raising, as on the sender);
        without it, the same halves weight_arenas() listed are taken to hold a sender's weights"""
        if loaded is not None:
            handles = [self._h] + ([self._enc] if self._enc else [])
            if len(loaded) != len(handles):
                raise _lib.S2VError(f"mark_weights_loaded: {len(loaded)} flags for {len(handles)} arenas")
            for h, flag in zip(handles, loaded):
                if flag:
                    _lib.check(_lib.lib().s2v_vae_mark_weights_loaded(h))
            self._dec_loaded = self._dec_loaded or bool(loaded[0])
            if len(loaded) > 1 and loaded[1]:
                self._enc_loaded = True
            return
        if with_encoder is None:
            with_encoder = bool(self._enc)
        for h in [self._h] + ([self._enc] if with_encoder else []):
            _lib.check(_lib.lib().s2v_vae_mark_weights_loaded(h))
        self._dec_loaded = True
        if with_encoder:
            self._enc_loaded = True

    def encode(self, x, return_dict=True):
        """AutoencoderKLCogVideoX.encode (:1205-1229) for the reference image: x [1,3,1,H,W] in [-1,1] ->
        `.latent_dist` whose `.sample(generator)` is the [1,C,1,h,w] draw (src/video_generate.py:35-37)."""
        if not self._enc or not self._enc_loaded:
            raise _lib.S2VError("encode: no `encoder.*` weights were loaded into this VAE")
        if x.ndim != 5 or x.shape[0] != 1 or x.shape[2] != 1:
            raise NotImplementedError("encode takes ONE frame [1,3,1,H,W]: video encode is outside the path")
        x = x.to(self.device, self.dtype).contiguous()
        H, W = x.shape[3], x.shape[4]
        ho, wo = _I32(), _I32()
        _lib.check(_lib.lib().s2v_vae_encode_shape(self._enc, H, W, int(self.use_tiling), ctypes.byref(ho), ctypes.byref(wo)))
        mom = torch.empty((1, 2 * self.cfg.latent_channels, 1, ho.value, wo.value), dtype=self.dtype, device=self.device)
        _lib.check(_lib.lib().s2v_vae_encode(self._enc, _lib.ptr(x), H, W, int(self.use_tiling), _lib.ptr(mom), _lib.stream_ptr()))
        post = HipDiagonalGaussianDistribution(mom)
        if not return_dict:
            return (post,)
        return SimpleNamespace(latent_dist=post)

    def workspace_info(self):
        """(workspace sets = tiles in flight, bytes per set) of the decoder's current capacity: the tiled decode takes as many
        sets as 70 % of the free HBM holds, at most six"""
        n, b = _I32(), _I64()
        _lib.check(_lib.lib().s2v_vae_workspace_info(self._h, ctypes.byref(n), ctypes.byref(b)))
        return n.value, b.value

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

    def frames_uint8(self, video):
        """decoded video [1,3,F,H,W] -> uint8 [F,H,W,3]: postprocess_video(..., "np")[0] followed by export_to_video's
        `(frame * 255).astype(np.uint8)` (utils/export_utils.py:175), converted on the device"""
        if video.shape[0] != 1:
            raise NotImplementedError("one video per call")
        _, C, F, H, W = video.shape
        v = video[0].contiguous()
        o = torch.empty((F, H, W, C), dtype=torch.uint8, device=v.device)
        _lib.check(_lib.lib().s2v_vae_postprocess_u8(_lib.ptr(v), C, F, H, W, _lib.ptr(o), _lib.DTYPE_OF[v.dtype], _lib.stream_ptr()))
        return o

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
