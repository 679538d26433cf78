"""Drop-in objects for the reference's three Python plug-in seams of the transformer:

  HipCogVideoXTransformer3DModel  -- the Transformer object   (cogvideox_transformer_3d.py:450-462,557-560)
  HipCogVideoXBlock               -- a transformer_blocks[i]  (cogvideox_transformer_3d.py:122-136,186)
  HipCogVideoXAttnProcessor2_0    -- an AttnProcessor         (attention_processor.py:2024-2036,2094-2097)

Same call signatures, argument meaning and error behaviour as the reference objects; all arithmetic runs in
libs2v_hip.so.  Arguments the fork accepts but never honours are accepted and ignored the same way; arguments whose
honouring would need code that does not exist here raise NotImplementedError instead of silently deviating.
"""
from types import SimpleNamespace

import torch

from . import _lib
from .config import TransformerConfig
from .engine import S2VEngine


def _rope_pair(image_rotary_emb, ref_image_rotary_emb, R):
    """[ref | video] cos/sin tables for the packed sequence; a missing ref table means identity rotation."""
    cos, sin = image_rotary_emb
    if ref_image_rotary_emb is not None:
        rc, rs = ref_image_rotary_emb
    else:
        rc = torch.ones((R, cos.shape[1]), dtype=cos.dtype, device=cos.device)
        rs = torch.zeros((R, cos.shape[1]), dtype=cos.dtype, device=cos.device)
    return torch.cat([rc.to(cos.device), cos], dim=0), torch.cat([rs.to(sin.device), sin], dim=0)


class _Cache:
    """recompute the conditioning only when the caller's tensors changed (the rotary tables have the same kind of key on the engine
    itself: S2VEngine.ensure_rope).

    The key holds STRONG references to the tensors (so their storage cannot be freed and handed to a different tensor at the
    same address) and compares object identity + in-place version; it also records the engine's epoch for its kind of state
    ("rope" / "cond"), which every S2VEngine.set_rope / set_pos_embed / set_conditioning / set_geometry bumps -- a write that
    did not go through this cache (the fused pipeline, another seam object sharing the engine) invalidates it."""

    def __init__(self, kind):
        self.kind = kind
        self.key = None

    def changed(self, engine, *tensors):
        epoch = engine.epoch[self.kind]
        k = self.key
        same = (k is not None and k[0] == epoch and len(k[1]) == len(tensors)
                and all((a is None and t is None) or (a is not None and t is not None and a[0] is t and a[1] == t._version)
                        for a, t in zip(k[1], tensors)))
        return not same

    def store(self, engine, *tensors):
        """call AFTER the engine was updated: records the post-update epoch"""
        self.key = (engine.epoch[self.kind], tuple(None if t is None else (t, t._version) for t in tensors))


class HipCogVideoXTransformer3DModel:
    def __init__(self, cfg: TransformerConfig, dtype=torch.bfloat16, device="cuda:0", force_simple=False):
        self.engine = S2VEngine(cfg, dtype, device, force_simple)
        self.config = SimpleNamespace(
            in_channels=cfg.in_channels, out_channels=cfg.out_channels, patch_size=cfg.patch_size,
            attention_head_dim=cfg.attention_head_dim, num_attention_heads=cfg.num_attention_heads,
            num_layers=cfg.num_layers, time_embed_dim=cfg.time_embed_dim, text_embed_dim=cfg.text_embed_dim,
            use_rotary_positional_embeddings=cfg.use_rotary_positional_embeddings)
        self.dtype = dtype
        self.device = self.engine.device
        self.qk_replace = False  # set by CustomCogVideoXPipeline.__init__ (custom_cogvideox_pipe.py:41); never read
        self._cond_cache = _Cache("cond")
        self.transformer_blocks = [HipCogVideoXBlock(self.engine, i) for i in range(cfg.num_layers)]

    def eval(self):
        return self

    def to(self, *args, **kwargs):
        return self

    def load_state_dict(self, sd, lora=None, lora_scale=0.5, strict=True):
        self.engine.load_state_dict(sd, lora, lora_scale)

    def __call__(self, hidden_states, ref_img_states=None, encoder_hidden_states=None, timestep=None,
                 timestep_cond=None, image_rotary_emb=None, ref_image_rotary_emb=None, attention_kwargs=None,
                 return_dict=True, eval=False):
        if timestep_cond is not None:
            raise NotImplementedError("timestep_cond: TimestepEmbedding.cond_proj does not exist in CogVideoX")
        if ref_img_states is None:
            raise TypeError("ref_img_states is required (cogvideox_transformer_3d.py:496 dereferences it)")
        B, F, C, H, W = hidden_states.shape
        if not eval and ref_img_states.shape[0] != B:
            raise RuntimeError("eval=False needs ref_img_states with the batch of hidden_states "
                               "(the reference only duplicates it under eval=True, :503-504)")
        if eval and B != 2 * ref_img_states.shape[0]:
            raise RuntimeError(f"Sizes of tensors must match: eval=True duplicates ref_img_states exactly x2 "
                               f"(:503-504) but hidden_states has batch {B}")
        if ref_img_states.shape[0] != 1:
            raise NotImplementedError("one reference image per call (what src/video_generate.py:35-38 provides)")
        use_rope = self.config.use_rotary_positional_embeddings
        if use_rope and image_rotary_emb is None:
            raise TypeError("'NoneType' object is not subscriptable: a RoPE model needs image_rotary_emb")
        eng = self.engine
        T = encoder_hidden_states.shape[1]
        if eng.geometry != (B, T, F, H, W):
            eng.set_geometry(B, T, F, H, W)
            if not use_rope:
                eng.prepare_tables(H * 8, W * 8)  # the reference rebuilds this table on every forward (:433-446)
        if image_rotary_emb is not None:
            ref = ref_image_rotary_emb
            key = (image_rotary_emb[0], image_rotary_emb[1], None if ref is None else ref[0], None if ref is None else ref[1])
            eng.ensure_rope(key, lambda: _rope_pair(image_rotary_emb, ref, (H // 2) * (W // 2)))
        if self._cond_cache.changed(eng, encoder_hidden_states, ref_img_states):
            eng.set_conditioning(encoder_hidden_states, ref_img_states)
            self._cond_cache.store(eng, encoder_hidden_states, ref_img_states)
        t = timestep if torch.is_tensor(timestep) else torch.tensor([timestep] * B)
        out = eng.forward(hidden_states, t.reshape(-1).float())
        if not return_dict:
            return (out,)
        return SimpleNamespace(sample=out)

    forward = __call__


class HipCogVideoXBlock(torch.nn.Module):
    """transformer.transformer_blocks[i] replacement (Block-module seam).  An nn.Module (without parameters: the weights live in the
    engine's arena) so that `transformer.transformer_blocks[i] = HipCogVideoXBlock(engine, i)` is accepted by the reference's
    nn.ModuleList (cogvideox_transformer_3d.py:315-330)."""

    def __init__(self, engine: S2VEngine, layer: int):
        super().__init__()
        self.engine, self.layer = engine, layer

    def forward(self, hidden_states, encoder_hidden_states, temb, enc_hidden_states1=None, image_rotary_emb=None,
                 embed_ref_img=False, ref_img_seq_start=None, ref_img_seq_end=None, position_delta=None,
                 timestep=None, layer=None, ref_image_rotary_emb=None):
        if enc_hidden_states1 is None:
            raise TypeError("enc_hidden_states1 is required (normalization.py:482 / cogvideox_transformer_3d.py:167)")
        if position_delta is not None and (torch.is_tensor(position_delta) or position_delta != 0):
            raise NotImplementedError("position_delta != 0 (the fork always passes 0, cogvideox_transformer_3d.py:513)")
        eng = self.engine
        B, V, D = hidden_states.shape
        T, R = encoder_hidden_states.shape[1], enc_hidden_states1.shape[1]
        if V % R != 0:
            raise RuntimeError("video tokens must be a whole number of frames of the reference image's token count")
        geo = (B, T, V // R, 2, 2 * R)
        if eng.geometry != geo:
            eng.set_geometry(*geo)
        if image_rotary_emb is not None:
            ref = ref_image_rotary_emb if embed_ref_img else None
            key = (image_rotary_emb[0], image_rotary_emb[1], None if ref is None else ref[0], None if ref is None else ref[1])
            eng.ensure_rope(key, lambda: _rope_pair(image_rotary_emb, ref, R))
        else:
            eng.ensure_no_rope()
        return eng.block_forward(self.layer, hidden_states, encoder_hidden_states, enc_hidden_states1, temb)


class HipCogVideoXAttnProcessor2_0:
    """AttnProcessor seam: `attn.set_processor(HipCogVideoXAttnProcessor2_0())`.  Weights are read from the
    Attention module that owns them (to_q/to_k/to_v/to_out[0]/norm_q/norm_k, attention_processor.py:2049-2090) and
    re-packed once per module into a one-layer context."""

    def __init__(self, force_simple=False):
        self._engines = {}
        self._force_simple = force_simple

    def _engine_for(self, attn, dtype, device):
        key = id(attn)
        if key not in self._engines:
            heads = attn.heads
            cfg = TransformerConfig(num_layers=1, num_attention_heads=heads, time_embed_dim=8, text_embed_dim=64,
                                    use_rotary_positional_embeddings=True)
            eng = S2VEngine(cfg, dtype, device, self._force_simple)
            p = "transformer_blocks.0.attn1."
            for name, mod in (("to_q", attn.to_q), ("to_k", attn.to_k), ("to_v", attn.to_v), ("to_out.0", attn.to_out[0]),
                              ("norm_q", attn.norm_q), ("norm_k", attn.norm_k)):
                eng.load_weight(p + name + ".weight", mod.weight.detach())
                eng.load_weight(p + name + ".bias", mod.bias.detach())
            torch.cuda.synchronize()
            eng.mark_weights_loaded()  # only attn1 is used through s2v_attn_forward
            self._engines[key] = (eng, attn)  # the module is kept alive: id(attn) stays unique
        return self._engines[key]

    def __call__(self, attn, hidden_states, encoder_hidden_states, attention_mask=None, image_rotary_emb=None,
                 ref_img_seq_start=0, ref_img_seq_end=0, position_delta=None, embed_ref_img=False,
                 ref_image_rotary_emb=None):
        if attention_mask is not None:
            raise NotImplementedError("attention_mask: the CogVideoX path always passes None (:2083-2085)")
        if position_delta is not None and (torch.is_tensor(position_delta) or position_delta != 0):
            raise NotImplementedError("position_delta != 0 (the fork always passes 0)")
        if getattr(attn, "is_cross_attention", False):
            raise NotImplementedError("cross-attention variant is not part of this path")
        B, V, D = hidden_states.shape
        TR = encoder_hidden_states.shape[1]
        if not embed_ref_img or ref_img_seq_end != TR or not (0 <= ref_img_seq_start < ref_img_seq_end):
            raise NotImplementedError("the fork always calls with embed_ref_img=True and the reference-image tokens at "
                                      "the tail of encoder_hidden_states (cogvideox_transformer_3d.py:510-512)")
        T, R = ref_img_seq_start, ref_img_seq_end - ref_img_seq_start
        eng, _ = self._engine_for(attn, hidden_states.dtype, hidden_states.device)
        geo = (B, T, V // R, 2, 2 * R)
        if V % R != 0:
            raise RuntimeError("video tokens must be a whole number of frames of the reference image's token count")
        if eng.geometry != geo:
            eng.set_geometry(*geo)
        if image_rotary_emb is not None:
            ref = ref_image_rotary_emb
            key = (image_rotary_emb[0], image_rotary_emb[1], None if ref is None else ref[0], None if ref is None else ref[1])
            eng.ensure_rope(key, lambda: _rope_pair(image_rotary_emb, ref, R))
        else:
            eng.ensure_no_rope()
        return eng.attn_forward(0, hidden_states, encoder_hidden_states)
