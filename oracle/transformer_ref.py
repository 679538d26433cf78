"""ORACLE -- test infrastructure only, never imported by the product package.

CPU restatement (plain torch, straight-line, no nn.Module) of the 3-stream CogVideoX transformer step that
carpedkm/disentangled-subject-to-vid runs.  Each function names the reference lines it follows (paths relative to
/root/reference).  Pinned against golden vectors captured from the imported reference by oracle/make_golden.py
(tests/golden/*.npz, tests/test_oracle_golden.py); parity status: PINNED for fp32 and bf16.

Weights are a flat dict keyed by the reference's state-dict names.  All tensor math runs in the dtype of the
inputs with the same torch ops (F.linear, F.layer_norm, SDPA) the reference uses, so a bf16 run rounds where the
reference's bf16 CPU run rounds.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------------------------
# tables
def timestep_sinusoid(t, dim):
    """diffusers/src/diffusers/models/embeddings.py:27-78 with flip_sin_to_cos=True, downscale_freq_shift=0."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def resize_crop_region(src_hw, tgt_w, tgt_h):
    """pipelines/cogvideo/pipeline_cogvideox.py:62-77 (get_resize_crop_region_for_grid)."""
    h, w = src_hw
    r = h / w
    if r > tgt_h / tgt_w:
        rh, rw = tgt_h, int(round(tgt_h / h * w))
    else:
        rw, rh = tgt_w, int(round(tgt_w / w * h))
    top, left = int(round((tgt_h - rh) / 2.0)), int(round((tgt_w - rw) / 2.0))
    return (top, left), (top + rh, left + rw)


def _rope_1d(dim, pos):
    """embeddings.py:673-727 (use_real, repeat_interleave_real)."""
    pos = torch.from_numpy(pos)
    freqs = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
    fr = torch.outer(pos, freqs)
    return fr.cos().repeat_interleave(2, dim=1).float(), fr.sin().repeat_interleave(2, dim=1).float()


def rope_3d(head_dim, crops, grid_hw, temporal_size):
    """embeddings.py:505-570 (get_3d_rotary_pos_embed): cos, sin of shape [t*h*w, head_dim]."""
    (s0, s1), (e0, e1) = crops
    gh, gw = grid_hw
    grid_h = np.linspace(s0, e0, gh, endpoint=False, dtype=np.float32)
    grid_w = np.linspace(s1, e1, gw, endpoint=False, dtype=np.float32)
    grid_t = np.linspace(0, temporal_size, temporal_size, endpoint=False, dtype=np.float32)
    dt, dh, dw = head_dim // 4, head_dim // 8 * 3, head_dim // 8 * 3
    tc, ts = _rope_1d(dt, grid_t)
    hc, hs = _rope_1d(dh, grid_h)
    wc, ws = _rope_1d(dw, grid_w)

    def comb(a, b, c):
        a = a[:, None, None, :].expand(-1, gh, gw, -1)
        b = b[None, :, None, :].expand(temporal_size, -1, gw, -1)
        c = c[None, None, :, :].expand(temporal_size, gh, -1, -1)
        return torch.cat([a, b, c], dim=-1).reshape(temporal_size * gh * gw, -1)

    return comb(tc, hc, wc), comb(ts, hs, ws)


def pipeline_rope(height, width, latent_frames, head_dim=64, patch=2, vae_sf=8, base_w=720, base_h=480):
    """pipeline_cogvideox.py:436-460 + the ref/video slicing of src/custom_cogvideox_pipe.py:223-235, generalised
    from the hard-coded 1350 tokens per frame to (H/16)(W/16).  Returns (ref_cos, ref_sin), (vid_cos, vid_sin)."""
    gh, gw = height // (vae_sf * patch), width // (vae_sf * patch)
    crops = resize_crop_region((gh, gw), base_w // (vae_sf * patch), base_h // (vae_sf * patch))
    cos, sin = rope_3d(head_dim, crops, (gh, gw), latent_frames + 1)
    n = gh * gw
    return (cos[:n], sin[:n]), (cos[n : n * (latent_frames + 1)], sin[n : n * (latent_frames + 1)])


def _sincos_1d(dim, pos):
    """embeddings.py:162-180 (float64)."""
    omega = np.arange(dim // 2, dtype=np.float64) / (dim / 2.0)
    omega = 1.0 / 10000**omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_3d(embed_dim, wp, hp, frames, spatial_scale=1.875, temporal_scale=1.0):
    """embeddings.py:81-125 as called from CogVideoXPatchEmbed._get_positional_embeddings (:380-401) with
    spatial_size=(post_patch_width, post_patch_height).  Returns [frames*hp*wp, embed_dim] float64->float32 exactly
    as torch.from_numpy(...) copied into the float32 joint table does."""
    ds, dt = 3 * embed_dim // 4, embed_dim // 4
    grid_h = np.arange(hp, dtype=np.float32) / spatial_scale
    grid_w = np.arange(wp, dtype=np.float32) / spatial_scale
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape([2, 1, hp, wp])
    emb0 = _sincos_1d(ds // 2, grid[0])  # "emb_h" of the reference encodes the W coordinate (quirk :155-156)
    emb1 = _sincos_1d(ds // 2, grid[1])
    spatial = np.concatenate([emb0, emb1], axis=1)  # [hp*wp, ds]
    temporal = _sincos_1d(dt, np.arange(frames, dtype=np.float32) / temporal_scale)  # [frames, dt]
    spatial = np.repeat(spatial[None], frames, axis=0)
    temporal = np.repeat(temporal[:, None], hp * wp, axis=1)
    pe = np.concatenate([temporal, spatial], axis=-1).reshape(frames * hp * wp, embed_dim)
    return torch.from_numpy(pe).float()


# ----------------------------------------------------------------------------------------------------------------
# modules
def apply_rope(x, cos, sin):
    """embeddings.py:759-778 (use_real_unbind_dim=-1); x [B,H,S,D]."""
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return (x.float() * cos[None, None] + rot.float() * sin[None, None]).to(x.dtype)


def attn_forward(sd, p, heads, hidden, enc, rope, ref_rope, ref_start, ref_end):
    """CogVideoXAttnProcessor2_0.__call__, models/attention_processor.py:2024-2097.
    hidden [B,V,D], enc [B,T+R,D] (already modulated); returns (hidden_out, enc_out)."""
    tl = enc.size(1)
    x = torch.cat([enc, hidden], dim=1)
    B = x.shape[0]
    q = F.linear(x, sd[p + "to_q.weight"], sd[p + "to_q.bias"])
    k = F.linear(x, sd[p + "to_k.weight"], sd[p + "to_k.bias"])
    v = F.linear(x, sd[p + "to_v.weight"], sd[p + "to_v.bias"])
    hd = q.shape[-1] // heads
    q = q.view(B, -1, heads, hd).transpose(1, 2)
    k = k.view(B, -1, heads, hd).transpose(1, 2)
    v = v.view(B, -1, heads, hd).transpose(1, 2)
    q = F.layer_norm(q, (hd,), sd[p + "norm_q.weight"], sd[p + "norm_q.bias"], 1e-6)
    k = F.layer_norm(k, (hd,), sd[p + "norm_k.weight"], sd[p + "norm_k.bias"], 1e-6)
    if rope is not None:
        q[:, :, tl:] = apply_rope(q[:, :, tl:], *rope)
        k[:, :, tl:] = apply_rope(k[:, :, tl:], *rope)
        if ref_rope is not None:
            q[:, :, ref_start:ref_end] = apply_rope(q[:, :, ref_start:ref_end], *ref_rope)
            k[:, :, ref_start:ref_end] = apply_rope(k[:, :, ref_start:ref_end], *ref_rope)
    o = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False)
    o = o.transpose(1, 2).reshape(B, -1, heads * hd)
    o = F.linear(o, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])
    return o[:, tl:], o[:, :tl]


def layernorm_zero(sd, p, h, e0, e1, temb, eps, cond_sd=None):
    """CogVideoXLayerNormZero.forward, models/normalization.py:467-484.  As shipped the second evaluation of self.linear equals
    the first (the `enable_lora` context sets an attribute nothing reads), so the ref stream takes the VIDEO chunks.  cond_sd (the
    "intended" reading of :468-478): the first evaluation uses the BASE linear (sd), the second -- cond_shift, cond_scale,
    cond_gate for the reference-image stream -- the LoRA-merged linear given in cond_sd."""
    m = F.linear(F.silu(temb), sd[p + "linear.weight"], sd[p + "linear.bias"])
    shift, scale, gate, eshift, escale, egate = m.chunk(6, dim=1)
    if cond_sd is not None:
        mc = F.linear(F.silu(temb), cond_sd[p + "linear.weight"], cond_sd[p + "linear.bias"])
        cshift, cscale, cgate = mc.chunk(6, dim=1)[:3]
        w, b = sd[p + "norm.weight"], sd[p + "norm.bias"]
        D = h.shape[-1]
        nh = F.layer_norm(h, (D,), w, b, eps) * (1 + scale)[:, None, :] + shift[:, None, :]
        ne0 = F.layer_norm(e0, (D,), w, b, eps) * (1 + escale)[:, None, :] + eshift[:, None, :]
        ne1 = F.layer_norm(e1, (D,), w, b, eps) * (1 + cscale)[:, None, :] + cshift[:, None, :]
        return nh, ne0, ne1, gate[:, None, :], egate[:, None, :], cgate[:, None, :]
    w, b = sd[p + "norm.weight"], sd[p + "norm.bias"]
    D = h.shape[-1]
    nh = F.layer_norm(h, (D,), w, b, eps) * (1 + scale)[:, None, :] + shift[:, None, :]
    ne0 = F.layer_norm(e0, (D,), w, b, eps) * (1 + escale)[:, None, :] + eshift[:, None, :]
    ne1 = F.layer_norm(e1, (D,), w, b, eps) * (1 + scale)[:, None, :] + shift[:, None, :]
    return nh, ne0, ne1, gate[:, None, :], egate[:, None, :], gate[:, None, :]


def block_forward(sd, p, heads, h, e0, e1, temb, rope, ref_rope, eps=1e-5, cond_sd=None):
    """CogVideoXBlock.forward, models/transformers/cogvideox_transformer_3d.py:122-186."""
    T, R = e0.size(1), e1.size(1)
    nh, ne0, ne1, g, ge, gc = layernorm_zero(sd, p + "norm1.", h, e0, e1, temb, eps, cond_sd)
    ah, ae = attn_forward(sd, p + "attn1.", heads, nh, torch.cat([ne0, ne1], dim=1), rope, ref_rope, T, T + R)
    h = h + g * ah
    e0 = e0 + ge * ae[:, :T]
    e1 = e1 + gc * ae[:, T:]
    nh, ne0, ne1, g, ge, gc = layernorm_zero(sd, p + "norm2.", h, e0, e1, temb, eps, cond_sd)
    x = torch.cat([ne0, ne1, nh], dim=1)
    x = F.linear(x, sd[p + "ff.net.0.proj.weight"], sd[p + "ff.net.0.proj.bias"])
    x = F.gelu(x, approximate="tanh")  # models/activations.py:65-90
    x = F.linear(x, sd[p + "ff.net.2.weight"], sd[p + "ff.net.2.bias"])  # models/attention.py:1237-1243
    h = h + g * x[:, T + R :]
    e0 = e0 + ge * x[:, :T]
    e1 = e1 + gc * x[:, T : T + R]
    return h, e0, e1


def patch_tokens(sd, lat):
    """2x2 stride-2 conv + flatten, embeddings.py:414-419: [B,F,C,H,W] -> [B, F*(H/2)*(W/2), D]."""
    B, Fr, C, H, W = lat.shape
    y = F.conv2d(lat.reshape(-1, C, H, W), sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=2)
    y = y.view(B, Fr, *y.shape[1:]).flatten(3).transpose(2, 3).flatten(1, 2)
    return y


def transformer_forward(sd, cfg, hidden_states, encoder_hidden_states, ref_img_states, timestep, rope=None,
                        ref_rope=None, cond_sd=None):
    """CogVideoXTransformer3DModel.forward with eval=True, cogvideox_transformer_3d.py:450-560.
    cfg: dict(num_heads, num_layers, use_rope, norm_eps, spatial_scale, temporal_scale)."""
    heads = cfg["num_heads"]
    D = heads * 64
    B, Fr, C, H, W = hidden_states.shape
    dt = hidden_states.dtype
    te = timestep_sinusoid(timestep, D).to(dt)
    emb = F.linear(te, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    emb = F.linear(F.silu(emb), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
    e0 = F.linear(encoder_hidden_states, sd["patch_embed.text_proj.weight"], sd["patch_embed.text_proj.bias"])
    e1 = patch_tokens(sd, ref_img_states)
    e1 = torch.cat([e1] * B, dim=0) if e1.shape[0] != B else e1  # :503-504 duplicates exactly x2
    h = patch_tokens(sd, hidden_states)
    if not cfg["use_rope"]:
        pe = sincos_3d(D, W // 2, H // 2, Fr, cfg.get("spatial_scale", 1.875), cfg.get("temporal_scale", 1.0))
        h = h + pe[None].to(dt)  # embeddings.py:440-446 (text rows of the joint table are zero and dropped)
    for i in range(cfg["num_layers"]):
        h, e0, e1 = block_forward(sd, f"transformer_blocks.{i}.", heads, h, e0, e1, emb, rope, ref_rope,
                                  cfg.get("norm_eps", 1e-5), cond_sd)
    h = F.layer_norm(h, (D,), sd["norm_final.weight"], sd["norm_final.bias"], cfg.get("norm_eps", 1e-5))
    m = F.linear(F.silu(emb), sd["norm_out.linear.weight"], sd["norm_out.linear.bias"])
    shift, scale = m.chunk(2, dim=1)  # normalization.py:72-82: shift FIRST
    h = F.layer_norm(h, (D,), sd["norm_out.norm.weight"], sd["norm_out.norm.bias"], cfg.get("norm_eps", 1e-5))
    h = h * (1 + scale[:, None, :]) + shift[:, None, :]
    y = F.linear(h, sd["proj_out.weight"], sd["proj_out.bias"])
    y = y.reshape(B, Fr, H // 2, W // 2, -1, 2, 2).permute(0, 1, 4, 2, 5, 3, 6).flatten(5, 6).flatten(3, 4)
    return y


def merge_lora(sd, lora, scale=0.5):
    """W' = W + (alpha/r) B A, the merge equivalent to the PEFT adapter of src/inference.py:218-229 (alpha/r=64/128).
    lora: {name: (A [r, in...], B [out, r])}; conv A is [r, C, 2, 2]."""
    out = dict(sd)
    for name, (A, Bm) in lora.items():
        w = sd[name].float()
        delta = (Bm.float() @ A.float().reshape(A.shape[0], -1)).reshape(w.shape)
        out[name] = (w + scale * delta).to(sd[name].dtype)
    return out


def merge_lora_scoped(sd, lora, scale=0.5):
    """the "intended" reading of normalization.py:468-478: every LoRA target is merged EXCEPT norm{1,2}.linear, whose merged form
    is returned separately (cond_sd) and reaches only the reference-image modulation.  Returns (sd_main, cond_sd)."""
    adaln = {k: v for k, v in lora.items() if ".norm1.linear." in k or ".norm2.linear." in k}
    main = merge_lora(sd, {k: v for k, v in lora.items() if k not in adaln}, scale)
    merged = merge_lora(sd, adaln, scale)
    cond = {k: merged[k] for k in adaln}
    for k in adaln:
        cond[k.replace(".weight", ".bias")] = sd[k.replace(".weight", ".bias")]
    return main, cond
