"""Host-side (numpy) builders of the step-invariant tables the HIP path consumes: 3-D RoPE cos/sin, the 2B model's
3-D sincos positional table, the scheduler's alphas_cumprod and trailing timesteps.  Once per video, fp64/fp32 on
the host, uploaded once -- this removes the per-forward numpy rebuild and upload of the reference
(embeddings.py:433-446) from the denoise loop.

Reference behaviour followed (paths relative to the reference checkout):
  get_resize_crop_region_for_grid  pipelines/cogvideo/pipeline_cogvideox.py:62-77
  get_3d_rotary_pos_embed          models/embeddings.py:505-570, get_1d_rotary_pos_embed :673-727
  slicing ref / video              src/custom_cogvideox_pipe.py:223-235 (generalised from 1350 to (H/16)(W/16))
  get_3d_sincos_pos_embed          models/embeddings.py:81-125,150-180 (incl. the w-first quirk)
  alphas_cumprod / timesteps       schedulers/scheduling_ddim_cogvideox.py:95-123,199-218,285-291
"""
import numpy as np
import torch


def crop_region(grid_h, grid_w, base_w=45, base_h=30):
    r = grid_h / grid_w
    if r > base_h / base_w:
        rh, rw = base_h, int(round(base_h / grid_h * grid_w))
    else:
        rw, rh = base_w, int(round(base_w / grid_w * grid_h))
    top, left = int(round((base_h - rh) / 2.0)), int(round((base_w - rw) / 2.0))
    return (top, left), (top + rh, left + rw)


def _rope_axis(dim, pos):
    # fp32 with torch's CPU pow / cos / sin so the table is bit-identical to the reference's (numpy's libm differs
    # in the last ulp): theta ** (arange/dim) in fp32, outer product in fp32, each value repeated twice
    freqs = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
    ang = torch.outer(torch.from_numpy(np.ascontiguousarray(pos, dtype=np.float32)), freqs)
    return (ang.cos().repeat_interleave(2, dim=1).float().numpy(), ang.sin().repeat_interleave(2, dim=1).float().numpy())


def rope_tables(height, width, latent_frames, head_dim=64, patch=2, vae_sf=8):
    """cos, sin float32 [R + V, head_dim]: reference-image rows (temporal index 0) first, then the video rows
    (temporal index f+1), i.e. exactly the [ref | video] order of the packed sequence."""
    gh, gw = height // (vae_sf * patch), width // (vae_sf * patch)
    (t0, l0), (t1, l1) = crop_region(gh, gw, 720 // (vae_sf * patch), 480 // (vae_sf * patch))
    T = latent_frames + 1
    grid_h = np.linspace(t0, t1, gh, endpoint=False, dtype=np.float32)
    grid_w = np.linspace(l0, l1, gw, endpoint=False, dtype=np.float32)
    grid_t = np.linspace(0, T, T, endpoint=False, dtype=np.float32)
    dt, dh, dw = head_dim // 4, head_dim // 8 * 3, head_dim // 8 * 3
    out = []
    for k in range(2):
        a = _rope_axis(dt, grid_t)[k][:, None, None, :]
        b = _rope_axis(dh, grid_h)[k][None, :, None, :]
        c = _rope_axis(dw, grid_w)[k][None, None, :, :]
        full = np.concatenate([np.broadcast_to(a, (T, gh, gw, dt)), np.broadcast_to(b, (T, gh, gw, dh)),
                               np.broadcast_to(c, (T, gh, gw, dw))], axis=-1)
        out.append(np.ascontiguousarray(full.reshape(T * gh * gw, head_dim), dtype=np.float32))
    return out[0], out[1]


def _sincos_axis(dim, pos):
    omega = np.arange(dim // 2, dtype=np.float64) / (dim / 2.0)
    omega = 1.0 / 10000**omega
    ang = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)


def sincos_table(embed_dim, hp, wp, frames, spatial_scale=1.875, temporal_scale=1.0):
    """float32 [frames*hp*wp, embed_dim] additive table for the video tokens of the non-RoPE (2B) model."""
    ds, dt = 3 * embed_dim // 4, embed_dim // 4
    grid_h = np.arange(hp, dtype=np.float32) / spatial_scale
    grid_w = np.arange(wp, dtype=np.float32) / spatial_scale
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape(2, 1, hp, wp)
    spatial = np.concatenate([_sincos_axis(ds // 2, grid[0]), _sincos_axis(ds // 2, grid[1])], axis=1)
    temporal = _sincos_axis(dt, np.arange(frames, dtype=np.float32) / temporal_scale)
    spatial = np.repeat(spatial[None], frames, axis=0)
    temporal = np.repeat(temporal[:, None], hp * wp, axis=1)
    return np.concatenate([temporal, spatial], axis=-1).reshape(frames * hp * wp, embed_dim).astype(np.float32)


def alphas_cumprod(snr_shift_scale, num_train=1000, beta_start=0.00085, beta_end=0.012):
    """float64 [num_train]: scaled_linear betas, SNR shift, zero-terminal-SNR rescale (alphas[-1] == 0 exactly)."""
    # torch fp64 ops in the reference's order (torch.linspace / cumprod differ from numpy's in the last ulp)
    betas = torch.linspace(beta_start**0.5, beta_end**0.5, num_train, dtype=torch.float64) ** 2
    ac = torch.cumprod(1.0 - betas, dim=0)
    ac = ac / (snr_shift_scale + (1 - snr_shift_scale) * ac)
    s = ac.sqrt()
    s0, sT = s[0].clone(), s[-1].clone()
    s -= sT
    s *= s0 / (s0 - sT)
    return (s**2).numpy()


def trailing_timesteps(n, num_train=1000):
    return np.round(np.arange(num_train, 0, -num_train / n)).astype(np.int64) - 1
