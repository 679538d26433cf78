"""ORACLE -- test infrastructure only.  CPU restatement (plain torch functional ops) of the CogVideoX 3-D causal
VAE *decode* path.  Pinned against the imported reference (tests/golden/vae_*.npz).

Reference: diffusers/src/diffusers/models/autoencoders/autoencoder_kl_cogvideox.py
  CausalConv3d :120-137, SpatialNorm3D :167-188, ResnetBlock3D :278-319, MidBlock3D :502-533, UpBlock3D :616-655,
  Decoder3D :921-981, _decode :1231-1257, blend_v/h :1284-1298, tiled_decode :1374-1455;
  diffusers/src/diffusers/models/upsampling.py:384-412 (CogVideoXUpsample3D);
  pipelines/cogvideo/pipeline_cogvideox.py:346-351 (decode_latents), video_processor.py:89-113 (postprocess).
Weights: flat dict keyed by the reference's state-dict names ("decoder.conv_in.conv.weight", ...).
cfg: dict(block_out_channels, layers_per_block, norm_num_groups, latent_channels, scaling_factor,
          temporal_compression_ratio, sample_height, sample_width).
"""
import numpy as np
import torch
import torch.nn.functional as F


def causal_conv3d(sd, p, x, cache, key, new_cache):
    """CogVideoXCausalConv3d.forward; p = prefix up to and including 'conv.'."""
    w, b = sd[p + "weight"], sd[p + "bias"]
    kt = w.shape[2]
    if kt > 1:
        pre = [cache[key]] if (cache is not None and key in cache) else [x[:, :, :1]] * (kt - 1)
        x = torch.cat(pre + [x], dim=2)
        new_cache[key] = x[:, :, -kt + 1 :].clone()
        pad = w.shape[3] // 2
        x = F.pad(x, (pad, pad, pad, pad))
    return F.conv3d(x, w, b)


def spatial_norm(sd, p, f, zq, groups, cache, key, new_cache):
    if f.shape[2] > 1 and f.shape[2] % 2 == 1:
        z_first = F.interpolate(zq[:, :, :1], size=f[:, :, :1].shape[-3:])
        z_rest = F.interpolate(zq[:, :, 1:], size=f[:, :, 1:].shape[-3:])
        zq = torch.cat([z_first, z_rest], dim=2)
    else:
        zq = F.interpolate(zq, size=f.shape[-3:])
    cy = causal_conv3d(sd, p + "conv_y.conv.", zq, cache, key + "conv_y", new_cache)
    cb = causal_conv3d(sd, p + "conv_b.conv.", zq, cache, key + "conv_b", new_cache)
    nf = F.group_norm(f, groups, sd[p + "norm_layer.weight"], sd[p + "norm_layer.bias"], 1e-6)
    return nf * cy + cb


def resnet(sd, p, x, zq, groups, cache, key, new_cache):
    h = spatial_norm(sd, p + "norm1.", x, zq, groups, cache, key + "norm1.", new_cache)
    h = F.silu(h)
    h = causal_conv3d(sd, p + "conv1.conv.", h, cache, key + "conv1", new_cache)
    h = spatial_norm(sd, p + "norm2.", h, zq, groups, cache, key + "norm2.", new_cache)
    h = F.silu(h)
    h = causal_conv3d(sd, p + "conv2.conv.", h, cache, key + "conv2", new_cache)
    if (p + "conv_shortcut.weight") in sd:
        x = F.conv3d(x, sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"])
    return h + x


def upsample3d(sd, p, x, compress_time):
    """upsampling.py:384-412."""
    if compress_time:
        if x.shape[2] > 1 and x.shape[2] % 2 == 1:
            first = F.interpolate(x[:, :, 0], scale_factor=2.0)[:, :, None]
            rest = F.interpolate(x[:, :, 1:], scale_factor=2.0)
            x = torch.cat([first, rest], dim=2)
        elif x.shape[2] > 1:
            x = F.interpolate(x, scale_factor=2.0)
        else:
            x = F.interpolate(x.squeeze(2), scale_factor=2.0)[:, :, None]
    else:
        b, c, t, h, w = x.shape
        x = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
        x = F.interpolate(x, scale_factor=2.0)
        x = x.reshape(b, t, c, *x.shape[2:]).permute(0, 2, 1, 3, 4)
    b, c, t, h, w = x.shape
    x = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    x = F.conv2d(x, sd[p + "conv.weight"], sd[p + "conv.bias"], padding=1)
    return x.reshape(b, t, *x.shape[1:]).permute(0, 2, 1, 3, 4)


def decoder_forward(sd, cfg, z, cache):
    """CogVideoXDecoder3D.forward on one frame batch; returns (frames, new_cache)."""
    g = cfg["norm_num_groups"]
    nblk = len(cfg["block_out_channels"])
    nres = cfg["layers_per_block"] + 1
    tlevel = int(np.log2(cfg["temporal_compression_ratio"]))
    nc = {}
    h = causal_conv3d(sd, "decoder.conv_in.conv.", z, cache, "conv_in", nc)
    for i in range(2):
        h = resnet(sd, f"decoder.mid_block.resnets.{i}.", h, z, g, cache, f"mid.{i}.", nc)
    for bi in range(nblk):
        for ri in range(nres):
            h = resnet(sd, f"decoder.up_blocks.{bi}.resnets.{ri}.", h, z, g, cache, f"up.{bi}.{ri}.", nc)
        if bi != nblk - 1:
            h = upsample3d(sd, f"decoder.up_blocks.{bi}.upsamplers.0.", h, bi < tlevel)
    h = spatial_norm(sd, "decoder.norm_out.", h, z, g, cache, "norm_out.", nc)
    h = F.silu(h)
    h = causal_conv3d(sd, "decoder.conv_out.conv.", h, cache, "conv_out", nc)
    return h, nc


def frame_batches(num_frames, fbs=2):
    """autoencoder_kl_cogvideox.py:1237-1245: (3,2,2,...) latent frames for 13."""
    nb = max(num_frames // fbs, 1)
    rem = num_frames % fbs
    return [(fbs * i + (0 if i == 0 else rem), fbs * (i + 1) + rem) for i in range(nb)]


def decode_untiled(sd, cfg, z):
    cache, out = None, []
    for s, e in frame_batches(z.shape[2]):
        y, cache = decoder_forward(sd, cfg, z[:, :, s:e], cache)
        out.append(y)
    return torch.cat(out, dim=2)


def tile_geometry(cfg):
    """:1102-1114, 1400-1406 with the reference's int() truncations."""
    nlev = len(cfg["block_out_channels"]) - 1
    ts_h, ts_w = cfg["sample_height"] // 2, cfg["sample_width"] // 2
    tl_h, tl_w = int(ts_h / 2**nlev), int(ts_w / 2**nlev)
    fh, fw = 1 / 6, 1 / 5
    return dict(tl_h=tl_h, tl_w=tl_w, ov_h=int(tl_h * (1 - fh)), ov_w=int(tl_w * (1 - fw)),
                bl_h=int(ts_h * fh), bl_w=int(ts_w * fw), lim_h=ts_h - int(ts_h * fh), lim_w=ts_w - int(ts_w * fw))


def _blend_v(a, b, ext):
    ext = min(a.shape[3], b.shape[3], ext)
    for y in range(ext):
        b[:, :, :, y, :] = a[:, :, :, -ext + y, :] * (1 - y / ext) + b[:, :, :, y, :] * (y / ext)
    return b


def _blend_h(a, b, ext):
    ext = min(a.shape[4], b.shape[4], ext)
    for x in range(ext):
        b[:, :, :, :, x] = a[:, :, :, :, -ext + x] * (1 - x / ext) + b[:, :, :, :, x] * (x / ext)
    return b


def decode_tiled(sd, cfg, z):
    """tiled_decode :1374-1455 (blends are in place, raster order, so neighbours are read already blended)."""
    tg = tile_geometry(cfg)
    H, W = z.shape[3], z.shape[4]
    rows = []
    for i in range(0, H, tg["ov_h"]):
        row = []
        for j in range(0, W, tg["ov_w"]):
            row.append(decode_untiled(sd, cfg, z[:, :, :, i : i + tg["tl_h"], j : j + tg["tl_w"]]))
        rows.append(row)
    res_rows = []
    for i, row in enumerate(rows):
        res = []
        for j, tile in enumerate(row):
            if i > 0:
                tile = _blend_v(rows[i - 1][j], tile, tg["bl_h"])
            if j > 0:
                tile = _blend_h(row[j - 1], tile, tg["bl_w"])
            res.append(tile[:, :, :, : tg["lim_h"], : tg["lim_w"]])
        res_rows.append(torch.cat(res, dim=4))
    return torch.cat(res_rows, dim=3)


def decode(sd, cfg, z, tiling):
    tg = tile_geometry(cfg)
    if tiling and (z.shape[4] > tg["tl_w"] or z.shape[3] > tg["tl_h"]):
        return decode_tiled(sd, cfg, z)
    return decode_untiled(sd, cfg, z)


def decode_latents(sd, cfg, latents, tiling):
    """pipeline_cogvideox.py:346-351: latents [B,F,C,H,W] -> frames [B,3,F',H',W']."""
    z = latents.permute(0, 2, 1, 3, 4)
    z = 1 / cfg["scaling_factor"] * z
    return decode(sd, cfg, z, tiling)


def postprocess_np(video):
    """video_processor.py:99-113 + image_processor.py:227-240,196-209 (output_type='np'): [B,F,H,W,3] float32."""
    out = []
    for b in range(video.shape[0]):
        v = video[b].permute(1, 0, 2, 3)
        v = (v / 2 + 0.5).clamp(0, 1)
        out.append(v.cpu().permute(0, 2, 3, 1).float().numpy())
    return np.stack(out)


# ---------------------------------------------------------------------------------------------------------------------
# Reference-image ENCODE: one frame through CogVideoXEncoder3D, DiagonalGaussianDistribution.sample
# Reference: autoencoder_kl_cogvideox.py  ResnetBlock3D with plain GroupNorm :225-319 (spatial_norm_dim None),
#   DownBlock3D :322-421, Encoder3D :755-814, _encode :1177-1203, tiled_encode :1300-1372;
#   downsampling.py:322-353 (CogVideoXDownsample3D); autoencoders/vae.py:767-790; src/video_generate.py:26-38.
def resnet_gn(sd, p, x, groups, cache, key, new_cache):
    h = F.silu(F.group_norm(x, groups, sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6))
    h = causal_conv3d(sd, p + "conv1.conv.", h, cache, key + "conv1", new_cache)
    h = F.silu(F.group_norm(h, groups, sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6))
    h = causal_conv3d(sd, p + "conv2.conv.", h, cache, key + "conv2", new_cache)
    if (p + "conv_shortcut.weight") in sd:
        x = F.conv3d(x, sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"])
    return h + x


def downsample3d(sd, p, x, compress_time):
    """downsampling.py:322-353: temporal average pool (first frame kept when the count is odd), then a per-frame
    3x3 stride-2 conv on the input zero-padded at the right / bottom only."""
    if compress_time:
        B, C, Fr, H, W = x.shape
        t = x.permute(0, 3, 4, 1, 2).reshape(B * H * W, C, Fr)
        if Fr % 2 == 1:
            first, rest = t[..., 0], t[..., 1:]
            if rest.shape[-1] > 0:
                rest = F.avg_pool1d(rest, kernel_size=2, stride=2)
            t = torch.cat([first[..., None], rest], dim=-1)
        else:
            t = F.avg_pool1d(t, kernel_size=2, stride=2)
        x = t.reshape(B, H, W, C, t.shape[-1]).permute(0, 3, 4, 1, 2)
    x = F.pad(x, (0, 1, 0, 1))
    B, C, Fr, H, W = x.shape
    y = F.conv2d(x.permute(0, 2, 1, 3, 4).reshape(B * Fr, C, H, W), sd[p + "conv.weight"], sd[p + "conv.bias"], stride=2)
    return y.reshape(B, Fr, y.shape[1], y.shape[2], y.shape[3]).permute(0, 2, 1, 3, 4)


def encoder_forward(sd, cfg, x, cache=None):
    groups = cfg["norm_num_groups"]
    nb = len(cfg["block_out_channels"])
    tlevel = int(np.log2(cfg["temporal_compression_ratio"]))
    nc = {}
    h = causal_conv3d(sd, "encoder.conv_in.conv.", x, cache, "conv_in", nc)
    for b in range(nb):
        for i in range(cfg["layers_per_block"]):
            p = f"encoder.down_blocks.{b}.resnets.{i}."
            h = resnet_gn(sd, p, h, groups, cache, p, nc)
        if b != nb - 1:
            h = downsample3d(sd, f"encoder.down_blocks.{b}.downsamplers.0.", h, b < tlevel)
    for i in range(2):
        p = f"encoder.mid_block.resnets.{i}."
        h = resnet_gn(sd, p, h, groups, cache, p, nc)
    h = F.silu(F.group_norm(h, groups, sd["encoder.norm_out.weight"], sd["encoder.norm_out.bias"], 1e-6))
    h = causal_conv3d(sd, "encoder.conv_out.conv.", h, cache, "conv_out", nc)
    return h, nc


def encode_untiled(sd, cfg, x):
    assert x.shape[2] == 1, "the path encodes one reference frame; video encode is outside it"
    return encoder_forward(sd, cfg, x)[0]


def encode_tile_geometry(cfg):
    """:1102-1114, 1317-1323 with the int() truncations (sample-space tiles, latent-space blends)."""
    nlev = len(cfg["block_out_channels"]) - 1
    ts_h, ts_w = cfg["sample_height"] // 2, cfg["sample_width"] // 2
    tl_h, tl_w = int(ts_h / 2**nlev), int(ts_w / 2**nlev)
    fh, fw = 1 / 6, 1 / 5
    return dict(ts_h=ts_h, ts_w=ts_w, ov_h=int(ts_h * (1 - fh)), ov_w=int(ts_w * (1 - fw)), bl_h=int(tl_h * fh),
                bl_w=int(tl_w * fw), lim_h=tl_h - int(tl_h * fh), lim_w=tl_w - int(tl_w * fw))


def encode_tiled(sd, cfg, x):
    tg = encode_tile_geometry(cfg)
    H, W = x.shape[3], x.shape[4]
    rows = []
    for i in range(0, H, tg["ov_h"]):
        rows.append([encode_untiled(sd, cfg, x[:, :, :, i : i + tg["ts_h"], j : j + tg["ts_w"]]) for j in range(0, W, tg["ov_w"])])
    res_rows = []
    for i, row in enumerate(rows):
        res = []
        for j, tile in enumerate(row):
            if i > 0:
                tile = _blend_v(rows[i - 1][j], tile, tg["bl_h"])
            if j > 0:
                tile = _blend_h(row[j - 1], tile, tg["bl_w"])
            res.append(tile[:, :, :, : tg["lim_h"], : tg["lim_w"]])
        res_rows.append(torch.cat(res, dim=4))
    return torch.cat(res_rows, dim=3)


def encode_moments(sd, cfg, x, tiling):
    tg = encode_tile_geometry(cfg)
    if tiling and (x.shape[4] > tg["ts_w"] or x.shape[3] > tg["ts_h"]):
        return encode_tiled(sd, cfg, x)
    return encode_untiled(sd, cfg, x)


def encode_image(sd, cfg, x, noise, tiling):
    """x [1,3,1,H,W] in [-1,1], noise [1,C,1,h,w] (the randn the caller draws) -> ref latents [1,1,C,h,w] (src/video_generate.py:35-38)."""
    mom = encode_moments(sd, cfg, x, tiling)
    mean, logvar = torch.chunk(mom, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    std = torch.exp(0.5 * logvar)
    z = (mean + std * noise) * cfg["scaling_factor"]
    return z.permute(0, 2, 1, 3, 4)
