"""Synthetic, seeded weights of the CogVideoX transformer with the reference's state-dict keys and shapes
 -- there are no real checkpoints offline.  `parity=True` uses larger-variance weights,
non-zero biases and LN affines so that gates, modulation and every residual term contribute measurably;
`parity=False` is the near-identity N(0, 0.02^2) initialisation used for timing runs."""
import torch

from .config import TransformerConfig


def state_dict_shapes(cfg: TransformerConfig):
    D, TE, TX = cfg.inner_dim, cfg.time_embed_dim, cfg.text_embed_dim
    C, p = cfg.in_channels, cfg.patch_size
    s = {
        "patch_embed.proj.weight": (D, C, p, p), "patch_embed.proj.bias": (D,),
        "patch_embed.text_proj.weight": (D, TX), "patch_embed.text_proj.bias": (D,),
        "time_embedding.linear_1.weight": (TE, D), "time_embedding.linear_1.bias": (TE,),
        "time_embedding.linear_2.weight": (TE, TE), "time_embedding.linear_2.bias": (TE,),
        "norm_final.weight": (D,), "norm_final.bias": (D,),
        "norm_out.linear.weight": (2 * D, TE), "norm_out.linear.bias": (2 * D,),
        "norm_out.norm.weight": (D,), "norm_out.norm.bias": (D,),
        "proj_out.weight": (cfg.out_channels * p * p, D), "proj_out.bias": (cfg.out_channels * p * p,),
    }
    for i in range(cfg.num_layers):
        b = f"transformer_blocks.{i}."
        for n in ("norm1", "norm2"):
            s[b + n + ".linear.weight"] = (6 * D, TE)
            s[b + n + ".linear.bias"] = (6 * D,)
            s[b + n + ".norm.weight"] = (D,)
            s[b + n + ".norm.bias"] = (D,)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            s[b + "attn1." + n + ".weight"] = (D, D)
            s[b + "attn1." + n + ".bias"] = (D,)
        for n in ("norm_q", "norm_k"):
            s[b + "attn1." + n + ".weight"] = (64,)
            s[b + "attn1." + n + ".bias"] = (64,)
        s[b + "ff.net.0.proj.weight"] = (4 * D, D)
        s[b + "ff.net.0.proj.bias"] = (4 * D,)
        s[b + "ff.net.2.weight"] = (D, 4 * D)
        s[b + "ff.net.2.bias"] = (D,)
    return s


LORA_TARGETS = ("to_q", "to_k", "to_v", "to_out.0", "proj", "text_proj", "norm1.linear", "norm2.linear", "ff.net.2")


def lora_target_keys(cfg):
    """weight keys the reference's LoraConfig(target_modules=...) matches (src/inference.py:218-225; PEFT matches
    module-name suffixes, so "proj" also catches ff.net.0.proj and patch_embed.proj)"""
    keys = []
    for k in state_dict_shapes(cfg):
        if not k.endswith(".weight"):
            continue
        mod = k[: -len(".weight")]
        if any(mod == t or mod.endswith("." + t) for t in LORA_TARGETS):
            keys.append(k)
    return keys


def synthetic_state_dict(cfg, seed=1234, device="cpu", dtype=torch.float32, parity=False):
    gen = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for k, shape in state_dict_shapes(cfg).items():
        is_norm = ".norm" in k or k.startswith("norm_final") or "norm_q" in k or "norm_k" in k
        if len(shape) >= 2:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            std = (0.7 / fan_in**0.5) if parity else 0.02
            if parity and ".linear." in k:
                std = 0.5 / fan_in**0.5
            t = torch.randn(shape, generator=gen, device=device, dtype=torch.float32) * std
        elif is_norm and k.endswith("weight") and ".linear." not in k:
            t = torch.ones(shape, device=device)
            if parity:
                t = t + 0.2 * torch.randn(shape, generator=gen, device=device)
        else:
            t = torch.zeros(shape, device=device)
            if parity:
                t = 0.1 * torch.randn(shape, generator=gen, device=device)
        sd[k] = t.to(dtype)
    return sd


def synthetic_lora(cfg, rank=128, seed=99, device="cpu", std=0.02):
    gen = torch.Generator(device=device).manual_seed(seed)
    shapes = state_dict_shapes(cfg)
    out = {}
    for k in lora_target_keys(cfg):
        shp = shapes[k]
        A = torch.randn((rank,) + tuple(shp[1:]), generator=gen, device=device) * std
        B = torch.randn((shp[0], rank), generator=gen, device=device) * std
        out[k] = (A, B)
    return out


def vae_decoder_shapes(cfg):
    """state-dict keys/shapes of AutoencoderKLCogVideoX.decoder (autoencoder_kl_cogvideox.py:860-919)"""
    rc = list(reversed(cfg.block_out_channels))
    Cz = cfg.latent_channels
    s = {}

    def conv(name, cin, cout, k):
        s[name + ".weight"] = (cout, cin) + k
        s[name + ".bias"] = (cout,)

    def snorm(name, C):
        s[name + ".norm_layer.weight"] = (C,)
        s[name + ".norm_layer.bias"] = (C,)
        conv(name + ".conv_y.conv", Cz, C, (1, 1, 1))
        conv(name + ".conv_b.conv", Cz, C, (1, 1, 1))

    def resnet(name, cin, cout):
        snorm(name + ".norm1", cin)
        conv(name + ".conv1.conv", cin, cout, (3, 3, 3))
        snorm(name + ".norm2", cout)
        conv(name + ".conv2.conv", cout, cout, (3, 3, 3))
        if cin != cout:
            conv(name + ".conv_shortcut", cin, cout, (1, 1, 1))

    conv("decoder.conv_in.conv", Cz, rc[0], (3, 3, 3))
    for i in range(2):
        resnet(f"decoder.mid_block.resnets.{i}", rc[0], rc[0])
    prev = rc[0]
    for b, ch in enumerate(rc):
        for i in range(cfg.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{b}.resnets.{i}", prev if i == 0 else ch, ch)
        prev = ch
        if b != len(rc) - 1:
            conv(f"decoder.up_blocks.{b}.upsamplers.0.conv", ch, ch, (3, 3))
    snorm("decoder.norm_out", rc[-1])
    conv("decoder.conv_out.conv", rc[-1], cfg.out_channels, (3, 3, 3))
    return s


def synthetic_vae_state_dict(cfg, seed=4321, device="cpu", dtype=torch.float32):
    """seeded decoder weights scaled so activations stay O(1) through the stack"""
    gen = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for k, shape in vae_decoder_shapes(cfg).items():
        if len(shape) >= 2:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = torch.randn(shape, generator=gen, device=device) * (1.0 / fan_in**0.5)
            if "conv_y" in k:
                t = t * 0.5
        elif k.endswith("norm_layer.weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=gen, device=device)
        elif "conv_y.conv.bias" in k:
            t = 1.0 + 0.1 * torch.randn(shape, generator=gen, device=device)
        else:
            t = 0.05 * torch.randn(shape, generator=gen, device=device)
        sd[k] = t.to(dtype)
    return sd


def vae_encoder_shapes(cfg):
    """state-dict keys / shapes of CogVideoXEncoder3D (autoencoder_kl_cogvideox.py:689-750) for a VAEConfig"""
    bo, out = list(cfg.block_out_channels), {}

    def conv(name, cin, cout, k):
        out[name + ".weight"] = (cout, cin) + k
        out[name + ".bias"] = (cout,)

    def resnet(name, cin, cout):
        out[name + ".norm1.weight"] = (cin,); out[name + ".norm1.bias"] = (cin,)
        out[name + ".norm2.weight"] = (cout,); out[name + ".norm2.bias"] = (cout,)
        conv(name + ".conv1.conv", cin, cout, (3, 3, 3))
        conv(name + ".conv2.conv", cout, cout, (3, 3, 3))
        if cin != cout:
            conv(name + ".conv_shortcut", cin, cout, (1, 1, 1))

    conv("encoder.conv_in.conv", cfg.out_channels, bo[0], (3, 3, 3))
    prev = bo[0]
    for b, co in enumerate(bo):
        for i in range(cfg.layers_per_block):
            resnet(f"encoder.down_blocks.{b}.resnets.{i}", prev if i == 0 else co, co)
        prev = co
        if b != len(bo) - 1:
            conv(f"encoder.down_blocks.{b}.downsamplers.0.conv", co, co, (3, 3))
    for i in range(2):
        resnet(f"encoder.mid_block.resnets.{i}", prev, prev)
    out["encoder.norm_out.weight"] = (prev,); out["encoder.norm_out.bias"] = (prev,)
    conv("encoder.conv_out.conv", prev, 2 * cfg.latent_channels, (3, 3, 3))
    return out


def synthetic_vae_encoder_state_dict(cfg, seed=8765, device="cpu", dtype=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd = {}
    for k, shp in vae_encoder_shapes(cfg).items():
        if "norm" in k and k.endswith("weight"):
            v = 1.0 + 0.2 * torch.randn(shp, generator=g)
        elif len(shp) == 1:
            v = 0.1 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            v = torch.randn(shp, generator=g) * (1.0 / fan_in) ** 0.5
        sd[k] = v.to(dtype).to(device)
    return sd


def t5_encoder_shapes(cfg):
    """HF state-dict keys / shapes of T5EncoderModel (v1.1, gated-gelu) for a t5.T5Config"""
    d, inner, F = cfg.d_model, cfg.num_heads * cfg.d_kv, cfg.d_ff
    out = {"shared.weight": (cfg.vocab_size, d), "encoder.final_layer_norm.weight": (d,),
           "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight": (cfg.relative_attention_num_buckets, cfg.num_heads)}
    for i in range(cfg.num_layers):
        p = f"encoder.block.{i}."
        for n in "qkv":
            out[p + f"layer.0.SelfAttention.{n}.weight"] = (inner, d)
        out[p + "layer.0.SelfAttention.o.weight"] = (d, inner)
        out[p + "layer.0.layer_norm.weight"] = (d,)
        out[p + "layer.1.DenseReluDense.wi_0.weight"] = (F, d)
        out[p + "layer.1.DenseReluDense.wi_1.weight"] = (F, d)
        out[p + "layer.1.DenseReluDense.wo.weight"] = (d, F)
        out[p + "layer.1.layer_norm.weight"] = (d,)
    return out


def synthetic_t5_state_dict(cfg, seed=2468, device="cpu", dtype=torch.float32, gain=0.7):
    """gain scales the Linear weights (std = gain / sqrt(fan_in)); T5 has no 1/sqrt(d) in its scores, so gains >~ 1 give
    near-one-hot attention and a stack whose bf16 run is chaotic (bf16 vs fp32 of the SAME model differ by 25 %)."""
    gdev = "cpu" if str(device) == "cpu" else device  # full-size models (4.7 B parameters) are drawn on the device
    g = torch.Generator(device=gdev).manual_seed(seed)
    sd = {}
    for k, shp in t5_encoder_shapes(cfg).items():
        if "layer_norm" in k:
            v = 1.0 + 0.2 * torch.randn(shp, generator=g, device=gdev)
        elif "relative_attention_bias" in k:
            v = 0.5 * torch.randn(shp, generator=g, device=gdev)
        elif k == "shared.weight":
            v = torch.randn(shp, generator=g, device=gdev)
        else:
            v = torch.randn(shp, generator=g, device=gdev) * (gain / shp[1] ** 0.5)
        sd[k] = v.to(dtype).to(device)
    return sd
