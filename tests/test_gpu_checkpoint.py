"""GPU parity of checkpoint ingest: weights that travel HF-layout safetensors shards + a
`pytorch_lora_weights.safetensors` (PEFT key names under the `transformer.` prefix) -> s2v_load_weight / s2v_merge_lora
must give bit-identical results to handing the same tensors over as a state dict, and match the CPU oracle with the
LoRA merged at alpha / r (src/inference.py:83-105,218-229)."""
import importlib
import json

import numpy as np
import pytest
import torch
from safetensors.torch import save_file

from oracle import transformer_ref as tr
from oracle import vae_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ck = importlib.import_module("disentangled-subject-to-vid_amd.checkpoint")


def _write_model(tmp_path, sd, lora, shards=2):
    root = tmp_path / "model"
    (root / "transformer").mkdir(parents=True)
    keys = sorted(sd)
    per = (len(keys) + shards - 1) // shards
    wmap = {}
    for i in range(shards):
        name = f"diffusion_pytorch_model-{i + 1:05d}-of-{shards:05d}.safetensors"
        part = {k: sd[k].contiguous() for k in keys[i * per:(i + 1) * per]}
        save_file(part, str(root / "transformer" / name))
        wmap.update({k: name for k in part})
    (root / "transformer" / "diffusion_pytorch_model.safetensors.index.json").write_text(json.dumps({"weight_map": wmap}))
    ldir = tmp_path / "lora"
    ldir.mkdir()
    flat = {}
    for k, (A, B) in lora.items():
        stem = k[:-len(".weight")]
        flat[f"transformer.{stem}.lora_A.weight"] = A.contiguous()
        flat[f"transformer.{stem}.lora_B.weight"] = B.contiguous()
    save_file(flat, str(ldir / "pytorch_lora_weights.safetensors"))
    save_file({"x": torch.zeros(1)}, str(ldir / "optimizer.safetensors"))
    return root, ldir


@pytest.mark.parametrize("dt_name", ["f32", "bf16"])
def test_safetensors_ingest_matches_state_dict_and_oracle(s2v, tmp_path, dt_name):
    dt = torch.float32 if dt_name == "f32" else torch.bfloat16
    cfg = s2v.tiny(use_rope=True, heads=3, layers=2, text_dim=128, temb=64)
    cfg.max_text_seq_length = 7
    sd = s2v.weights.synthetic_state_dict(cfg, seed=21, parity=True)
    lora = s2v.weights.synthetic_lora(cfg, rank=8, seed=22, std=0.05)
    root, ldir = _write_model(tmp_path, {k: v.to(dt) for k, v in sd.items()}, lora)
    g = torch.Generator().manual_seed(23)
    B, F, C, H, W, T = 2, 3, 16, 16, 24, 7
    lat = torch.randn(B, F, C, H, W, generator=g).to(dt)
    text = torch.randn(B, T, 128, generator=g).to(dt)
    ref = (torch.randn(1, 1, C, H, W, generator=g) * 0.7).to(dt)
    ts = torch.tensor([321, 321])
    ref_rope, rope = tr.pipeline_rope(H * 8, W * 8, F)
    kw = dict(image_rotary_emb=tuple(x.to(DEV) for x in rope), ref_image_rotary_emb=tuple(x.to(DEV) for x in ref_rope))

    def run(m):
        y = m(hidden_states=lat.to(DEV), encoder_hidden_states=text.to(DEV), ref_img_states=ref.to(DEV),
              timestep=ts.to(DEV), return_dict=False, eval=True, **kw)[0]
        torch.cuda.synchronize()
        return y.float().cpu()

    m1 = s2v.HipCogVideoXTransformer3DModel(cfg, dt, DEV)
    merged_keys = ck.load_transformer(m1, str(root / "transformer"), str(ldir), lora_alpha=4, rank=8)
    assert merged_keys == sorted(lora)
    y1 = run(m1)
    m2 = s2v.HipCogVideoXTransformer3DModel(cfg, dt, DEV)
    m2.load_state_dict({k: v.to(dt) for k, v in sd.items()}, lora=lora, lora_scale=0.5)
    y2 = run(m2)
    assert torch.equal(y1, y2), "file ingest and state-dict ingest must be bit-identical"

    ocfg = dict(num_heads=3, num_layers=2, use_rope=True, norm_eps=1e-5)
    merged = tr.merge_lora({k: v.to(dt).float() for k, v in sd.items()}, lora, 0.5)
    with torch.no_grad():
        exp = tr.transformer_forward({k: v.to(dt) for k, v in merged.items()}, ocfg, lat, text, ref, ts, rope, ref_rope).float()
    if dt_name == "f32":
        assert (y1 - exp).abs().max().item() <= 2e-4
    else:
        assert ((y1 - exp).norm() / exp.norm()).item() <= 2e-2


def test_vae_decoder_ingest_from_safetensors(s2v, tmp_path):
    cfgd = dict(block_out_channels=(16, 16, 32, 32), layers_per_block=1, norm_num_groups=4, latent_channels=16,
                sample_height=96, sample_width=160, scaling_factor=0.7, temporal_compression_ratio=4)
    cfg = s2v.VAEConfig(**cfgd)
    sd = s2v.weights.synthetic_vae_state_dict(cfg, seed=31)
    (tmp_path / "vae").mkdir()
    extra = {"encoder.conv_in.conv.weight": torch.zeros(4, 3, 3, 3, 3)}   # encoder tensors are skipped by the decoder
    save_file({**{k: v.contiguous() for k, v in sd.items()}, **extra}, str(tmp_path / "vae" / "diffusion_pytorch_model.safetensors"))
    (tmp_path / "vae" / "config.json").write_text(json.dumps({k: list(v) if isinstance(v, tuple) else v for k, v in cfgd.items()}))
    cfg2 = ck.vae_config(str(tmp_path))
    assert cfg2 == cfg
    v1 = s2v.HipAutoencoderKLCogVideoX(cfg2, torch.float32, DEV)
    ck.load_vae_decoder(v1, str(tmp_path / "vae"))
    v2 = s2v.HipAutoencoderKLCogVideoX(cfg, torch.float32, DEV)
    v2.load_state_dict(sd)
    lat = torch.randn(1, 3, 16, 12, 20, generator=torch.Generator().manual_seed(32))
    y1 = v1.decode_latents(lat.to(DEV))
    y2 = v2.decode_latents(lat.to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(y1, y2)
    with torch.no_grad():
        exp = vae_ref.decode_latents(dict(sd), cfgd, lat, False).float()
    assert (y1.float().cpu() - exp).abs().max().item() <= 1e-3


def test_unknown_lora_keys_are_reported_not_fatal(s2v, capsys):
    """src/inference.py:96-105 prints adapter keys it cannot place and continues; so does the ingest"""
    cfg = s2v.tiny(use_rope=True, heads=2, layers=1)
    sd = s2v.weights.synthetic_state_dict(cfg, seed=1, parity=True)
    lora = s2v.weights.synthetic_lora(cfg, rank=4, seed=2, std=0.1)
    lora["transformer_blocks.0.attn1.to_z.weight"] = (torch.zeros(4, 128), torch.zeros(128, 4))
    eng = s2v.S2VEngine(cfg, torch.float32, DEV)
    eng.load_state_dict(sd, lora=lora, lora_scale=0.5)
    assert eng.unexpected_lora_keys == ["transformer_blocks.0.attn1.to_z.weight"]
    assert "unexpected keys" in capsys.readouterr().out
    eng2 = s2v.S2VEngine(cfg, torch.float32, DEV)
    del lora["transformer_blocks.0.attn1.to_z.weight"]
    eng2.load_state_dict(sd, lora=lora, lora_scale=0.5)
    assert torch.equal(eng.weight_arena(), eng2.weight_arena())
