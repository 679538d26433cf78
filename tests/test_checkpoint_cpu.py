"""Checkpoint ingest, host logic only: file discovery, key renames, shard order, configs.
Rules follow the reference's loaders/lora_base.py:314-354, utils/state_dict_utils.py:39-51 and src/inference.py:83-105."""
import importlib
import json
import os

import pytest
import torch
from safetensors.torch import save_file

ck = importlib.import_module("disentangled-subject-to-vid_amd.checkpoint")


def _t(*s):
    return torch.arange(int(torch.tensor(s).prod()), dtype=torch.float32).reshape(*s)


def test_find_lora_prefers_canonical_name_and_skips_training_state(tmp_path):
    for n in ("optimizer.safetensors", "scheduler_state.safetensors", "checkpoint-5.safetensors",
              "pytorch_lora_weights.safetensors", "other.safetensors", "notes.txt"):
        (tmp_path / n).write_bytes(b"")
    assert ck.find_lora_file(str(tmp_path)) == str(tmp_path / "pytorch_lora_weights.safetensors")


def test_find_lora_single_other_name(tmp_path):
    (tmp_path / "my_adapter.safetensors").write_bytes(b"")
    (tmp_path / "optimizer.safetensors").write_bytes(b"")
    assert ck.find_lora_file(str(tmp_path)).endswith("my_adapter.safetensors")


def test_find_lora_ambiguous_and_missing(tmp_path):
    with pytest.raises(FileNotFoundError):
        ck.find_lora_file(str(tmp_path))
    (tmp_path / "a.safetensors").write_bytes(b"")
    (tmp_path / "b.safetensors").write_bytes(b"")
    with pytest.raises(ValueError, match="more than one weights file"):
        ck.find_lora_file(str(tmp_path))


@pytest.mark.parametrize("old,new", [
    ("transformer_blocks.0.attn1.to_q_lora.down.weight", "transformer_blocks.0.attn1.to_q.lora_A.weight"),
    ("transformer_blocks.0.attn1.to_out_lora.up.weight", "transformer_blocks.0.attn1.to_out.0.lora_B.weight"),
    ("transformer_blocks.3.ff.net.2.lora.up.weight", "transformer_blocks.3.ff.net.2.lora_B.weight"),
    ("patch_embed.proj.lora.down.weight", "patch_embed.proj.lora_A.weight"),
    ("transformer_blocks.1.norm1.linear.lora_A.weight", "transformer_blocks.1.norm1.linear.lora_A.weight"),
])
def test_peft_key(old, new):
    assert ck.peft_key(old) == new


def test_read_lora_selects_transformer_and_pairs_halves(tmp_path, capsys):
    sd = {
        "transformer.transformer_blocks.0.attn1.to_q.lora_A.weight": _t(4, 8),
        "transformer.transformer_blocks.0.attn1.to_q.lora_B.weight": _t(8, 4),
        "transformer.patch_embed.proj.lora.down.weight": _t(4, 16, 2, 2),   # old-style names are renamed
        "transformer.patch_embed.proj.lora.up.weight": _t(8, 4),
        "transformer.transformer_blocks.1.ff.net.2.lora_A.weight": _t(4, 8),  # half missing -> reported, skipped
        "text_encoder.layer.0.q.lora_A.weight": _t(4, 8),                    # other sub-models are ignored
    }
    save_file(sd, str(tmp_path / "pytorch_lora_weights.safetensors"))
    out = ck.read_lora(str(tmp_path))
    assert sorted(out) == ["patch_embed.proj.weight", "transformer_blocks.0.attn1.to_q.weight"]
    A, B = out["transformer_blocks.0.attn1.to_q.weight"]
    assert torch.equal(A, sd["transformer.transformer_blocks.0.attn1.to_q.lora_A.weight"])
    assert torch.equal(B, sd["transformer.transformer_blocks.0.attn1.to_q.lora_B.weight"])
    assert out["patch_embed.proj.weight"][0].shape == (4, 16, 2, 2)
    assert "Unexpected keys" in capsys.readouterr().out


def test_model_files_single_and_sharded(tmp_path):
    d = tmp_path / "transformer"
    d.mkdir()
    with pytest.raises(FileNotFoundError):
        ck.model_files(str(d))
    save_file({"b.weight": _t(2, 2)}, str(d / "diffusion_pytorch_model-00002-of-00002.safetensors"))
    save_file({"a.weight": _t(3, 2)}, str(d / "diffusion_pytorch_model-00001-of-00002.safetensors"))
    (d / "diffusion_pytorch_model.safetensors.index.json").write_text(json.dumps({"weight_map": {
        "b.weight": "diffusion_pytorch_model-00002-of-00002.safetensors",
        "a.weight": "diffusion_pytorch_model-00001-of-00002.safetensors"}}))
    files = ck.model_files(str(d))
    assert [os.path.basename(f)[-26:-12] for f in files] == ["00001-of-00002", "00002-of-00002"]
    assert [k for k, _ in ck.iter_tensors(files)] == ["a.weight", "b.weight"]
    save_file({"c.weight": _t(1, 2)}, str(d / "diffusion_pytorch_model.safetensors"))
    assert ck.model_files(str(d)) == [str(d / "diffusion_pytorch_model.safetensors")]


def test_configs_from_model_root(tmp_path):
    for sub in ("transformer", "vae", "scheduler"):
        (tmp_path / sub).mkdir()
    (tmp_path / "transformer" / "config.json").write_text(json.dumps({
        "_class_name": "CogVideoXTransformer3DModel", "num_layers": 42, "num_attention_heads": 48,
        "attention_head_dim": 64, "use_rotary_positional_embeddings": True, "text_embed_dim": 4096,
        "sample_width": 90, "dropout": 0.0}))
    (tmp_path / "vae" / "config.json").write_text(json.dumps({
        "scaling_factor": 0.7, "block_out_channels": [128, 256, 256, 512], "layers_per_block": 3, "act_fn": "silu"}))
    (tmp_path / "scheduler" / "scheduler_config.json").write_text(json.dumps({"snr_shift_scale": 1.0}))
    cfg = ck.transformer_config(str(tmp_path))
    assert (cfg.num_layers, cfg.num_attention_heads, cfg.inner_dim) == (42, 48, 3072)
    assert cfg.use_rotary_positional_embeddings and cfg.snr_shift_scale == 1.0 and cfg.vae_scaling_factor == 0.7
    v = ck.vae_config(str(tmp_path))
    assert v.block_out_channels == (128, 256, 256, 512) and v.scaling_factor == 0.7


def test_resize_token_embeddings_is_a_truncation_for_the_shipped_sizes():
    """src/inference.py:179-189: `<cls>` added to the 32 100-entry tokenizer, then resize_token_embeddings(32 101) on a 32 128-row table"""
    tab = _t(12, 4)
    out = ck.resize_token_embeddings(tab, 9)
    assert torch.equal(out, tab[:9]) and out.is_contiguous()
    assert torch.equal(ck.resize_token_embeddings(tab, 12), tab)
    with pytest.raises(ValueError, match="grows the table"):
        ck.resize_token_embeddings(tab, 13)
    grown = ck.resize_token_embeddings(tab, 13, new_rows=torch.full((1, 4), 5.0))
    assert grown.shape == (13, 4) and torch.equal(grown[:12], tab) and bool((grown[12] == 5).all())


def test_resize_matches_transformers():
    from transformers import T5Config, T5EncoderModel

    m = T5EncoderModel(T5Config(vocab_size=40, d_model=16, d_kv=4, num_heads=4, d_ff=32, num_layers=1, feed_forward_proj="gated-gelu"))
    w = m.shared.weight.detach().clone()
    m.resize_token_embeddings(33)
    assert torch.equal(m.shared.weight.detach(), ck.resize_token_embeddings(w, 33))


def test_t5_config_and_sharded_model_files(tmp_path):
    d = tmp_path / "text_encoder"
    d.mkdir()
    (d / "config.json").write_text(json.dumps(dict(vocab_size=32128, d_model=64, d_kv=16, num_heads=4, d_ff=128, num_layers=3,
                                                   relative_attention_num_buckets=32, relative_attention_max_distance=128,
                                                   layer_norm_epsilon=1e-6, feed_forward_proj="gated-gelu", model_type="t5")))
    c = ck.t5_config(str(d), num_tokens=32101)
    assert (c.vocab_size, c.d_model, c.num_layers) == (32101, 64, 3)
    assert ck.t5_config(str(d)).vocab_size == 32128
    (d / "model.safetensors.index.json").write_text(json.dumps({"weight_map": {"a": "model-00002-of-00002.safetensors", "b": "model-00001-of-00002.safetensors"}}))
    assert [os.path.basename(f) for f in ck.model_files(str(d), stem="model")] == ["model-00001-of-00002.safetensors", "model-00002-of-00002.safetensors"]
    (d / "config.json").write_text(json.dumps(dict(vocab_size=10, feed_forward_proj="relu")))
    with pytest.raises(ValueError, match="not T5 v1.1"):
        ck.t5_config(str(d))


def test_load_t5_streams_one_table_and_only_encoder_tensors(tmp_path):
    class Fake:
        def __init__(self):
            self.cfg = type("C", (), {"vocab_size": 5})()
            self.got = {}

        def load_state_dict(self, sd):
            for k, t in sd.items():
                assert k not in self.got
                self.got[k] = t

    d = tmp_path / "te"
    d.mkdir()
    save_file({"shared.weight": _t(8, 2), "encoder.embed_tokens.weight": _t(8, 2), "encoder.final_layer_norm.weight": _t(2),
               "decoder.x": _t(1), "lm_head.weight": _t(8, 2)}, str(d / "model.safetensors"))
    f = Fake()
    assert ck.load_t5(str(d), model=f, num_tokens=5) is f
    assert sorted(f.got) == ["encoder.final_layer_norm.weight", "shared.weight"]
    assert torch.equal(f.got["shared.weight"], _t(8, 2)[:5])
    with pytest.raises(ValueError, match="built for 5 tokens"):
        ck.load_t5(str(d), model=f, num_tokens=6)


def test_load_vae_filters_and_rejects(tmp_path):
    class Fake:
        def load_state_dict(self, sd):
            self.keys = sorted(sd)

    d = tmp_path / "vae"
    d.mkdir()
    save_file({"decoder.a": _t(1), "encoder.b": _t(1)}, str(d / "diffusion_pytorch_model.safetensors"))
    f = Fake()
    assert ck.load_vae(f, str(d)) == ["decoder", "encoder"] and f.keys == ["decoder.a", "encoder.b"]
    assert ck.load_vae(f, str(d), with_encoder=False) == ["decoder"] and f.keys == ["decoder.a"]
    save_file({"decoder.a": _t(1)}, str(d / "diffusion_pytorch_model.safetensors"))
    with pytest.raises(FileNotFoundError, match="encoder"):
        ck.load_vae(f, str(d))
    save_file({"decoder.a": _t(1), "encoder.b": _t(1), "post_quant_conv.weight": _t(1)}, str(d / "diffusion_pytorch_model.safetensors"))
    with pytest.raises(ValueError, match="outside encoder / decoder"):
        ck.load_vae(f, str(d))
