"""Checkpoint ingest for the path: HF-layout safetensors of the CogVideoX transformer / VAE and the
subject-LoRA safetensors, streamed tensor by tensor into the context-owned, re-packed device buffers
(s2v_load_weight / s2v_merge_lora / s2v_vae_load_weight).  Host-side file handling only; no arithmetic here.

Reference behaviour followed (paths relative to the reference checkout):
  transformer / VAE files   diffusers `from_pretrained` layout: `<dir>/diffusion_pytorch_model.safetensors` or the
                            sharded `...-0000k-of-0000n.safetensors` + `diffusion_pytorch_model.safetensors.index.json`
                            (models/modeling_utils.py), as loaded by src/inference.py:191-207
  LoRA file discovery       `_best_guess_weight_name`, loaders/lora_base.py:314-354: the single `.safetensors` of the
                            checkpoint dir, ignoring names containing scheduler / optimizer / checkpoint, preferring
                            `pytorch_lora_weights.safetensors`
  LoRA key handling         src/inference.py:83-92: keep `transformer.*`, strip the prefix, then
                            `convert_unet_state_dict_to_peft` (utils/state_dict_utils.py:39-51,248-253)
  text encoder              `T5EncoderModel.from_pretrained(subfolder="text_encoder")` + `resize_token_embeddings(len(tokenizer))`
                            after the `<cls>` token was added (src/inference.py:177-189): transformers' layout
                            `model.safetensors` / `model-0000k-of-0000n.safetensors` + `model.safetensors.index.json`,
                            tied `shared.weight` / `encoder.embed_tokens.weight`; the T5 tokenizer has 32 100 entries, the
                            checkpoint's table 32 128 rows, so the "resize" to 32 101 is a TRUNCATION (rows [:n] kept bit for
                            bit -- checked against transformers 5.15.0 in this container)
  full VAE                  `AutoencoderKLCogVideoX.from_pretrained(subfolder="vae")` (src/inference.py:201,231): decoder AND
                            encoder halves (the encoder runs the reference image, src/video_generate.py:26-38)
  LoRA scale                lora_alpha / r = 64 / 128 (src/inference.py:47-48,218-223); PEFT computes W x + (alpha/r) B A x,
                            which the merge W + (alpha/r) B A reproduces (SURVEY preamble 6)
"""
import json
import os

from safetensors import safe_open

LORA_WEIGHT_NAME_SAFE = "pytorch_lora_weights.safetensors"
_UNET_TO_PEFT = [
    (".to_out_lora.up", ".to_out.0.lora_B"), (".to_out_lora.down", ".to_out.0.lora_A"),
    (".to_q_lora.down", ".to_q.lora_A"), (".to_q_lora.up", ".to_q.lora_B"),
    (".to_k_lora.down", ".to_k.lora_A"), (".to_k_lora.up", ".to_k.lora_B"),
    (".to_v_lora.down", ".to_v.lora_A"), (".to_v_lora.up", ".to_v.lora_B"),
    (".lora.up", ".lora_B"), (".lora.down", ".lora_A"),
]


def model_files(model_dir, stem="diffusion_pytorch_model"):
    """the safetensors file(s) of one diffusers sub-model directory, in shard order"""
    single = os.path.join(model_dir, stem + ".safetensors")
    if os.path.isfile(single):
        return [single]
    index = os.path.join(model_dir, stem + ".safetensors.index.json")
    if os.path.isfile(index):
        with open(index) as f:
            weight_map = json.load(f)["weight_map"]
        return [os.path.join(model_dir, n) for n in sorted(set(weight_map.values()))]
    raise FileNotFoundError(f"no {stem}.safetensors (or sharded index) under {model_dir}")


def iter_tensors(files, device="cpu"):
    """(key, tensor) pairs, one tensor resident at a time"""
    for fn in files:
        with safe_open(fn, framework="pt", device=device) as f:
            for k in f.keys():
                yield k, f.get_tensor(k)


def find_lora_file(checkpoint_dir):
    if os.path.isfile(checkpoint_dir):
        return checkpoint_dir
    names = [f for f in os.listdir(checkpoint_dir) if f.endswith(".safetensors")]
    names = [f for f in names if all(s not in f for s in ("scheduler", "optimizer", "checkpoint"))]
    if any(f.endswith(LORA_WEIGHT_NAME_SAFE) for f in names):
        names = [f for f in names if f.endswith(LORA_WEIGHT_NAME_SAFE)]
    if len(names) == 0:
        raise FileNotFoundError(f"no LoRA .safetensors file in {checkpoint_dir}")
    if len(names) > 1:
        raise ValueError(f"Provided path contains more than one weights file in the .safetensors format: {sorted(names)}")
    return os.path.join(checkpoint_dir, names[0])


def peft_key(k):
    for old, new in _UNET_TO_PEFT:
        if old in k:
            k = k.replace(old, new)
    return k


def read_lora(checkpoint_dir, device="cpu"):
    """{base weight key: (A [r, in...], B [out, r])} for the transformer adapters of the subject-LoRA checkpoint"""
    fn = find_lora_file(checkpoint_dir)
    halves = {}
    for k, t in iter_tensors([fn], device):
        if not k.startswith("transformer."):
            continue
        k = peft_key(k[len("transformer."):])
        for tag in (".lora_A", ".lora_B"):
            if tag in k:
                base = k.replace(tag + ".default", "").replace(tag, "")  # "...to_q.lora_A.weight" -> "...to_q.weight"
                halves.setdefault(base, {})[tag] = t
    out, unexpected = {}, []
    for base, h in halves.items():
        if ".lora_A" in h and ".lora_B" in h:
            out[base] = (h[".lora_A"], h[".lora_B"])
        else:
            unexpected.append(base)
    if unexpected:
        print(f"Unexpected keys in transformer state dict: {sorted(unexpected)}")  # src/inference.py:104-105
    return out


def _config_json(model_dir):
    with open(os.path.join(model_dir, "config.json")) as f:
        return json.load(f)


def transformer_config(model_root):
    """TransformerConfig from `<root>/transformer/config.json`, `<root>/scheduler/scheduler_config.json` and
    `<root>/vae/config.json` -- the three files `from_pretrained` reads for the numbers this path needs."""
    from .config import TransformerConfig
    fields = TransformerConfig.__dataclass_fields__
    c = _config_json(os.path.join(model_root, "transformer"))
    kw = {k: v for k, v in c.items() if k in fields}
    sched = os.path.join(model_root, "scheduler", "scheduler_config.json")
    if os.path.isfile(sched):
        with open(sched) as f:
            kw["snr_shift_scale"] = float(json.load(f).get("snr_shift_scale", 3.0))
    vae = os.path.join(model_root, "vae", "config.json")
    if os.path.isfile(vae):
        kw["vae_scaling_factor"] = float(_config_json(os.path.join(model_root, "vae")).get("scaling_factor", 1.15258426))
    return TransformerConfig(**kw)


def vae_config(model_root):
    from .config import VAEConfig
    fields = VAEConfig.__dataclass_fields__
    c = _config_json(os.path.join(model_root, "vae"))
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in c.items() if k in fields}
    return VAEConfig(**kw)


def load_transformer(model, transformer_dir, lora_dir=None, lora_alpha=64, rank=128):
    """`model`: HipCogVideoXTransformer3DModel (or an S2VEngine).  Streams the shards, merges the LoRA, finalizes."""
    eng = getattr(model, "engine", model)
    for k, t in iter_tensors(model_files(transformer_dir)):
        if "pos_embedding" in k:
            continue
        eng.load_weight(k, t)
        eng._keep.clear()  # safe: the staging tensor's block is only reused by later work on the same stream
    lora = read_lora(lora_dir) if lora_dir else None
    eng.load_state_dict({}, lora=lora, lora_scale=lora_alpha / rank)
    return sorted(lora) if lora else []


def load_vae_decoder(vae, vae_dir):
    sd = {k: t for k, t in iter_tensors(model_files(vae_dir)) if k.startswith("decoder.")}
    vae.load_state_dict(sd)


def load_vae(vae, vae_dir, with_encoder=True):
    """the whole AutoencoderKLCogVideoX checkpoint (src/inference.py:201,231) into a HipAutoencoderKLCogVideoX: `decoder.*` and,
    with_encoder, `encoder.*` (215.6 M parameters in all: one state dict on the host is fine).  Returns the loaded key prefixes.
    Keys of neither half (`quant_conv.*` / `post_quant_conv.*` exist only when `use_quant_conv`, which CogVideoX does not set)
    are rejected loudly rather than dropped."""
    want = ("decoder.", "encoder.") if with_encoder else ("decoder.",)
    sd, other = {}, []
    for k, t in iter_tensors(model_files(vae_dir)):
        if k.startswith(want):
            sd[k] = t
        elif not k.startswith(("decoder.", "encoder.")):
            other.append(k)
    if other:
        raise ValueError(f"VAE checkpoint holds tensors outside encoder / decoder that this path does not evaluate: {sorted(other)[:4]}")
    if with_encoder and not any(k.startswith("encoder.") for k in sd):
        raise FileNotFoundError(f"no `encoder.*` tensors under {vae_dir}")
    vae.load_state_dict(sd)
    return sorted({k.split(".", 1)[0] for k in sd})


def t5_config(text_encoder_dir, num_tokens=None):
    """T5Config from `<dir>/config.json`; num_tokens = len(tokenizer) after the `<cls>` token was added (src/inference.py:179-189)"""
    from .t5 import T5Config
    c = _config_json(text_encoder_dir)
    kw = {k: v for k, v in c.items() if k in T5Config.__dataclass_fields__}
    if c.get("feed_forward_proj", "gated-gelu") != "gated-gelu" or c.get("is_gated_act", True) is False:
        raise ValueError(f"text encoder is not T5 v1.1 (gated-gelu): feed_forward_proj={c.get('feed_forward_proj')!r}")
    if num_tokens is not None:
        kw["vocab_size"] = int(num_tokens)
    return T5Config(**kw)


def resize_token_embeddings(table, num_tokens, new_rows=None):
    """`text_encoder.resize_token_embeddings(n)` on the checkpoint's `shared.weight`: the first min(old, n) rows are kept bit for
    bit (transformers `_get_resized_embeddings`).  For the shipped checkpoints n = 32 101 < 32 128, a truncation.  Growing the table
    makes transformers draw the new rows from a distribution fitted to the old ones -- not reproducible from the file -- so a
    caller who needs that passes the rows themselves (`new_rows` [n - old, d]) and anything else is an error."""
    import torch

    old = table.shape[0]
    if num_tokens <= old:
        return table[:num_tokens].contiguous()
    if new_rows is None or tuple(new_rows.shape) != (num_tokens - old, table.shape[1]):
        raise ValueError(f"resize_token_embeddings {old} -> {num_tokens} grows the table: pass new_rows [{num_tokens - old}, {table.shape[1]}]")
    return torch.cat([table, new_rows.to(table.dtype)], 0).contiguous()


def load_t5(text_encoder_dir, dtype=None, device="cuda:0", num_tokens=None, new_rows=None, model=None):
    """HipT5EncoderModel from a transformers-layout T5 encoder directory, the way src/inference.py:183-189,213 builds it: shards
    streamed one tensor at a time (4.7 B parameters), the tied embedding loaded once, the table resized to `num_tokens`
    (= len(tokenizer) with `<cls>`; None keeps the checkpoint's rows), decoder / lm_head tensors of a full T5 checkpoint ignored."""
    import torch

    from .t5 import HipT5EncoderModel

    files = model_files(text_encoder_dir, stem="model")
    if model is None:
        model = HipT5EncoderModel(t5_config(text_encoder_dir, num_tokens), dtype or torch.bfloat16, device)
    elif num_tokens is not None and model.cfg.vocab_size != num_tokens:
        raise ValueError(f"model built for {model.cfg.vocab_size} tokens, num_tokens={num_tokens}")

    def stream():
        seen_table = False
        for k, t in iter_tensors(files):
            if k in ("shared.weight", "encoder.embed_tokens.weight"):
                if seen_table:
                    continue  # the tied copy
                seen_table = True
                yield "shared.weight", resize_token_embeddings(t, model.cfg.vocab_size, new_rows)
            elif k.startswith("encoder."):
                yield k, t
        if not seen_table:
            raise FileNotFoundError(f"no shared.weight / encoder.embed_tokens.weight under {text_encoder_dir}")

    model.load_state_dict(_Streamed(stream))
    return model


class _Streamed:
    """a state dict that is read once, in file order, one tensor resident at a time (what load_state_dict's `.items()` loop needs)"""

    def __init__(self, gen):
        self._gen = gen

    def items(self):
        return self._gen()

    def __contains__(self, k):
        return False
