"""ORACLE -- test infrastructure only.  CPU restatement (plain torch functional ops) of the T5 v1.1 ENCODER stack that
produces the prompt embeddings: `self.text_encoder(text_input_ids)[0]` of
pipelines/cogvideo/pipeline_cogvideox.py:197-237, built by src/inference.py:183-187 with `T5EncoderModel.from_pretrained`.

The arithmetic lives in a third-party dependency that is NOT in the reference tree: `transformers` (installer.sh:5, unpinned;
the build container has 5.15.0).  Restated from its published algorithm (models/t5/modeling_t5.py: T5LayerNorm,
T5Attention with `_relative_position_bucket` / `compute_bias`, T5DenseGatedActDense with `gelu_new`, T5Block, T5Stack) and
PINNED against `transformers.T5EncoderModel` run in the build container (tests/golden/t5_tiny.npz, oracle/make_golden.py:gen_t5).
No attention mask: the pipeline passes none, so padding tokens attend and are attended like any other.

Weights: flat dict keyed by the HF state-dict names ("shared.weight", "encoder.block.0.layer.0.SelfAttention.q.weight", ...).
cfg: dict(d_model, d_kv, num_heads, d_ff, num_layers, relative_attention_num_buckets, relative_attention_max_distance,
          layer_norm_epsilon).
"""
import math

import torch
import torch.nn.functional as F


def rms_norm(x, w, eps):
    """T5LayerNorm: no mean subtraction, no bias; variance in fp32, the scaled tensor is cast to the weight dtype first."""
    var = x.float().pow(2).mean(-1, keepdim=True)
    y = x.float() * torch.rsqrt(var + eps)
    if w.dtype in (torch.float16, torch.bfloat16):
        y = y.to(w.dtype)
    return w * y


def relative_position_bucket(rel, num_buckets=32, max_distance=128):
    """bidirectional form of T5Attention._relative_position_bucket (rel = memory_position - query_position)"""
    nb = num_buckets // 2
    ret = (rel > 0).long() * nb
    rel = rel.abs()
    max_exact = nb // 2
    is_small = rel < max_exact
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return ret + torch.where(is_small, rel, large)


def position_bias(table, T, num_buckets, max_distance):
    """compute_bias: [1, H, T, T] from the block-0 embedding table [num_buckets, H]"""
    ctx = torch.arange(T)[:, None]
    mem = torch.arange(T)[None, :]
    bucket = relative_position_bucket(mem - ctx, num_buckets, max_distance)
    return table[bucket].permute(2, 0, 1).unsqueeze(0)


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def self_attention(sd, p, x, bias, H, dk):
    B, T, _ = x.shape
    q = F.linear(x, sd[p + "q.weight"]).view(B, T, H, dk).transpose(1, 2)
    k = F.linear(x, sd[p + "k.weight"]).view(B, T, H, dk).transpose(1, 2)
    v = F.linear(x, sd[p + "v.weight"]).view(B, T, H, dk).transpose(1, 2)
    scores = torch.matmul(q, k.transpose(3, 2))  # no 1/sqrt(d) in T5
    scores = scores + bias
    w = F.softmax(scores.float(), dim=-1).type_as(scores)
    o = torch.matmul(w, v).transpose(1, 2).contiguous().view(B, T, H * dk)
    return F.linear(o, sd[p + "o.weight"])


def fp16_clamp(x):
    """T5Block.forward after every sub-layer, fp16 only ("clamp inf values to enable fp16 training"): the residual stream is clamped to
    +-finfo.max, or to +-(finfo.max - 1000) when it holds an inf (T5-XXL's feed-forward outputs overflow fp16)"""
    if x.dtype != torch.float16:
        return x
    c = torch.where(torch.isinf(x).any(), torch.finfo(x.dtype).max - 1000, torch.finfo(x.dtype).max)
    return torch.clamp(x, min=-c, max=c)


def encoder_forward(sd, cfg, input_ids):
    H, dk, eps = cfg["num_heads"], cfg["d_kv"], cfg.get("layer_norm_epsilon", 1e-6)
    x = sd["shared.weight"][input_ids]
    T = input_ids.shape[1]
    bias = position_bias(sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"], T,
                         cfg["relative_attention_num_buckets"], cfg["relative_attention_max_distance"]).to(x.dtype)
    for i in range(cfg["num_layers"]):
        p = f"encoder.block.{i}."
        h = rms_norm(x, sd[p + "layer.0.layer_norm.weight"], eps)
        x = fp16_clamp(x + self_attention(sd, p + "layer.0.SelfAttention.", h, bias, H, dk))
        h = rms_norm(x, sd[p + "layer.1.layer_norm.weight"], eps)
        g = gelu_new(F.linear(h, sd[p + "layer.1.DenseReluDense.wi_0.weight"]))
        u = F.linear(h, sd[p + "layer.1.DenseReluDense.wi_1.weight"])
        x = fp16_clamp(x + F.linear(g * u, sd[p + "layer.1.DenseReluDense.wo.weight"]))
    return rms_norm(x, sd["encoder.final_layer_norm.weight"], eps)
