"""TEST INFRASTRUCTURE ONLY — CPU restatement (plain PyTorch fp32) of the T5 v1.1 encoder the reference pipelines call
(`T5EncoderModel` of transformers — THIRD-PARTY, not in /root/reference; call sites pipeline_open_sora.py:211-214,269-287).

Restated from the published modeling_t5.py: T5Stack (encoder) -> T5Block -> T5LayerSelfAttention (T5LayerNorm, T5Attention with
compute_bias / _relative_position_bucket, mask added to the position bias, no score scaling) -> T5LayerFF
(T5DenseGatedActDense: wo(gelu_new(wi_0 x) * wi_1 x)) -> final_layer_norm.  PINNED: transformers IS installed in this image, so
tests/test_t5_cpu.py checks this file against the real `transformers.T5EncoderModel` (here and on the GPU box) and against
tests/golden/t5_small.pt minted from it by oracle/make_golden_t5.py.  Nothing under videosys_amd/ imports this module.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def bucket(rel, num_buckets=32, max_distance=128):
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    rel = rel.abs()
    me = nb // 2
    large = me + (torch.log(rel.float() / me) / math.log(max_distance / me) * (nb - me)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return out + torch.where(rel < me, rel, large)


def rms(x, w, eps):
    return w * (x * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + eps)).to(x.dtype)


@torch.no_grad()
def encode(sd, input_ids, attention_mask, num_layers, num_heads, eps=1e-6, num_buckets=32, max_distance=128):
    x = sd["shared.weight"][input_ids]
    B, L, D = x.shape
    pos = torch.arange(L)
    rel = pos[None, :] - pos[:, None]                                    # memory - query
    bias = sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"][bucket(rel, num_buckets, max_distance)]
    bias = bias.permute(2, 0, 1)[None]                                   # [1, H, L, L]
    if attention_mask is not None:
        bias = bias + (1.0 - attention_mask[:, None, None, :].to(x.dtype)) * torch.finfo(x.dtype).min
    for i in range(num_layers):
        p = f"encoder.block.{i}.layer."
        a = p + "0.SelfAttention."
        h = rms(x, sd[p + "0.layer_norm.weight"], eps)
        split = lambda t: t.view(B, L, num_heads, -1).transpose(1, 2)
        q, k, v = (split(F.linear(h, sd[a + n + ".weight"])) for n in "qkv")
        s = q @ k.transpose(-1, -2) + bias
        o = (torch.softmax(s.float(), -1).to(x.dtype) @ v).transpose(1, 2).reshape(B, L, -1)
        x = x + F.linear(o, sd[a + "o.weight"])
        f = p + "1.DenseReluDense."
        h = rms(x, sd[p + "1.layer_norm.weight"], eps)
        g = F.gelu(F.linear(h, sd[f + "wi_0.weight"]), approximate="tanh") * F.linear(h, sd[f + "wi_1.weight"])
        x = x + F.linear(g, sd[f + "wo.weight"])
    return rms(x, sd["encoder.final_layer_norm.weight"], eps)
