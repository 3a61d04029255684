"""T5 (v1.1 "gated-gelu") text encoder on MI355X — what every reference pipeline calls once per prompt
(`T5EncoderModel` of transformers, third-party: pipeline_open_sora.py:211-214,269-287 "DeepFloyd/t5-v1_1-xxl";
pipeline_cogvideox.py:211-247; pipeline_latte.py).  State-dict keys are the HF ones (``shared.weight``,
``encoder.block.N.layer.0.SelfAttention.{q,k,v,o,relative_attention_bias}``, ``layer.0.layer_norm``,
``layer.1.DenseReluDense.{wi_0,wi_1,wo}``, ``layer.1.layer_norm``, ``encoder.final_layer_norm``).

Per layer: RMS-norm -> fused q|k|v GEMM -> attention with the additive relative-position bias (bucketed on the host exactly
like T5Attention._relative_position_bucket, shared by all layers) and the key-padding mask -> o GEMM (+ residual in the
epilogue) -> RMS-norm -> fused wi_0|wi_1 GEMM -> gelu_new(a) * b -> wo GEMM (+ residual).  No CPU path.

A prompt is a few hundred rows against 4.7 B weights: every weight element is used for ~300 MACs, the encoder is a weight
stream.  Up to ``skinny_rows`` rows the linears therefore run as ops.linear_skinny (weight as the row operand, split over K,
fp32 slices summed and transposed back by vsys_splitk_reduce_t): every weight panel leaves HBM once and >= 256 workgroups pull
on it, instead of 64 - 320 workgroups that each re-stream the activations.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, Optional

import torch

from . import ops


def relative_position_bucket(rel: torch.Tensor, num_buckets: int = 32, max_distance: int = 128) -> torch.Tensor:
    """T5Attention._relative_position_bucket, bidirectional (transformers modeling_t5.py), on int64 relative positions
    (memory - query)."""
    nb = num_buckets // 2
    out = (rel > 0).to(torch.long) * nb
    rel = rel.abs()
    max_exact = nb // 2
    is_small = rel < max_exact
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, nb - 1))
    return out + torch.where(is_small, rel, large)


def relative_bias_table(rel_weight: torch.Tensor, L: int, num_buckets: int = 32, max_distance: int = 128) -> torch.Tensor:
    """[heads, 2L-1] fp32: the bias of relative position d = j - i (memory - query) at column d + L - 1, i.e.
    T5Attention.compute_bias(L, L)[0, h, i, j] == table[h, j - i + L - 1].  rel_weight: the [num_buckets, heads] embedding."""
    rel = torch.arange(-(L - 1), L, dtype=torch.long)
    bucket = relative_position_bucket(rel, num_buckets, max_distance)
    return rel_weight.float()[bucket].t().contiguous()


def padded_bias_table(rel_weight: torch.Tensor, L: int, num_buckets: int = 32, max_distance: int = 128):
    """(table fp32 [heads, ld], center) for ops.t5_attention_mfma: log2(e) * bias of relative position d = key - query at column
    center + d, zero outside |d| <= L - 1, wide enough for every (key < 64 ceil(L/64), query < 128 ceil(L/128)) the kernel's
    tiles touch (those positions are masked or discarded, the reads must only stay inside the table)."""
    center = (L + 127) // 128 * 128 - 1
    ld = center + (L + 63) // 64 * 64
    t = torch.zeros(rel_weight.shape[1], ld, dtype=torch.float32)
    t[:, center - (L - 1):center + L] = relative_bias_table(rel_weight, L, num_buckets, max_distance) * math.log2(math.e)
    return t.contiguous(), center


class T5Encoder:
    def __init__(self, d_model=4096, d_kv=64, d_ff=10240, num_layers=24, num_heads=64, vocab_size=32128,
                 relative_attention_num_buckets=32, relative_attention_max_distance=128, layer_norm_epsilon=1e-6, device="cuda"):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("videosys_amd.T5Encoder needs a HIP device (no CPU path)")
        if d_kv != 64:
            raise NotImplementedError("the attention kernel is built for d_kv = 64 (every T5 v1.1 size)")
        if d_model % 128 or d_ff % 128 or (num_heads * d_kv) % 128:
            raise NotImplementedError("d_model, d_ff and the attention inner dim must be multiples of 128")
        self.config = SimpleNamespace(d_model=d_model, d_kv=d_kv, d_ff=d_ff, num_layers=num_layers, num_heads=num_heads,
                                      vocab_size=vocab_size, relative_attention_num_buckets=relative_attention_num_buckets,
                                      relative_attention_max_distance=relative_attention_max_distance,
                                      layer_norm_epsilon=layer_norm_epsilon)
        self.device, self.dtype = dev, torch.bfloat16
        self.w: Dict[str, torch.Tensor] = {}
        self._bias_cache: Dict[int, torch.Tensor] = {}
        self.skinny_rows = 1024            # B * L up to which the linears take the weight-streaming path
        self.mfma_attention = True         # False: the one-wave-per-query-row VALU kernel (kept for A/B and as the checker's twin)
        self._ws: Dict[tuple, torch.Tensor] = {}

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        c = self.config
        dev = lambda t: t.detach().to(device=self.device, dtype=self.dtype).contiguous()
        emb = sd["shared.weight"] if "shared.weight" in sd else sd["encoder.embed_tokens.weight"]
        self.w["emb"] = dev(emb)
        for i in range(c.num_layers):
            p = f"encoder.block.{i}.layer."
            a = p + "0.SelfAttention."
            self.w[f"{i}.qkv"] = dev(torch.cat([sd[a + "q.weight"], sd[a + "k.weight"], sd[a + "v.weight"]], 0))
            self.w[f"{i}.o"] = dev(sd[a + "o.weight"])
            self.w[f"{i}.ln0"] = dev(sd[p + "0.layer_norm.weight"])
            f = p + "1.DenseReluDense."
            self.w[f"{i}.wi"] = dev(torch.cat([sd[f + "wi_0.weight"], sd[f + "wi_1.weight"]], 0))
            self.w[f"{i}.wo"] = dev(sd[f + "wo.weight"])
            self.w[f"{i}.ln1"] = dev(sd[p + "1.layer_norm.weight"])
        self.w["ln_f"] = dev(sd["encoder.final_layer_norm.weight"])
        # [num_buckets, heads] fp32 (the embedding is looked up in the model dtype, then added to the scores)
        self._rel = sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"].detach().to(self.dtype).float().cpu()
        self._bias_cache.clear()
        return self

    def init_random_(self, seed: int = 0):
        """Random weights generated ON the device (benchmarks at the XXL size: 4.8 B parameters are slow to draw on the host)."""
        c, dev = self.config, self.device
        g = torch.Generator(device=dev).manual_seed(seed)
        r = lambda *s, scale=1.0: (torch.randn(*s, generator=g, device=dev) * scale).to(self.dtype)
        inner = c.num_heads * c.d_kv
        ones = lambda n: torch.ones(n, dtype=self.dtype, device=dev)
        self.w["emb"] = r(c.vocab_size, c.d_model)
        for i in range(c.num_layers):
            self.w[f"{i}.qkv"] = torch.cat([r(inner, c.d_model, scale=(c.d_model * c.d_kv) ** -0.5),
                                            r(2 * inner, c.d_model, scale=c.d_model ** -0.5)], 0)
            self.w[f"{i}.o"] = r(c.d_model, inner, scale=inner ** -0.5)
            self.w[f"{i}.wi"] = r(2 * c.d_ff, c.d_model, scale=c.d_model ** -0.5)
            self.w[f"{i}.wo"] = r(c.d_model, c.d_ff, scale=c.d_ff ** -0.5)
            self.w[f"{i}.ln0"], self.w[f"{i}.ln1"] = ones(c.d_model), ones(c.d_model)
        self.w["ln_f"] = ones(c.d_model)
        self._rel = (torch.randn(c.relative_attention_num_buckets, c.num_heads, generator=torch.Generator().manual_seed(seed)) * 0.5)
        self._bias_cache.clear()
        return self

    def _relbias(self, L: int) -> torch.Tensor:
        t = self._bias_cache.get(L)
        if t is None:
            c = self.config
            t = relative_bias_table(self._rel, L, c.relative_attention_num_buckets, c.relative_attention_max_distance).to(self.device)
            self._bias_cache[L] = t
        return t

    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None):
        """-> SimpleNamespace(last_hidden_state [B, L, d_model] bf16).  attention_mask: [B, L], a prefix of ones per sample
        (what tokenizer(padding="max_length") produces)."""
        c, w = self.config, self.w
        B, L = input_ids.shape
        if attention_mask is None:
            lens = [L] * B
        else:
            m = attention_mask.reshape(B, L).to("cpu") != 0
            lens = [int(v) for v in m.sum(1).tolist()]
            for b in range(B):
                if not bool(m[b, :lens[b]].all()):
                    raise NotImplementedError("attention_mask must be a prefix of ones per sample")
        klen = torch.tensor(lens, dtype=torch.int32, device=self.device)
        ids = input_ids.reshape(-1).to(device=self.device, dtype=torch.int64).contiguous()
        bias = self._relbias(L)
        self._lens = lens
        if B * L <= self.skinny_rows:
            return self._forward_skinny(ids, bias, klen, B, L)
        x = ops.gather_rows(w["emb"], ids)                                  # [B*L, d_model]
        h = torch.empty_like(x)
        for i in range(c.num_layers):
            ops.rms_norm_rows(x, w[f"{i}.ln0"], c.layer_norm_epsilon, out=h)
            qkv = ops.gemm128(h, w[f"{i}.qkv"])
            ao = self._attention(qkv, bias, klen, B, L)
            x = ops.gemm128(ao, w[f"{i}.o"], res=x)
            ops.rms_norm_rows(x, w[f"{i}.ln1"], c.layer_norm_epsilon, out=h)
            g = ops.geglu(ops.gemm128(h, w[f"{i}.wi"]))
            x = ops.gemm128(g, w[f"{i}.wo"], res=x)
        out = ops.rms_norm_rows(x, w["ln_f"], c.layer_norm_epsilon)
        return SimpleNamespace(last_hidden_state=out.view(B, L, c.d_model))

    def _attention(self, qkv, bias, klen, B, L, out=None):
        c = self.config
        if not self.mfma_attention:
            return ops.t5_attention(qkv, bias, klen, B, L, c.num_heads, out=out)
        key = ("pad", L)
        if key not in self._bias_cache:
            t, center = padded_bias_table(self._rel, L, c.relative_attention_num_buckets, c.relative_attention_max_distance)
            self._bias_cache[key] = (t.to(self.device), center)
        t, center = self._bias_cache[key]
        kv_pad = (L + 63) // 64 * 64
        ws = (self._buf("t5_kp", (c.num_heads * kv_pad * 64,)), self._buf("t5_vt", (c.num_heads * kv_pad * 64,)))
        return ops.t5_attention_mfma(qkv, t, center, self._lens, B, L, c.num_heads, out=out, ws=ws)

    def _buf(self, name, shape, dtype=torch.bfloat16):
        t = self._ws.get((name, shape, dtype))
        if t is None:
            t = torch.zeros(shape, dtype=dtype, device=self.device)
            self._ws[(name, shape, dtype)] = t
        return t

    def _forward_skinny(self, ids, bias, klen, B, L):
        """The same layer sequence on row-padded buffers ([Mp, .], Mp = rows rounded up to the 384-column tile; rows >= M are
        never read into a result) with the weight-streaming linears."""
        c, w = self.config, self.w
        M = B * L
        Mp = (M + 383) // 384 * 384        # the 256 x 384 tile of ops.linear_skinny
        inner = c.num_heads * c.d_kv
        x = self._buf("x", (Mp, c.d_model))
        h = self._buf("h", (Mp, c.d_model))
        qkv = self._buf("qkv", (Mp, 3 * inner))
        ao = self._buf("ao", (Mp, inner))
        hw = self._buf("wi", (Mp, 2 * c.d_ff))
        g = self._buf("g", (Mp, c.d_ff))
        shapes = ((3 * inner, c.d_model), (c.d_model, inner), (2 * c.d_ff, c.d_model), (c.d_model, c.d_ff))
        part = self._buf("part", (max(ops.skinny_split(n, Mp, k) * n * Mp for n, k in shapes),), torch.float32)
        x[:M].copy_(ops.gather_rows(w["emb"], ids))
        for i in range(c.num_layers):
            ops.rms_norm_rows(x[:M], w[f"{i}.ln0"], c.layer_norm_epsilon, out=h[:M])
            ops.linear_skinny(h, M, w[f"{i}.qkv"], out=qkv, part=part)
            self._attention(qkv[:M], bias, klen, B, L, out=ao[:M])
            ops.linear_skinny(ao, M, w[f"{i}.o"], res=x, out=x, part=part)
            ops.rms_norm_rows(x[:M], w[f"{i}.ln1"], c.layer_norm_epsilon, out=h[:M])
            ops.linear_skinny(h, M, w[f"{i}.wi"], out=hw, part=part)
            ops.geglu(hw[:M], out=g[:M])
            ops.linear_skinny(g, M, w[f"{i}.wo"], res=x, out=x, part=part)
        out = ops.rms_norm_rows(x[:M], w["ln_f"], c.layer_norm_epsilon)
        return SimpleNamespace(last_hidden_state=out.view(B, L, c.d_model))

    __call__ = forward


class T5TextEncoder:
    """The callable the pipelines take as ``text_encoder``: prompt(s) -> (embeddings [B, 1, L, d_model], mask [B, L])
    (pipeline_open_sora.py:269-292 get_text_embeddings / encode_prompt).  ``tokenizer`` is any HF-style tokenizer callable
    (the sentencepiece model of "DeepFloyd/t5-v1_1-xxl" cannot be fetched here).  ``use_attention_mask=False``: the encoder attends
    the padding too — how CogVideoX calls it (pipelines/cogvideox/pipeline_cogvideox.py:244, ``text_encoder(input_ids)`` alone)."""

    def __init__(self, encoder: T5Encoder, tokenizer, max_length: int = 300, use_attention_mask: bool = True):
        self.encoder, self.tokenizer, self.max_length, self.use_attention_mask = encoder, tokenizer, max_length, use_attention_mask

    def __call__(self, prompts):
        if isinstance(prompts, str):
            prompts = [prompts]
        tok = self.tokenizer(prompts, max_length=self.max_length, padding="max_length", truncation=True, return_attention_mask=True,
                             add_special_tokens=True, return_tensors="pt")
        emb = self.encoder(tok["input_ids"], tok["attention_mask"] if self.use_attention_mask else None).last_hidden_state
        return emb[:, None], tok["attention_mask"]


class ByteTokenizer:
    """Offline stand-in for the HF tokenizer of ``text_encoder="synthetic:<seed>"`` pipelines (the sentencepiece model of
    DeepFloyd/t5-v1_1-xxl cannot be fetched here): UTF-8 bytes + 3 as token ids, id 1 (``</s>``) appended, id 0 padding — the
    call signature and the returned ``input_ids`` / ``attention_mask`` tensors of ``tokenizer(..., return_tensors="pt")``."""

    def __init__(self, vocab_size: int = 32128):
        self.vocab_size = vocab_size

    def __call__(self, prompts, max_length=300, padding="max_length", truncation=True, return_attention_mask=True,
                 add_special_tokens=True, return_tensors="pt"):
        if isinstance(prompts, str):
            prompts = [prompts]
        ids = torch.zeros(len(prompts), max_length, dtype=torch.int64)
        mask = torch.zeros(len(prompts), max_length, dtype=torch.int64)
        for b, text in enumerate(prompts):
            toks = [(v + 3) % self.vocab_size for v in text.encode("utf-8")][: max_length - 1] + [1]
            ids[b, : len(toks)] = torch.tensor(toks)
            mask[b, : len(toks)] = 1
        return {"input_ids": ids, "attention_mask": mask}


def synth_state_dict(d_model=4096, d_ff=10240, num_layers=24, num_heads=64, vocab_size=32128, num_buckets=32, seed: int = 5):
    """Seeded random weights with the HF T5EncoderModel key names (bf16-representable fp32)."""
    g = torch.Generator().manual_seed(seed)
    inner = num_heads * 64
    r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(torch.bfloat16).float()
    sd = {"shared.weight": r(vocab_size, d_model)}
    for i in range(num_layers):
        p = f"encoder.block.{i}.layer."
        a = p + "0.SelfAttention."
        sd[a + "q.weight"] = r(inner, d_model, scale=(d_model * 64) ** -0.5)
        sd[a + "k.weight"] = r(inner, d_model, scale=d_model ** -0.5)
        sd[a + "v.weight"] = r(inner, d_model, scale=d_model ** -0.5)
        sd[a + "o.weight"] = r(d_model, inner, scale=inner ** -0.5)
        if i == 0:
            sd[a + "relative_attention_bias.weight"] = r(num_buckets, num_heads, scale=0.5)
        sd[p + "0.layer_norm.weight"] = 1 + r(d_model, scale=0.1)
        f = p + "1.DenseReluDense."
        sd[f + "wi_0.weight"] = r(d_ff, d_model, scale=d_model ** -0.5)
        sd[f + "wi_1.weight"] = r(d_ff, d_model, scale=d_model ** -0.5)
        sd[f + "wo.weight"] = r(d_model, d_ff, scale=d_ff ** -0.5)
        sd[p + "1.layer_norm.weight"] = 1 + r(d_model, scale=0.1)
    sd["encoder.final_layer_norm.weight"] = 1 + r(d_model, scale=0.1)
    return sd
