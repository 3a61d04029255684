#!/usr/bin/env python
"""Tiny workload for rocprofv3 --pmc passes: each hot kernel at config-2 shape, a few launches each."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videosys_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
N, C, H = 38912, 1152, 16
variant = int(os.environ.get("VSYS_GEMM_VARIANT", "0"))
_lib.load().vsys_tune_gemm_variant(variant)


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(dev)


x, h = rnd(N, C), rnd(N, 4 * C)
mod = rnd(2, 6 * C, scale=0.3)
resid = rnd(N, C)
for name, n, k, epi in (("qkv", 3 * C, C, 0), ("proj", C, C, 2), ("fc1", 4 * C, C, 1), ("fc2", C, 4 * C, 2)):
    w, b = rnd(n, k, scale=1 / math.sqrt(k)), rnd(n, scale=0.1)
    out = torch.empty(N, n, dtype=torch.bfloat16, device=dev)
    a = h if k == 4 * C else x
    for _ in range(3):
        if epi == 2:
            ops.gemm(a, w, b, epilogue=epi, gate=mod[0, 2 * C:3 * C], gate_stride=6 * C, rows_per_sample=N // 2, res=resid, out=out)
        else:
            ops.gemm(a, w, b, epilogue=epi, out=out)
qkv = rnd(N, 3 * C)
qw = rnd(72) + 1
ao = torch.empty(N, C, dtype=torch.bfloat16, device=dev)
kp, vt = ops.alloc_kv_buffers(38, H, 1024, dev)
for _ in range(3):
    ops.attn_prep_kv(qkv[:, C:2 * C], qkv[:, 2 * C:], qw, kp, vt, 38, H, 1024)
    ops.flash_attn(qkv[:, :C], qw, kp, vt, ao, 38, H, 1024, 1024)
_lib.load().vsys_tune_flash_variant(16)   # the persistent 64-rows-per-wave instruction stream (attention_w64.hip), opt-in in the product
for _ in range(3):
    ops.flash_attn(qkv[:, :C], qw, kp, vt, ao, 38, H, 1024, 1024)
_lib.load().vsys_tune_flash_variant(0)
# the AdaLN-folded forms of qkv / fc1 (LayerNorm + modulation in the epilogue) and the statistics-emitting gate + residual GEMM
st = ops.ln_stats_buffer(N, C, dev)
ops.ln_row_stats(x, st)
for name, n, gelu in (("qkv_ln", 3 * C, False), ("fc1_ln", 4 * C, True)):
    wp = rnd(n, C, scale=1 / math.sqrt(C))
    cs, cv = wp.float().sum(1).contiguous(), torch.zeros(n, dtype=torch.float32, device=dev)
    out = torch.empty(N, n, dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        ops.gemm_ln(x, wp, cs, cv, st, gelu=gelu, out=out)
for k_, a_ in ((C, x), (4 * C, h)):
    w, b = rnd(C, k_, scale=1 / math.sqrt(k_)), rnd(C, scale=0.1)
    out = torch.empty(N, C, dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        ops.gemm_stats(a_, w, b, st, gate=mod[0, 2 * C:3 * C], gate_stride=6 * C, rows_per_sample=N // 2, res=resid, out=out)
kv = rnd(600, 2 * C)   # cross attention: 300 text keys per sample (resident-K/V kernel)
kpc, vtc = ops.alloc_kv_buffers(2, H, 300, dev)
ops.attn_prep_kv(kv[:, :C], kv[:, C:], None, kpc, vtc, 2, H, 300)
for _ in range(3):
    ops.flash_attn(x, None, kpc, vtc, ao, 2, H, 19456, 300)
freqs = 1.0 / (10000 ** (torch.arange(0, 72, 2).float() / 72))
ang = torch.einsum("p,f->pf", torch.arange(19).float(), freqs).repeat_interleave(2, -1)
cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
for _ in range(3):
    ops.attn_temporal(qkv, C, qw, qw, cos, sin, ao, 2, 19, 1024, H)
    ops.adaln_modulate(x, mod[0, :C], mod[0, C:2 * C], N // 2, 6 * C, out=ao)
torch.cuda.synchronize()
