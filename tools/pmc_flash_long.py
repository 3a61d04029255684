#!/usr/bin/env python
"""rocprofv3 workload: the long-sequence attention kernels at their shapes, a few launches each — the 720p spatial shape (76 frames x 16
heads x 3600 tokens, head_dim 72) and the CogVideoX-5B joint shape (2 x 48 heads x 17776 tokens, head_dim 64): the 32-row kernels
(flash variant 15), the 64-rows-per-wave stream (14) and the stream without the running max (17, with the promise about the key norms)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosys_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
g = torch.Generator().manual_seed(0)
F, S, H, C = 76, 3600, 16, 1152
qkv = torch.randn(F * S, 3 * C, generator=g).to(torch.bfloat16).to(dev)
qw = (torch.randn(72, generator=g) * 0.1 + 1).to(torch.bfloat16).to(dev)
kp, vt = ops.alloc_kv_buffers(F, H, S, dev)
ops.attn_prep_kv(qkv[:, C:2 * C], qkv[:, 2 * C:], qw, kp, vt, F, H, S)
ao = torch.empty(F * S, C, dtype=torch.bfloat16, device=dev)
kb = ops.rms_key_bound(qw, qw)
for fv, bound in ((15, None), (14, None), (17, kb)):
    lib.vsys_tune_flash_variant(fv)
    for _ in range(3):
        ops.flash_attn(qkv[:, :C], qw, kp, vt, ao, F, H, S, S, k_norm_bound=bound)
del qkv, kp, vt, ao
B, L, H, D = 2, 17776, 48, 64
C = H * D
qkv = torch.randn(B * L, 3 * C, generator=g).to(torch.bfloat16).to(dev)
w = (torch.randn(D, generator=g) * 0.1 + 1).to(torch.bfloat16).to(dev)
bia = (torch.randn(D, generator=g) * 0.1).to(torch.bfloat16).to(dev)
ang = torch.rand(L - 226, D // 2, generator=g) * 6.0
cos, sin = ang.cos().repeat_interleave(2, -1).contiguous().to(dev), ang.sin().repeat_interleave(2, -1).contiguous().to(dev)
kp, vt = ops.alloc_kv_buffers64(B, H, L, dev)
ops.attn_prep_kv64(qkv[:, C:2 * C], qkv[:, 2 * C:], w, bia, cos, sin, 226, kp, vt, B, H, L)
ao = torch.empty(B * L, C, dtype=torch.bfloat16, device=dev)
kb = ops.ln_key_bound(w, bia, w, bia)
for fv, bound in ((15, None), (14, None), (17, kb)):
    lib.vsys_tune_flash_variant(fv)
    for _ in range(3):
        ops.flash_attn64(qkv[:, :C], w, bia, cos, sin, 226, kp, vt, ao, B, H, L, L, k_norm_bound=bound)
lib.vsys_tune_flash_variant(0)
torch.cuda.synchronize()
