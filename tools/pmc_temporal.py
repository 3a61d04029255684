#!/usr/bin/env python
"""rocprofv3 --pmc workload: the temporal-attention kernel at config-2 shape (B 2, T 19, S 1024, 16 heads), default (v3, matrix
pipe) and flash variant 4 (v2, VALU), 5 launches each.   rocprofv3 --kernel-trace --pmc ... -- python tools/pmc_temporal.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from videosys_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
C, H, T, S, B = 1152, 16, 19, 1024, 2
N = B * T * S
qkv = torch.randn(N, 3 * C, generator=g).to(torch.bfloat16).to(dev)
qw = (torch.randn(72, generator=g) * 0.1 + 1).to(torch.bfloat16).to(dev)
ao = torch.empty(N, C, dtype=torch.bfloat16, device=dev)
freqs = 1.0 / (10000 ** (torch.arange(0, 72, 2).float() / 72))
ang = torch.einsum("p,f->pf", torch.arange(T).float(), freqs).repeat_interleave(2, -1)
cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
for fv in (0, 4):
    lib.vsys_tune_flash_variant(fv)
    for _ in range(5):
        ops.attn_temporal(qkv, C, qw, qw, cos, sin, ao, B, T, S, H)
    torch.cuda.synchronize()
lib.vsys_tune_flash_variant(0)
