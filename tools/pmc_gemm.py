#!/usr/bin/env python
"""rocprofv3 --pmc workload for the GEMM only: qkv and fc2 shapes, the schedule variants named in VSYS_GEMM_VARIANTS."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videosys_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
N, C = 38912, 1152
lib = _lib.load()


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(dev)


x, h = rnd(N, C), rnd(N, 4 * C)
mod = rnd(2, 6 * C, scale=0.3)
resid = rnd(N, C)
shapes = [("qkv", 3 * C, C, 0), ("fc2", C, 4 * C, 2)]
if os.environ.get("VSYS_GEMM_ALL_SHAPES"):  # dispatch order = qkv, proj, fc1, fc2 (3 launches each) per variant
    shapes = [("qkv", 3 * C, C, 0), ("proj", C, C, 2), ("fc1", 4 * C, C, 1), ("fc2", C, 4 * C, 2)]
bufs = {name: (rnd(n, k, scale=1 / math.sqrt(k)), rnd(n, scale=0.1), torch.empty(N, n, dtype=torch.bfloat16, device=dev))
        for name, n, k, epi in shapes}
for variant in [int(v) for v in os.environ.get("VSYS_GEMM_VARIANTS", "6,8").split(",")]:
    lib.vsys_tune_gemm_variant(variant)
    for name, n, k, epi in shapes:
        w, b, out = bufs[name]
        a = h if k == 4 * C else x
        for _ in range(3):
            if epi == 2:
                ops.gemm(a, w, b, epilogue=epi, gate=mod[0, 2 * C:3 * C], gate_stride=6 * C, rows_per_sample=N // 2, res=resid, out=out)
            else:
                ops.gemm(a, w, b, epilogue=epi, out=out)
torch.cuda.synchronize()
