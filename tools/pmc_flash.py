#!/usr/bin/env python
"""rocprofv3 --pmc workload: spatial flash attention at config-2 size (38 x 16 heads, 1024 x 1024), variants 0 and 1."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videosys_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
N, C, H = 38912, 1152, 16
lib = _lib.load()
qkv = (torch.randn(N, 3 * C, generator=g)).to(torch.bfloat16).to(dev)
qw = (torch.randn(72, generator=g) + 1).to(torch.bfloat16).to(dev)
ao = torch.empty(N, C, dtype=torch.bfloat16, device=dev)
kp, vt = ops.alloc_kv_buffers(38, H, 1024, dev)
ops.attn_prep_kv(qkv[:, C:2 * C], qkv[:, 2 * C:], qw, kp, vt, 38, H, 1024)
for fv in (0, 1):
    lib.vsys_tune_flash_variant(fv)
    for _ in range(3):
        ops.flash_attn(qkv[:, :C], qw, kp, vt, ao, 38, H, 1024, 1024)
torch.cuda.synchronize()
if os.environ.get("VSYS_FLASH_STAMPS"):
    dbg = torch.zeros(5, dtype=torch.int64, device=dev)
    lib.vsys_lab_flash_debug_buffer(dbg.data_ptr())
    lib.vsys_tune_flash_variant(2)
    ops.flash_attn(qkv[:, :C], qw, kp, vt, ao, 38, H, 1024, 1024)
    torch.cuda.synchronize()
    n = 38 * H * 8 * 4 * 15  # wave-tiles stamped (last tile is peeled, not stamped)
    names = ["K reads + QK issue", "V reads + max chain", "exp + PV", "vmcnt(0)", "barrier"]
    tot = dbg.cpu().tolist()
    print("flash phase cycles per wave-tile (s_memtime ticks, 100 MHz?):")
    for nm, v in zip(names, tot):
        print(f"  {nm:22s} {v / n:10.1f}")
    print("  sum", sum(tot) / n)
    lib.vsys_tune_flash_variant(0)
    lib.vsys_lab_flash_debug_buffer(None)
