#!/usr/bin/env python
"""Lab probe of the ping-pong flash-attention kernel (flash variant 5; stamps = lab variant 6): time against the default kernel at
the config-2 spatial and cross shapes (interleaved rounds) and, in a lab build, the s_memtime cycles a wave spends per KV tile in
its matrix phase / the barrier behind it / its VALU phase / the barrier behind that, per wave group.
    VSYS_LAB=1 python tools/flash_pp_probe.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
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
kv = (torch.randn(600, 2 * C, generator=g)).to(torch.bfloat16).to(dev)
kpc, vtc = ops.alloc_kv_buffers(2, H, 300, dev)
ops.attn_prep_kv(kv[:, :C], kv[:, C:], None, kpc, vtc, 2, H, 300)


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


res = {}
variants = [int(v) for v in os.environ.get("FLASH_VARIANTS", "0,5").split(",")]
for rd in range(3):
    for fv in variants:
        assert lib.vsys_tune_flash_variant(fv) == 0
        res.setdefault(f"spatial_v{fv}", []).append(timeit(lambda: ops.flash_attn(qkv[:, :C], qw, kp, vt, ao, 38, H, 1024, 1024)))
        res.setdefault(f"cross_v{fv}", []).append(timeit(lambda: ops.flash_attn(qkv[:, :C], None, kpc, vtc, ao, 2, H, 19456, 300)))
lib.vsys_tune_flash_variant(0)
out = {k: round(min(v), 4) for k, v in res.items()}
for sv in ([6, 61, 62] if hasattr(lib, "vsys_lab_flash_debug_buffer") else []):
    if lib.vsys_tune_flash_variant(sv) != 0:
        continue
    for name, fn, ntile in (("spatial", lambda: ops.flash_attn(qkv[:, :C], qw, kp, vt, ao, 38, H, 1024, 1024), 16),
                            ("cross", lambda: ops.flash_attn(qkv[:, :C], None, kpc, vtc, ao, 2, H, 19456, 300), 5)):
        dbg = torch.zeros(8, dtype=torch.int64, device=dev)
        lib.vsys_lab_flash_debug_buffer(dbg.data_ptr())
        fn()
        torch.cuda.synchronize()
        wave_tiles = (N // 32 // 2) * H * ntile          # per wave group: rows / 32 waves, half of them, x heads x tiles
        t = dbg.cpu().tolist()
        out[f"stamps{sv}_{name}"] = {f"G{gi}": dict(zip(("matrix", "barrier_after_matrix", "valu", "barrier_after_valu"),
                                                    [round(v / wave_tiles, 1) for v in t[4 * gi:4 * gi + 4]])) for gi in (0, 1)}
lib.vsys_lab_flash_debug_buffer(None) if hasattr(lib, "vsys_lab_flash_debug_buffer") else None
lib.vsys_tune_flash_variant(0)
print(json.dumps(out, indent=1))
