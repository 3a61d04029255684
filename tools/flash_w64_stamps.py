#!/usr/bin/env python
"""Where a workgroup of flash_attn_d72_w64 spends its life (lab variant 147: s_memtime at kernel entry, asm start, loop head, asm
end, after the O read-out, after the stores) at the config-2 spatial shape.  Needs the lab library:
    VSYS_LIB=videosys_amd/libvideosys_amd_lab.so python tools/flash_w64_stamps.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from videosys_amd import _lib, ops

    lib = _lib.load()
    assert hasattr(lib, "vsys_lab_flash_debug_buffer"), "lab library needed (VSYS_LIB=...)"
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    N, C, H = 38912, 1152, 16
    qkv = torch.randn(N, 3 * C, generator=g).to(torch.bfloat16).to(dev)
    qw = (torch.randn(72, generator=g) * 0.1 + 1).to(torch.bfloat16).to(dev)
    ao = torch.empty(N, C, dtype=torch.bfloat16, device=dev)
    kp, vt = ops.alloc_kv_buffers(38, H, 1024, dev)
    ops.attn_prep_kv(qkv[:, C:2 * C], qkv[:, 2 * C:], qw, kp, vt, 38, H, 1024)
    nblk = 38 * H * 4
    dbg = torch.zeros(nblk * 4 * 8, dtype=torch.int64, device=dev)
    lib.vsys_lab_flash_debug_buffer(dbg.data_ptr())
    assert lib.vsys_tune_flash_variant(147) == 0
    for _ in range(3):
        ops.flash_attn(qkv[:, :C], qw, kp, vt, ao, 38, H, 1024, 1024)
    torch.cuda.synchronize()
    lib.vsys_tune_flash_variant(0)
    d = dbg.view(nblk, 4, 8).cpu().double()
    seg = {"c++ prologue (descriptors, Vt zero rows, Q fetch + norm)": d[..., 1] - d[..., 0],
           "asm prologue (O / -m init, 20 LDS-DMA pieces, K(0), S(0), adopt)": d[..., 2] - d[..., 1],
           "tile loop (16 tiles)": d[..., 3] - d[..., 2],
           "O read-out": d[..., 4] - d[..., 3],
           "normalise + store + drain": d[..., 5] - d[..., 4],
           "whole wave": d[..., 5] - d[..., 0]}
    out = {k: {"mean_cycles": round(float(v.mean()), 1), "p10": round(float(v.flatten().kthvalue(max(1, v.numel() // 10)).values), 1),
               "p90": round(float(v.flatten().kthvalue(v.numel() * 9 // 10).values), 1)} for k, v in seg.items()}
    out["per_tile_cycles"] = round(out["tile loop (16 tiles)"]["mean_cycles"] / 16, 1)
    out["kernel_span_cycles"] = float(d[..., 5].max() - d[..., 0].min())
    print(json.dumps(out, indent=1))
    # ---- the persistent form (lab variant 146): per wave sums over the items it walked
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    dbg2 = torch.zeros(ncu * 4 * 8, dtype=torch.int64, device=dev)
    lib.vsys_lab_flash_debug_buffer(dbg2.data_ptr())
    assert lib.vsys_tune_flash_variant(146) == 0
    ops.flash_attn(qkv[:, :C], qw, kp, vt, ao, 38, H, 1024, 1024)
    torch.cuda.synchronize()
    dbg2.zero_()
    ops.flash_attn(qkv[:, :C], qw, kp, vt, ao, 38, H, 1024, 1024)
    torch.cuda.synchronize()
    lib.vsys_tune_flash_variant(0)
    e = dbg2.view(ncu, 4, 8).cpu().double()
    n = e[..., 6].clamp_min(1)
    names = ["wait for the Q image", "Q fragments (LDS reads, norm, accumulator writes, descriptors)",
             "opening wait + barrier of the statement", "O / -m init, K(0), S(0), adopt", "tile loop", "O read-out, normalise, stores issued"]
    out2 = {nm: round(float((e[..., i] / n).mean()), 1) for i, nm in enumerate(names)}
    out2["cycles per item"] = round(sum(out2.values()), 1)
    out2["items per wave (min, max)"] = [float(e[..., 6].min()), float(e[..., 6].max())]
    print(json.dumps({"persistent form, mean cycles per item": out2}, indent=1))


if __name__ == "__main__":
    main()
