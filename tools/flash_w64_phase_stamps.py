#!/usr/bin/env python
"""Per-phase cycles of the w64 flash tile loop (lab variant 150: s_memtime at the head of X, in front of the wait + barrier, behind
the barrier and at the end of Y of every loop tile, summed per wave): where a tile's ~1700 cycles go against its 44 MFMAs
(X: 20 = 640 cycles, Y: 24 = 768).  The stamps themselves cost a few percent.  Needs the lab library:
    VSYS_LIB=videosys_amd/libvideosys_amd_lab.so python tools/flash_w64_phase_stamps.py [tokens_per_frame] [frames]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from videosys_amd import _lib, ops

    lib = _lib.load()
    assert hasattr(lib, "vsys_lab_flash_debug_buffer"), "lab library needed (VSYS_LIB=...)"
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 3600
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    C, H = 1152, 16
    qkv = torch.randn(F * S, 3 * C, generator=g).to(torch.bfloat16).to(dev)
    qw = (torch.randn(72, generator=g) * 0.1 + 1).to(torch.bfloat16).to(dev)
    ao = torch.empty(F * S, C, dtype=torch.bfloat16, device=dev)
    kp, vt = ops.alloc_kv_buffers(F, H, S, dev)
    ops.attn_prep_kv(qkv[:, C:2 * C], qkv[:, 2 * C:], qw, kp, vt, F, H, S)
    nblk = F * H * ((S + 255) // 256)
    dbg = torch.zeros(nblk * 4 * 8, dtype=torch.int64, device=dev)
    lib.vsys_lab_flash_debug_buffer(dbg.data_ptr())
    assert lib.vsys_tune_flash_variant(150) == 0
    for _ in range(2):
        ops.flash_attn(qkv[:, :C], qw, kp, vt, ao, F, H, S, S)
    torch.cuda.synchronize()
    lib.vsys_tune_flash_variant(0)
    d = dbg.view(nblk, 4, 8).cpu().double()
    ntl = (S + 63) // 64 - 1          # tiles that run through body() (the last one is final())
    out = {"tokens": S, "frames": F, "stamped tiles per wave": ntl,
           "X: 20 QK^T MFMAs beside exp / cvt / Vt reads (640 MFMA cycles)": round(float(d[..., 1].mean()) / ntl, 1),
           "wait (vmcnt, lgkmcnt) + s_barrier": round(float(d[..., 2].mean()) / ntl, 1),
           "Y: 24 PV MFMAs beside exp / max / K reads / LDS-DMA (768 MFMA cycles)": round(float(d[..., 3].mean()) / ntl, 1)}
    out["sum per tile"] = round(sum(v for k, v in out.items() if k[0] in "XwY"), 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
