"""Spatial flash attention (d = 72, q = kv = S tokens per frame) at an arbitrary shape, per vsys_tune_flash_variant id: median ms and
algorithmic TFLOP/s (4 frames heads S^2 72).  Every variant's output is compared with variant 15 (the 32-row kernel).
    python tools/flash_shape_probe.py --frames 76 --tokens 3600 --variants 15,141,15,141"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosys_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=76)
    ap.add_argument("--tokens", type=int, default=3600)
    ap.add_argument("--variants", default="15,141,15,141")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--d64", action="store_true", help="head_dim 64 joint attention (CogVideoX-5B: --frames 2 --tokens 17776 --heads 48)")
    ap.add_argument("--heads", type=int, default=16)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    if a.d64:
        return d64(a, dev, lib, g)
    F, S, H, C = a.frames, a.tokens, 16, 1152
    qkv = torch.randn(F * S, 3 * C, generator=g).to(torch.bfloat16).to(dev)
    qw = (torch.randn(72, generator=g) * 0.1 + 1).to(torch.bfloat16).to(dev)
    kp, vt = ops.alloc_kv_buffers(F, H, S, dev)
    ops.attn_prep_kv(qkv[:, C:2 * C], qkv[:, 2 * C:], qw, kp, vt, F, H, S)
    ao = torch.empty(F * S, C, dtype=torch.bfloat16, device=dev)
    kb = ops.rms_key_bound(qw, qw)   # variants 17 / 18 / 0 run without the running max when the promise is passed (ids >= 1000: id - 1000 with it)
    flops = 4.0 * F * H * S * S * 72
    ref, out = None, []
    for v in [int(x) for x in a.variants.split(",")]:
        bound = kb if (v in (17, 18) or v >= 1000) else None
        lib.vsys_tune_flash_variant(v % 1000)
        for _ in range(2):
            ops.flash_attn(qkv[:, :C], qw, kp, vt, ao, F, H, S, S, k_norm_bound=bound)
        ts = []
        for _ in range(a.reps):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.flash_attn(qkv[:, :C], qw, kp, vt, ao, F, H, S, S, k_norm_bound=bound)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
        ts.sort()
        if ref is None:
            ref = ao.float().clone()
        err = (ao.float() - ref).abs().max().item()
        out.append({"variant": v, "ms_med": round(ts[len(ts) // 2], 4), "ms_min": round(ts[0], 4),
                    "tflops_med": round(flops / (ts[len(ts) // 2] * 1e-3) / 1e12, 1), "max_abs_diff_vs_first": err})
    lib.vsys_tune_flash_variant(0)
    print(json.dumps({"frames": F, "tokens": S, "heads": H, "results": out}, indent=1))


def d64(a, dev, lib, g):
    B, L, H, D = a.frames, a.tokens, a.heads, 64
    C = H * D
    qkv = torch.randn(B * L, 3 * C, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(D, generator=g) * 0.1 + 1).to(torch.bfloat16).to(dev)
    bia = (torch.randn(D, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    Lt = 226 if L > 226 else 0
    ang = torch.rand(L - Lt, D // 2, generator=g) * 6.0
    cos, sin = ang.cos().repeat_interleave(2, -1).contiguous().to(dev), ang.sin().repeat_interleave(2, -1).contiguous().to(dev)
    kp, vt = ops.alloc_kv_buffers64(B, H, L, dev)
    ops.attn_prep_kv64(qkv[:, C:2 * C], qkv[:, 2 * C:], w, bia, cos, sin, Lt, kp, vt, B, H, L)
    ao = torch.empty(B * L, C, dtype=torch.bfloat16, device=dev)
    kb = ops.ln_key_bound(w, bia, w, bia)   # ids >= 1000: id - 1000 with the promise (1017 = the stream without the running max)
    flops = 4.0 * B * H * L * L * D
    ref, out = None, []
    for v in [int(x) for x in a.variants.split(",")]:
        bound = kb if v >= 1000 else None
        lib.vsys_tune_flash_variant(v % 1000)
        for _ in range(2):
            ops.flash_attn64(qkv[:, :C], w, bia, cos, sin, Lt, kp, vt, ao, B, H, L, L, k_norm_bound=bound)
        ts = []
        for _ in range(a.reps):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.flash_attn64(qkv[:, :C], w, bia, cos, sin, Lt, kp, vt, ao, B, H, L, L, k_norm_bound=bound)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
        ts.sort()
        if ref is None:
            ref = ao.float().clone()
        out.append({"variant": v, "ms_med": round(ts[len(ts) // 2], 4), "ms_min": round(ts[0], 4),
                    "tflops_med": round(flops / (ts[len(ts) // 2] * 1e-3) / 1e12, 1),
                    "max_abs_diff_vs_first": (ao.float() - ref).abs().max().item()})
    lib.vsys_tune_flash_variant(0)
    print(json.dumps({"head_dim": 64, "batch": B, "tokens": L, "heads": H, "results": out}, indent=1))


if __name__ == "__main__":
    main()
