#!/usr/bin/env python
"""Per-kernel timing at BASELINE config-2 shapes (HIP events on the launch stream, interleaved A/B of variants).
Run on the GPU box:  python tools/kernel_bench.py [--reps 20]  -> table on stdout + gpurun_out/kernel_bench.json"""
import argparse
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps  # ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--variants", default="0", help="comma list of GEMM pipeline variants to A/B (interleaved); 0 = default")
    ap.add_argument("--flash-variants", default="0")
    ap.add_argument("--only", default="", help="'gemm' = skip the non-GEMM kernels")
    ap.add_argument("--rows", type=int, default=38912, help="token rows (38912 = config 2; 4864 = one rank of 8-way DSP)")
    ap.add_argument("--vendor", action="store_true", help="also time torch.nn.functional.linear (hipBLASLt / rocBLAS) on the four "
                    "GEMM shapes: a yardstick only, never on the product path")
    args = ap.parse_args()
    if args.rows != 38912 and args.only != "gemm":
        print(f"[kernel_bench] --rows {args.rows}: the attention / row-wise section is sized for the 38912 rows of config 2; GEMMs only",
              flush=True)
        args.only = "gemm"
    import __graft_entry__ as ge

    ge.build()
    from videosys_amd import _lib, ops

    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    N, C, H = args.rows, 1152, 16
    res = {}

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(dev)

    x = rnd(N, C)
    h = rnd(N, 4 * C)
    mod = rnd(2, 6 * C, scale=0.3)
    shapes = [("qkv", 3 * C, C, ops.EPI_BIAS), ("proj", C, C, ops.EPI_GATE_RES), ("fc1", 4 * C, C, ops.EPI_BIAS_GELU),
              ("fc2", C, 4 * C, ops.EPI_GATE_RES), ("crossq", C, C, ops.EPI_BIAS)]
    if args.only == "flash":
        shapes = []
    bufs = {}
    for name, n, k, epi in shapes:
        bufs[name] = (rnd(n, k, scale=1 / math.sqrt(k)), rnd(n, scale=0.1), torch.empty(N, n, dtype=torch.bfloat16, device=dev))
    resid = rnd(N, C)
    for rd in range(args.rounds):
        for variant in [int(v) for v in args.variants.split(',')]:
            assert lib.vsys_tune_gemm_variant(variant) == 0, f"GEMM variant {variant} is not in this build"
            for name, n, k, epi in shapes:
                w, b, out = bufs[name]
                a = h if k == 4 * C else x
                if epi == ops.EPI_GATE_RES:
                    fn = lambda: ops.gemm(a, w, b, epilogue=epi, gate=mod[0, 2 * C:3 * C], gate_stride=6 * C,
                                          rows_per_sample=N // 2, res=resid, out=out)
                else:
                    fn = lambda: ops.gemm(a, w, b, epilogue=epi, out=out)
                ms = timeit(fn, args.reps)
                tf = 2.0 * N * n * k / (ms * 1e-3) / 1e12
                res.setdefault(f"gemm_{name}_pipe{variant}", []).append((ms, tf))
                if rd == 0 and (variant % 100 < 10 or variant in (20, 28, 30, 40, 60, 70)):  # schedules must not change results: same k order, same MFMA
                    got = out.clone()
                    lib.vsys_tune_gemm_variant(0)
                    fn()
                    lib.vsys_tune_gemm_variant(variant)
                    same = torch.equal(got, out)
                    print(f"  check gemm_{name} variant {variant} == default: {same}", flush=True)
                elif rd == 0 and variant in (80, 24, 16):   # stream-K tail / 16x16x32 MFMAs: fp32 partial sums in another order
                    got = out.clone()
                    lib.vsys_tune_gemm_variant(8)
                    fn()
                    lib.vsys_tune_gemm_variant(variant)
                    d = (got.float() - out.float()).abs()
                    print(f"  check gemm_{name} variant {variant} vs schedule 8: {float((d > 0).float().mean()):.2e} of the elements differ, "
                          f"max |diff| / max |out| = {float(d.max() / out.float().abs().max()):.2e}", flush=True)
    lib.vsys_tune_gemm_variant(0)
    if args.vendor:  # yardstick: the vendor library's plain GEMM + bias (no GELU / gate / residual fusion) on the same operands
        import torch.nn.functional as F

        for name, n, k, epi in shapes:
            w, b, _ = bufs[name]
            a = h if k == 4 * C else x
            ms = timeit(lambda: F.linear(a, w, b), args.reps)
            res[f"vendor_linear_{name}"] = [(ms, 2.0 * N * n * k / (ms * 1e-3) / 1e12)]
    if args.only == "gemm":
        return report(res)
    if args.only == "flash":
        res.clear()

    # attention: spatial (38 frames x 1024), cross (2 x 19456 q, 300 keys), temporal
    qkv = rnd(N, 3 * C)
    qw = rnd(72) + 1
    ao = torch.empty(N, C, dtype=torch.bfloat16, device=dev)
    kp, vt = ops.alloc_kv_buffers(38, H, 1024, dev)
    ms = timeit(lambda: ops.attn_prep_kv(qkv[:, C:2 * C], qkv[:, 2 * C:], qw, kp, vt, 38, H, 1024), args.reps)
    res["attn_prep_kv_spatial"] = [(ms, (179.4 + 209.2) / ms)]  # GB/s
    for rd in range(args.rounds):   # interleaved rounds: the first kernel timed after a pause runs on a colder clock
        for fv in [int(v) for v in args.flash_variants.split(',')]:
            assert lib.vsys_tune_flash_variant(fv) == 0, f"flash variant {fv} is not in this build"
            ms = timeit(lambda: ops.flash_attn(qkv[:, :C], qw, kp, vt, ao, 38, H, 1024, 1024), args.reps)
            res.setdefault("flash_spatial" + (f"_v{fv}" if fv else ""), []).append((ms, 4.0 * 38 * H * 1024 * 1024 * 72 / (ms * 1e-3) / 1e12))
    lib.vsys_tune_flash_variant(0)
    kv = rnd(600, 2 * C)
    kpc, vtc = ops.alloc_kv_buffers(2, H, 300, dev)
    ops.attn_prep_kv(kv[:, :C], kv[:, C:], None, kpc, vtc, 2, H, 300)
    for rd in range(0 if os.environ.get("VSYS_KB_SPATIAL_ONLY") else args.rounds):
        for fv in [int(v) for v in args.flash_variants.split(',')]:
            assert lib.vsys_tune_flash_variant(fv) == 0
            ms = timeit(lambda: ops.flash_attn(x, None, kpc, vtc, ao, 2, H, 19456, 300), args.reps)
            res.setdefault("flash_cross_L300" + (f"_v{fv}" if fv else ""), []).append((ms, 4.0 * 2 * H * 19456 * 300 * 72 / (ms * 1e-3) / 1e12))
    lib.vsys_tune_flash_variant(0)
    # the padding promise (vsys_flash_attn_d72_exact: no mask on the ragged last tile of the resident-K/V kernel) against the masked kernel
    for rd in range(0 if os.environ.get("VSYS_KB_SPATIAL_ONLY") else args.rounds):
        for tag, ex in (("masked", False), ("exact", True)):
            ms = timeit(lambda: ops.flash_attn(x, None, kpc, vtc, ao, 2, H, 19456, 300, keys_exact=ex), args.reps)
            res.setdefault("flash_cross_L300_" + tag, []).append((ms, 4.0 * 2 * H * 19456 * 300 * 72 / (ms * 1e-3) / 1e12))
    # every flash variant must give the default's bits (they differ in schedule only)
    ref_s, ref_c = torch.empty_like(ao), torch.empty_like(ao)
    ops.flash_attn(qkv[:, :C], qw, kp, vt, ref_s, 38, H, 1024, 1024)
    ops.flash_attn(x, None, kpc, vtc, ref_c, 2, H, 19456, 300)
    for fv in sorted(set(int(v) for v in args.flash_variants.split(',')) - {0}):
        lib.vsys_tune_flash_variant(fv)
        o1, o2 = torch.empty_like(ao), torch.empty_like(ao)
        ops.flash_attn(qkv[:, :C], qw, kp, vt, o1, 38, H, 1024, 1024)
        ops.flash_attn(x, None, kpc, vtc, o2, 2, H, 19456, 300)
        print(f"  check flash variant {fv} == default: spatial {bool(torch.equal(o1, ref_s))}, cross {bool(torch.equal(o2, ref_c))}")
    lib.vsys_tune_flash_variant(0)
    if args.only == "flash":
        return report(res)
    freqs = 1.0 / (10000 ** (torch.arange(0, 72, 2).float() / 72))
    ang = torch.einsum("p,f->pf", torch.arange(19).float(), freqs).repeat_interleave(2, -1)
    cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
    for rd in range(args.rounds):       # interleaved rounds (21 = the reference's rounding points stage by stage: the round-4 kernel)
        for fv in sorted(set([0, 21] + [int(v) for v in args.flash_variants.split(',')])):
            lib.vsys_tune_flash_variant(fv)
            ms = timeit(lambda: ops.attn_temporal(qkv, C, qw, qw, cos, sin, ao, 2, 19, 1024, H), args.reps)
            res.setdefault("attn_temporal" + (f"_v{fv}" if fv else ""), []).append((ms, (269.0 + 89.7) / ms))  # GB/s
    lib.vsys_tune_flash_variant(0)
    ms = timeit(lambda: ops.adaln_modulate(x, mod[0, :C], mod[0, C:2 * C], N // 2, 6 * C, out=ao), args.reps)
    res["adaln_modulate"] = [(ms, 179.3 / ms)]  # GB/s
    ms = timeit(lambda: ops.add_rows(ao, x), args.reps)
    res["add_rows"] = [(ms, 269.0 / ms)]

    report(res)


def report(res):
    out = {}
    print(f"{'kernel':32s} {'ms(min)':>9s} {'ms(med)':>9s} {'rate(max)':>10s}")
    for k, v in res.items():
        mss = sorted(m for m, _ in v)
        rate = max(r for _, r in v)
        out[k] = {"ms_min": mss[0], "ms_med": mss[len(mss) // 2], "rate_max": rate}
        print(f"{k:32s} {mss[0]:9.4f} {mss[len(mss) // 2]:9.4f} {rate:10.1f}")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "kernel_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
