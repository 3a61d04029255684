#!/usr/bin/env python
"""Yardstick run for the GEMM family: a square-ish problem with a long K loop and an exact number of tile rounds
(M = 8192, N = 3072, K = 4096: 512 tiles of 256 x 192 = 2 rounds on 256 CUs), operands uniform in [-1, 1) — the data
cdna_hip_programming.md quotes its 256^2 8-phase template on (1320-1340 TFLOP/s at 4096^3) — next to the config-2 shapes.
    python tools/gemm_yardstick.py > gpurun_out/gemm_yardstick.json"""
import contextlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


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


def main():
    import __graft_entry__ as ge

    with contextlib.redirect_stdout(sys.stderr):
        ge.build()
    from videosys_amd import _lib, ops

    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    variants = [int(v) for v in os.environ.get("G4_VARIANTS", "8,20,70").split(",")]
    res = {}
    for name, M, N, K, fill in (("square_uniform", 8192, 3072, 4096, "uniform"), ("square_zero", 8192, 3072, 4096, "zero"),
                                ("square_randn", 8192, 3072, 4096, "randn"), ("qkv_uniform", 38912, 3456, 1152, "uniform")):
        if fill == "uniform":
            x = (torch.rand(M, K, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
            w = (torch.rand(N, K, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
        elif fill == "zero":
            x = torch.zeros(M, K, dtype=torch.bfloat16, device=dev)
            w = torch.zeros(N, K, dtype=torch.bfloat16, device=dev)
        else:
            x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
            w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        r = {}
        for rd in range(3):
            for v in variants:
                lib.vsys_tune_gemm_variant(v)
                ms = timeit(lambda: ops.gemm(x, w, None, out=out))
                r.setdefault(f"v{v}", []).append(ms)
            ms = timeit(lambda: torch.nn.functional.linear(x, w))
            r.setdefault("vendor_linear", []).append(ms)
        lib.vsys_tune_gemm_variant(0)
        res[name] = {k: {"ms_min": round(min(v), 4), "tflops": round(2.0 * M * N * K / min(v) / 1e9, 1)} for k, v in r.items()}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
