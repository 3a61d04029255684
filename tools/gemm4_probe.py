#!/usr/bin/env python
"""Lab probe of the ping-pong GEMM (gemm4_bf16.hip): ablations (no in-loop DMA / no in-loop fragment reads / neither) against the
full kernel and schedule 8, interleaved rounds, plus per-segment s_memtime stamps split by wave group and an effective-clock
estimate (ticks between the first wave's start and the last wave's end / event-timed duration).
    python tools/gemm4_probe.py > gpurun_out/gemm4_probe.json"""
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
    import contextlib

    import __graft_entry__ as ge

    with contextlib.redirect_stdout(sys.stderr):
        ge.build()
    from videosys_amd import _lib, ops

    lib = _lib.load()
    if not hasattr(lib, "vsys_lab_flash_debug_buffer"):
        sys.exit("this probe uses ablation / stamp variants: build the lab flavour first (VSYS_LAB=1 python -c 'import __graft_entry__ as g; g.build()') and run with VSYS_LAB=1")
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    M = 38912
    res = {}
    variants = [int(v) for v in os.environ.get("G4_VARIANTS", "8,60,70,78").split(",")]
    for name, N, K in (("qkv", 3456, 1152), ("fc2shape", 1152, 4608)):
        x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
        b = torch.zeros(N, dtype=torch.bfloat16, device=dev)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        r = {}
        for rd in range(3):
            for v in variants:
                lib.vsys_tune_gemm_variant(v)
                ms = timeit(lambda: ops.gemm(x, w, b, out=out))
                r.setdefault(f"v{v}", []).append(round(ms, 4))
        lib.vsys_tune_gemm_variant(0)
        r = {k: {"ms_min": min(v), "tflops": round(2.0 * M * N * K / min(v) / 1e9, 1)} for k, v in r.items()}
        # stamps
        blocks = (M // 256) * (N // 192)
        dbg = torch.zeros(blocks * 8 * 8, dtype=torch.int64, device=dev)
        lib.vsys_lab_flash_debug_buffer(dbg.data_ptr())
        lib.vsys_tune_gemm_variant(int(os.environ.get("G4_STAMP_VARIANT", "74")))
        try:
            for _ in range(3):
                ops.gemm(x, w, b, out=out)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); ops.gemm(x, w, b, out=out); e.record(); torch.cuda.synchronize()
            ms64 = s.elapsed_time(e)
        finally:
            lib.vsys_tune_gemm_variant(0)
            lib.vsys_lab_flash_debug_buffer(None)
        d = dbg.view(blocks, 8, 8).double().cpu()
        nt = int(d[0, 0, 7].item())
        names = ["load", "bar_after_load", "mfma", "bar_after_mfma"]
        st = {}
        for gname, sl in (("G0", slice(0, 4)), ("G1", slice(4, 8))):
            st[gname] = {n: round(d[:, sl, i].mean().item() / nt, 1) for i, n in enumerate(names)}
            used = d[:, sl, 4] > 0
            st[gname]["total_ticks_per_wave"] = round(d[:, sl, 4][used].mean().item(), 1)
            st[gname]["epilogue_ticks_per_wave"] = round(d[:, sl, 5][used].mean().item(), 1)
            st[gname] = {k: (round(v * blocks / max(int(used[:, 0].sum().item()), 1), 1) if k in names else v) for k, v in st[gname].items()}
        live = d[:, :, 4] > 0
        begin = d[:, :, 6][live].min().item()
        end = (d[:, :, 6] + d[:, :, 4])[live].max().item()
        r["stamps"] = st
        r["stamped_kernel_ms"] = round(ms64, 4)
        r["ticks_first_start_to_last_end"] = end - begin
        r["ghz_if_memtime_is_shader_clock"] = round((end - begin) / (ms64 * 1e6), 3)
        r["nt"] = nt
        r["tiles_per_cu"] = round(blocks / 256, 2)
        res[name] = r
    res["note"] = ("per K-tile and wave: 24 MFMAs x 32 cycles = 768; two waves per SIMD -> 1536 matrix-pipe cycles per K-tile; "
                   "stamps are s_memtime ticks summed over the K loop / nt")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
