"""Cycle accounting of one GEMM stage (lab variant 31 of the wide-tile kernel): per wave, summed over the K loop, the s_memtime
cycles spent until the stage's fragment reads have landed / issuing the 24 MFMAs (+ DMA pieces + fragment reloads) / in the
counted vmcnt+lgkmcnt wait / in the stage barrier, and the epilogue.   python tools/gemm_stamps.py [--n 3456 --k 1152]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=38912)
    ap.add_argument("--n", type=int, default=3456)
    ap.add_argument("--k", type=int, default=1152)
    args = ap.parse_args()
    import __graft_entry__ as ge

    ge.build()
    from videosys_amd import _lib, ops

    lib = _lib.load()
    if not hasattr(lib, "vsys_lab_flash_debug_buffer"):
        sys.exit("this probe uses ablation / stamp variants: build the lab flavour first (VSYS_LAB=1 python -c 'import __graft_entry__ as g; g.build()') and run with VSYS_LAB=1")
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(args.m, args.k, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(args.n, args.k, generator=g) / args.k ** 0.5).to(torch.bfloat16).to(dev)
    b = torch.zeros(args.n, dtype=torch.bfloat16, device=dev)
    blocks = ((args.m + 255) // 256) * (args.n // 384)
    dbg = torch.zeros(blocks * 8 * 8, dtype=torch.int64, device=dev)
    lib.vsys_lab_flash_debug_buffer(dbg.data_ptr())
    lib.vsys_tune_gemm_variant(31)
    try:
        for _ in range(3):
            out = ops.gemm(x, w, b)
        torch.cuda.synchronize()
    finally:
        lib.vsys_tune_gemm_variant(0)
        lib.vsys_lab_flash_debug_buffer(None)
    d = dbg.view(blocks, 8, 8).double().cpu()
    nt = int(d[0, 0, 7].item())
    names = ["reads_landed", "mfma_block_issue", "counted_wait", "barrier"]
    per_stage = {n: round(d[:, :, i].mean().item() / nt, 1) for i, n in enumerate(names)}
    loop = d[:, :, 4].mean().item()
    res = {"shape": [args.m, args.n, args.k], "stages": nt, "cycles_per_stage_per_wave": per_stage,
           "stage_total": round(loop / nt, 1), "loop_cycles": round(loop), "epilogue_cycles": round(d[:, :, 5].mean().item()),
           "note": "s_memtime ticks; 24 MFMAs of 32 cycles per wave and stage, two waves per SIMD: 1536 MFMA-pipe cycles per stage if the clock matches"}
    # clock ratio: s_memtime may tick at a fixed reference clock; compare with the event-timed kernel duration
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib.vsys_tune_gemm_variant(30)
    ops.gemm(x, w, b)
    s.record(); ops.gemm(x, w, b); e.record(); torch.cuda.synchronize()
    lib.vsys_tune_gemm_variant(0)
    res["variant30_ms"] = round(s.elapsed_time(e), 4)
    starts = d[:, 0, 6]
    res["tiles_per_cu"] = round(blocks / 256, 2)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
