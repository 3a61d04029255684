#!/usr/bin/env python
"""GEMM tile rasters (vsys_tune_gemm_variant(20000 + 100 gw + ph), csrc/common.h gemm_raster) at the config-2 shapes.

  python tools/gemm_raster_probe.py                      time: HIP events, interleaved rounds, JSON lines
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/gemm_raster_probe.py --pmc   one launch per (raster, shape) in a fixed
      order + a calibration launch (N = 192: A is read exactly once, so FETCH_SIZE x 2 must equal its 89.7 MB);
  python tools/gemm_raster_probe.py --report <db> [...]  per (raster, shape): counter sums and duration from those databases."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RASTERS = [int(v) for v in os.environ.get("VSYS_RASTERS", "600,1200,400,300,808,605,1805,2404").split(",")]
N, C = 38912, 1152
SHAPES = [("qkv", 3 * C, C, 0), ("fc1", 4 * C, C, 1), ("fc2", C, 4 * C, 2), ("proj", C, C, 2)]


def report(paths):
    import sqlite3

    order = [("cal", 0)] + [(s[0], r) for r in RASTERS for s in SHAPES]
    out = {}
    for p in paths:
        cur = sqlite3.connect(p).cursor()
        rows = cur.execute("select dispatch_id, counter_name, sum(value), min(end - start) from counters_collection where kernel_name "
                           "like '%gemm%kernel%' group by dispatch_id, counter_name order by dispatch_id").fetchall()
        ids = sorted({r[0] for r in rows})
        assert len(ids) == len(order), (len(ids), len(order))
        for d, c, v, dur in rows:
            name, ras = order[ids.index(d)]
            e = out.setdefault(f"{name}@{ras}", {})
            e[c] = v
            e.setdefault("dur_us", []).append(dur / 1e3)
    for k, e in out.items():
        e["dur_us"] = round(sum(e["dur_us"]) / len(e["dur_us"]), 1)
        if "FETCH_SIZE" in e:
            e["read_MB_x2"] = round(e["FETCH_SIZE"] * 1024 * 2 / 1e6, 1)
        if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e:
            e["l2_hit"] = round(e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"]), 4)
        print(json.dumps({"case": k, **e}))


def main():
    if "--report" in sys.argv:
        return report(sys.argv[sys.argv.index("--report") + 1:])
    import torch

    from videosys_amd import _lib, ops

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    lib = _lib.load()

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(dev)

    x, h = rnd(N, C), rnd(N, 4 * C)
    mod, resid = rnd(2, 6 * C, scale=0.3), rnd(N, C)
    bufs = {name: (rnd(n, k, scale=1 / math.sqrt(k)), rnd(n, scale=0.1), torch.empty(N, n, dtype=torch.bfloat16, device=dev))
            for name, n, k, epi in SHAPES}

    def run(name, n, k, epi):
        w, b, out = bufs[name]
        a = h if k == 4 * C else x
        if epi == 2:
            ops.gemm(a, w, b, epilogue=epi, gate=mod[0, 2 * C:3 * C], gate_stride=6 * C, rows_per_sample=N // 2, res=resid, out=out)
        else:
            ops.gemm(a, w, b, epilogue=epi, out=out)

    if "--pmc" in sys.argv:
        wc, bc, oc = rnd(192, C, scale=1 / math.sqrt(C)), rnd(192, scale=0.1), torch.empty(N, 192, dtype=torch.bfloat16, device=dev)
        ops.gemm(x, wc, bc, out=oc)
        for r in RASTERS:
            assert lib.vsys_tune_gemm_variant(20000 + r) == 0
            for s in SHAPES:
                run(*s)
        torch.cuda.synchronize()
        return
    res = {}
    for rnd_i in range(6):
        for r in RASTERS:
            assert lib.vsys_tune_gemm_variant(20000 + r) == 0
            for s in SHAPES:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                run(*s)
                e0.record()
                for _ in range(5):
                    run(*s)
                e1.record()
                torch.cuda.synchronize()
                if rnd_i:
                    res.setdefault((s[0], r), []).append(e0.elapsed_time(e1) / 5)
    for (name, r), v in sorted(res.items()):
        v.sort()
        fl = 2.0 * N * [s for s in SHAPES if s[0] == name][0][1] * [s for s in SHAPES if s[0] == name][0][2]
        print(json.dumps({"shape": name, "raster_gw": r // 100, "raster_ph": r % 100, "ms_median": round(v[len(v) // 2], 4),
                          "ms_min": round(v[0], 4), "tflops_median": round(fl / v[len(v) // 2] / 1e9, 1)}))
    lib.vsys_tune_gemm_variant(20600)


if __name__ == "__main__":
    main()
