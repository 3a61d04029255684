#!/usr/bin/env python
"""Where do the GEMM's watts go?  One experiment, one table.

For the qkv / fc1 / fc2 shapes of config 2 (38 912 token rows) run the same kernel family in four modes

  shipped   the shipped shape dispatch on random operands (what the denoise step runs)
  sched8    schedule 8 forced for every shape on random operands (the like-for-like partner of ``a_l2``)
  a_l2      schedule 8 with every tile streaming the A rows of tile 0 (lab id 38): A becomes L2-resident, the fabric carries W only
            — the "what if the activation traffic were gone" bound; output is NOT valid
  zeros     the shipped dispatch on all-zero operands (the matrix pipe's switching energy removed, traffic unchanged)

and report, per (mode, shape): sustained TFLOP/s over ~3 s back to back, shader clock and socket power from rocm-smi while it runs.
A second invocation under ``rocprofv3 --pmc`` (``--pmc-workload``: three launches of every (mode, shape) in the same order) gives
FETCH_SIZE / MfmaUtil per launch; ``--report time.jsonl fetch.db [mfma.db]`` joins the two into profiles/r06_gemm_power_traffic.json.

Needs the lab build for ``a_l2`` (VSYS_LIB=videosys_amd/libvideosys_amd_lab.so); without it that mode is skipped.
"""
import json
import math
import os
import sqlite3
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MODES = [("shipped", 0, False), ("sched8", 8, False), ("a_l2", 38, False), ("zeros", 0, True)]
SHAPES = [("qkv", 3456, 1152, 0), ("fc1", 4608, 1152, 1), ("fc2", 1152, 4608, 2)]
M = 38912


def report(argv):
    times = {}
    for ln in open(argv[0]):
        ln = ln.strip()
        if ln.startswith("{"):
            j = json.loads(ln)
            if "mode" in j:
                times[(j["mode"], j["shape"])] = j
    order = [(m, s) for m, _, _ in MODES for s, _, _, _ in SHAPES]

    def per_dispatch(db, counter):
        cur = sqlite3.connect(db).cursor()
        rows = cur.execute("select dispatch_id, sum(value), min(end - start) from counters_collection where counter_name = ? and "
                           "kernel_name like '%gemm%kernel%' group by dispatch_id order by dispatch_id", (counter,)).fetchall()
        return rows

    out = {"rows": []}
    fetch = per_dispatch(argv[1], "FETCH_SIZE")
    extra = {}
    if len(argv) > 2:
        for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
            extra[c] = per_dispatch(argv[2], c)
    present = [k for k in order if k in times]
    assert len(fetch) == 3 * len(present), (len(fetch), len(present))
    for i, k in enumerate(present):
        t = times[k]
        n, kk = next((n, kk) for s, n, kk, _ in SHAPES if s == k[1])
        row = dict(mode=k[0], shape=k[1], tflops=t["tflops"], sclk_mhz=t.get("sclk_mhz"), power_w=t.get("power_w"),
                   fetch_mb=round(sum(v for _, v, _ in fetch[3 * i:3 * i + 3]) / 3 * 1024 * 2 / 1e6, 1),
                   algorithmic_read_mb=round((M * kk + n * kk) * 2 / 1e6, 1))
        if extra:
            busy = sum(v for _, v, _ in extra["SQ_VALU_MFMA_BUSY_CYCLES"][3 * i:3 * i + 3]) / 3
            act = sum(v for _, v, _ in extra["GRBM_GUI_ACTIVE"][3 * i:3 * i + 3]) / 3
            dur = sum(d for _, _, d in extra["GRBM_GUI_ACTIVE"][3 * i:3 * i + 3]) / 3
            row["mfma_util"] = round(busy / (act / 8 * 1024), 3)
            row["pmc_pass_clock_ghz"] = round(act / 8 / dur, 3)
        out["rows"].append(row)
    out["note"] = ("tflops / sclk / power: ~3 s of back-to-back launches per row, rocm-smi sampled every 0.2 s (median); fetch_mb: rocprofv3 "
                   "--pmc FETCH_SIZE x 2 (MI355X_MICROARCH.md gfx950 correction), mean of 3 launches, fabric side (Infinity-Cache hits "
                   "included); mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) in the (serialised) counter pass")
    print(json.dumps(out, indent=1))


def main():
    if "--report" in sys.argv:
        return report(sys.argv[sys.argv.index("--report") + 1:])
    import torch

    from videosys_amd import _lib, ops

    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    pmc = "--pmc-workload" in sys.argv
    seconds = 3.0

    def rnd(*shape, scale=1.0, zero=False):
        if zero:
            return torch.zeros(*shape, dtype=torch.bfloat16, device=dev)
        return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(dev)

    data = {}
    for zero in (False, True):
        x, h = rnd(M, 1152, zero=zero), rnd(M, 4608, zero=zero)
        mod = rnd(2, 6 * 1152, scale=0.3, zero=zero)
        resid = rnd(M, 1152, zero=zero)
        ws = {s: (rnd(n, k, scale=1 / math.sqrt(k), zero=zero), rnd(n, scale=0.1, zero=zero)) for s, n, k, _ in SHAPES}
        data[zero] = (x, h, mod, resid, ws)
    outs = {s: torch.empty(M, n, dtype=torch.bfloat16, device=dev) for s, n, _, _ in SHAPES}
    samples = []
    stop = [False]

    def poll():
        while not stop[0]:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True)
            try:
                j = json.loads(r.stdout)
                c = j[sorted(j)[0]]
                sclk = next((v for k, v in c.items() if "sclk" in k.lower()), None)
                pw = next((v for k, v in c.items() if "power" in k.lower() and "W" in k), None)
                samples.append((time.perf_counter(), sclk, pw))
            except Exception:
                pass
            time.sleep(0.2)

    def num(v):
        if v is None:
            return None
        s = "".join(ch for ch in str(v).replace("Mhz", "").replace("(", "").replace(")", "") if ch.isdigit() or ch == ".")
        try:
            return float(s)
        except ValueError:
            return None

    th = None
    if not pmc:
        th = threading.Thread(target=poll)
        th.start()
    for mode, variant, zero in MODES:
        if lib.vsys_tune_gemm_variant(variant) != 0:
            print(json.dumps({"skipped": mode, "why": f"GEMM variant {variant} is not in this build (lab build needed)"}), flush=True)
            continue
        x, h, mod, resid, ws = data[zero]
        for s, n, k, epi in SHAPES:
            w, b = ws[s]
            a = h if k == 4608 else x
            out = outs[s]
            if epi == 2:
                fn = lambda: ops.gemm(a, w, b, epilogue=2, gate=mod[0, 2304:3456], gate_stride=6912, rows_per_sample=M // 2, res=resid, out=out)
            else:
                fn = lambda: ops.gemm(a, w, b, epilogue=epi, out=out)
            if pmc:
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                continue
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            cnt = 0
            while time.perf_counter() - t0 < seconds:
                for _ in range(100):
                    fn()
                torch.cuda.synchronize()
                cnt += 100
            t1 = time.perf_counter()
            win = [(c, p) for t, c, p in samples if t0 + 0.6 <= t <= t1]
            clk = sorted(num(c) for c, _ in win if num(c))
            pw = sorted(num(p) for _, p in win if num(p))
            print(json.dumps({"mode": mode, "shape": s, "variant": variant, "zero_operands": zero,
                              "tflops": round(2.0 * M * n * k * cnt / (t1 - t0) / 1e12, 1), "us": round((t1 - t0) / cnt * 1e6, 1),
                              "sclk_mhz": clk[len(clk) // 2] if clk else None, "power_w": pw[len(pw) // 2] if pw else None,
                              "n_samples": len(win)}), flush=True)
    lib.vsys_tune_gemm_variant(0)
    stop[0] = True
    if th:
        th.join()


if __name__ == "__main__":
    main()
