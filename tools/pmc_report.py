#!/usr/bin/env python
"""Summarise rocprofv3 --pmc SQLite outputs (rocpd .db): per kernel, counter values summed over all hardware
instances of a dispatch, averaged over dispatches; plus the mean dispatch duration of that pass (us)."""
import glob
import sqlite3
import sys
from collections import defaultdict


def short(k):
    k = k.replace("void ", "").replace("vsys::(anonymous namespace)::", "")
    return k.split("(")[0][:40]


def main(paths):
    table = defaultdict(dict)
    for p in paths:
        cur = sqlite3.connect(p).cursor()
        q = ("select kernel_name, counter_name, dispatch_id, sum(value), min(end - start) from counters_collection "
             "group by kernel_name, counter_name, dispatch_id")
        per = defaultdict(list)
        durs = defaultdict(list)
        for k, c, d, v, dur in cur.execute(q):
            per[(short(k), c)].append(v)
            durs[short(k)].append(dur / 1e3)
        for (k, c), vs in per.items():
            table[k][c] = sum(vs) / len(vs)
        for k, ds in durs.items():
            table[k]["dur_us[" + p.split("/")[-2] + "]"] = sum(ds) / len(ds)
    # derived lines.  Normalisation: a counter value here is the SUM over the hardware instances rocprofv3 reports for a dispatch —
    # GRBM_GUI_ACTIVE has one instance per XCD (8), so GRBM_GUI_ACTIVE / 8 = shader-clock cycles the dispatch was active;
    # SQ_VALU_MFMA_BUSY_CYCLES counts, summed over all 1024 SIMDs (256 CUs x 4), the cycles a SIMD's matrix pipe was busy
    # (32 per v_mfma_f32_32x32x16_bf16: check SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA).  MfmaUtil = busy / (active cycles x 1024).
    for k, t in table.items():
        act = t.get("GRBM_GUI_ACTIVE")
        dur = next((v for n, v in t.items() if n.startswith("dur_us[")), None)
        if act and dur:
            t["derived: effective clock GHz (GRBM_GUI_ACTIVE / 8 / dur)"] = act / 8 / (dur * 1e3)
        if act and t.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            t["derived: MfmaUtil (MFMA_BUSY / (GUI_ACTIVE / 8 x 1024 SIMDs))"] = t["SQ_VALU_MFMA_BUSY_CYCLES"] / (act / 8 * 1024)
            if t.get("SQ_INSTS_MFMA"):
                t["derived: busy cycles per MFMA instruction"] = t["SQ_VALU_MFMA_BUSY_CYCLES"] / t["SQ_INSTS_MFMA"]
    names = sorted({c for k in table for c in table[k]})
    for k in sorted(table):
        if "rocclr" in k or "at::" in k:
            continue
        print(k)
        for n in names:
            if n in table[k]:
                print(f"    {n:70s} {table[k][n]:14.6g}")


if __name__ == "__main__":
    main(sys.argv[1:] or sorted(glob.glob("gpurun_out/pmc*/pmc_results.db")))
