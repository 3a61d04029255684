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
    names = sorted({c for k in table for c in table[k]})
    for k in sorted(table):
        if "rocclr" in k or "at::" in k:
            continue
        print(k)
        for n in names:
            if n in table[k]:
                print(f"    {n:32s} {table[k][n]:18.6g}")


if __name__ == "__main__":
    main(sys.argv[1:] or sorted(glob.glob("gpurun_out/pmc*/pmc_results.db")))
