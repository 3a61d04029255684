#!/usr/bin/env python
"""Per-shape HBM/fabric traffic of the GEMM from two rocprofv3 --pmc passes of tools/pmc_gemm.py (VSYS_GEMM_ALL_SHAPES=1,
one variant): usage  python tools/gemm_traffic.py <fetch.db> <write.db> > profiles/rNN_gemm_traffic.json
FETCH_SIZE / WRITE_SIZE are KiB; reads are doubled (gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced
streams, MI355X_MICROARCH.md section HBM); WRITE_SIZE matches the output bytes 1:1 on this kernel (269.0 MB at qkv)."""
import json
import sqlite3
import sys

NAMES = ["qkv", "proj", "fc1", "fc2"]


def per_dispatch(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select dispatch_id, sum(value) from counters_collection where counter_name = ? and kernel_name like "
                       "'%gemm%kernel%' group by dispatch_id order by dispatch_id", (counter,)).fetchall()
    return [v for _, v in rows]


def main():
    f, w = per_dispatch(sys.argv[1], "FETCH_SIZE"), per_dispatch(sys.argv[2], "WRITE_SIZE")
    assert len(f) == len(w) == 12, (len(f), len(w))
    out = {}
    for i, n in enumerate(NAMES):
        fr = sum(f[3 * i:3 * i + 3]) / 3 * 1024 * 2
        wr = sum(w[3 * i:3 * i + 3]) / 3 * 1024
        out[n] = {"read_bytes": fr, "write_bytes": wr, "total_bytes": fr + wr}
    # launches per denoise step at config 2: qkv 56, N=K=1152 (proj, cross-q, cross-proj) 168, fc1 56, fc2 56
    mix = {"qkv": 56, "proj": 168, "fc1": 56, "fc2": 56}
    out["avg_bytes_per_launch_config2_mix"] = sum(out[n]["total_bytes"] * c for n, c in mix.items()) / sum(mix.values())
    out["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), per launch; reads x2 per MI355X_MICROARCH.md; "
                   "fabric-side counters: Infinity-Cache hits are included, so this is an upper bound on HBM bytes")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
