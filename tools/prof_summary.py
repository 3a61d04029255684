#!/usr/bin/env python
"""Per-kernel summary (calls, total ms, avg/min/max us, % of kernel time) of a rocprofv3 --kernel-trace rocpd database.
Usage: python tools/prof_summary.py gpurun_out/prof/bench_results.db ["header line"] > profiles/<name>.txt"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                            "from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    if len(sys.argv) > 2:
        print(sys.argv[2])
    print(f"{'kernel':100s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for n, c, s, a, mn, mx in rows:
        print(f"{n[:100]:100s} {c:7d} {s / 1e6:10.2f} {a / 1e3:10.1f} {mn / 1e3:9.1f} {mx / 1e3:9.1f} {100.0 * s / tot:6.2f}")
    print(f"TOTAL kernel time {tot / 1e6:.1f} ms")


if __name__ == "__main__":
    main()
