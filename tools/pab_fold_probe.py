"""What a folded PAB broadcast costs: the gate + residual GEMMs of config 2/3 (M = 38912; proj K = 1152, fc2 K = 4608) with their
store phase in every form the model uses — plain, + slab copy (aux), + statistics, + one / two folded broadcasts — next to the
passes the fold removes (add_rows, ln_row_stats).  Prints one JSON object (median ms over --reps launches, HIP events)."""
import argparse
import json
import math
import sys

import torch

sys.path.insert(0, ".")
from videosys_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=38912)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    M, N = a.rows, 1152
    g = torch.Generator().manual_seed(0)
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
    slabs = [torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev) for _ in range(3)]
    gate = torch.randn(2, N, generator=g).to(torch.bfloat16).to(dev)
    st = ops.ln_stats_buffer(M, N, dev)
    out = {}

    def timeit(fn):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(a.reps):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fn()
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
        ts.sort()
        return round(ts[len(ts) // 2], 4)

    for K in (1152, 4608):
        x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
        w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(dev)
        b = torch.zeros(N, dtype=torch.bfloat16, device=dev)
        kw = dict(gate=gate[0], gate_stride=N, rows_per_sample=M // 2)
        r = {}
        r["plain"] = timeit(lambda: ops.gemm(x, w, b, epilogue=ops.EPI_GATE_RES, res=res, out=res, **kw))
        r["aux"] = timeit(lambda: ops.gemm(x, w, b, epilogue=ops.EPI_GATE_RES, res=res, aux=slabs[2], out=res, **kw))
        r["stats"] = timeit(lambda: ops.gemm_stats(x, w, b, st, res=res, out=res, **kw))
        for name, k2 in (("add1", dict(adds=slabs[:1])), ("add2", dict(adds=slabs[:2])), ("add1_stats", dict(adds=slabs[:1], stats=st)),
                         ("add2_stats", dict(adds=slabs[:2], stats=st)), ("aux_stats", dict(aux=slabs[2], stats=st)),
                         ("aux_add1_stats", dict(aux=slabs[2], adds=slabs[:1], stats=st)),
                         ("aux_add2_stats", dict(aux=slabs[2], adds=slabs[:2], stats=st))):
            r[name] = timeit(lambda: ops.gemm_gate_res_add(x, w, b, res=res, out=res, **kw, **k2))
        out[f"K{K}"] = r
    out["add_rows"] = timeit(lambda: ops.add_rows(res, slabs[0]))
    out["ln_row_stats"] = timeit(lambda: ops.ln_row_stats(res, st))
    out["rows"] = M
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
