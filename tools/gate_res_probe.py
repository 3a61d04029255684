#!/usr/bin/env python
"""Where the gate + residual epilogue's time goes (schedule 8, proj shape 38912 x 1152 x 1152): plain bias vs gate only vs residual
only vs both (in place / out of place) vs + PAB slab copy.   python tools/gate_res_probe.py"""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from videosys_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M, C = 38912, 1152
lib = _lib.load()


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(dev)


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


x, w, b = rnd(M, C), rnd(C, C, scale=1 / math.sqrt(C)), rnd(C, scale=0.1)
mod, res, out, aux = rnd(2, 6 * C, scale=0.3), rnd(M, C), torch.empty(M, C, dtype=torch.bfloat16, device=dev), torch.empty(M, C, dtype=torch.bfloat16, device=dev)
gk = dict(gate=mod[0, 2 * C:3 * C], gate_stride=6 * C, rows_per_sample=M // 2)
cases = {
    "bias": lambda: ops.gemm(x, w, b, out=out),
    "gate_res_epilogue_no_gate_no_res": lambda: ops.gemm(x, w, b, epilogue=ops.EPI_GATE_RES, out=out),
    "gate_only": lambda: ops.gemm(x, w, b, epilogue=ops.EPI_GATE_RES, out=out, **gk),
    "res_only": lambda: ops.gemm(x, w, b, epilogue=ops.EPI_GATE_RES, res=res, out=out),
    "gate_res": lambda: ops.gemm(x, w, b, epilogue=ops.EPI_GATE_RES, res=res, out=out, **gk),
    "gate_res_in_place": lambda: ops.gemm(x, w, b, epilogue=ops.EPI_GATE_RES, res=res, out=res, **gk),
    "gate_res_aux": lambda: ops.gemm(x, w, b, epilogue=ops.EPI_GATE_RES, res=res, out=out, aux=aux, **gk),
}
r = {}
lib.vsys_tune_gemm_variant(int(os.environ.get("GEMM_VARIANT", "8")))
for rd in range(3):
    for k, fn in cases.items():
        r.setdefault(k, []).append(timeit(fn))
lib.vsys_tune_gemm_variant(0)
print(json.dumps({k: round(min(v), 4) for k, v in r.items()}, indent=1))
