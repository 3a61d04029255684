#!/usr/bin/env python
"""Few-tile GEMMs (one rank of an 8-way DSP group: 4864 token rows) with HOT weights (the same W every launch, as tools/kernel_bench.py
times them) against COLD weights (a rotation over enough copies of W that a copy left every cache level before its turn comes again —
what a denoise step does: 2.1 GB of weights stream through a 256 MB Infinity Cache every step).  HIP events around every launch."""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosys_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M = int(sys.argv[sys.argv.index("--rows") + 1]) if "--rows" in sys.argv else 4864
g = torch.Generator().manual_seed(0)
SHAPES = [("qkv", 3456, 1152), ("proj", 1152, 1152), ("fc1", 4608, 1152), ("fc2", 1152, 4608)]
out = {"rows": M}
for name, n, k in SHAPES:
    copies = max(2, int(700e6 // (n * k * 2)))          # ~700 MB of distinct weights per rotation
    ws = [(torch.randn(n, k, generator=g) / math.sqrt(k)).to(torch.bfloat16).to(dev) for _ in range(min(copies, 8))]
    while len(ws) < copies:
        ws.append(ws[len(ws) % 8].clone())
    b = torch.zeros(n, dtype=torch.bfloat16, device=dev)
    x = torch.randn(M, k, generator=g).to(torch.bfloat16).to(dev)
    y = torch.empty(M, n, dtype=torch.bfloat16, device=dev)
    res = {}
    for mode in ("hot", "cold", "hot", "cold"):
        ev = []
        for i in range(2 * copies if mode == "cold" else 200):
            w = ws[i % copies] if mode == "cold" else ws[0]
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.gemm(x, w, b, out=y)
            e.record()
            ev.append((s, e))
        torch.cuda.synchronize()
        ts = sorted(s.elapsed_time(e) * 1e3 for s, e in ev[len(ev) // 2:])
        res.setdefault(mode, []).append(round(ts[len(ts) // 2], 1))
    out[name] = {"copies": copies, "us_hot": res["hot"], "us_cold": res["cold"]}
print(json.dumps(out))
