#!/usr/bin/env python
"""adaln_modulate at the config-2 shape (38912 x 1152): time, rate and a hash of the output (used to A/B the rows-per-wave forms:
one row 33.2 us, two rows 31.2 us (shipped), four rows 34-35 us; identical bits)."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosys_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
N, C = 38912, 1152
x = (torch.randn(N, C, generator=g) * 2 + 0.5).to(torch.bfloat16).to(dev)
mod = (torch.randn(2, 6 * C, generator=g) * 0.3).to(torch.bfloat16).to(dev)
out = torch.empty_like(x)


def run():
    ops.adaln_modulate(x, mod[0, :C], mod[0, C:2 * C], N // 2, 6 * C, out=out)


for _ in range(5):
    run()
torch.cuda.synchronize()
best = 1e9
for rd in range(5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        run()
    e.record()
    torch.cuda.synchronize()
    best = min(best, s.elapsed_time(e) / 50)
print(f"adaln_modulate  {best * 1e3:.1f} us  {2 * N * C * 2 / best / 1e9:.2f} TB/s  sha1 {hashlib.sha1(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12]}")
