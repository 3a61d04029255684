"""attn_temporal at the config-2 shape (2 x 1024 tokens x 19 frames x 16 heads): median / min ms over --reps launches and the
algorithmic GB/s (269.0 MB of q|k|v read + 89.7 MB written).  VSYS_T3_PREFETCH=0 selects the kernel without the next-head prefetch."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosys_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
N, C, H = 38912, 1152, 16
qkv = torch.randn(N, 3 * C, generator=g).to(torch.bfloat16).to(dev)
qw = (torch.randn(72, generator=g) * 0.1 + 1).to(torch.bfloat16).to(dev)
ao = torch.empty(N, C, dtype=torch.bfloat16, device=dev)
freqs = 1.0 / (10000 ** (torch.arange(0, 72, 2).float() / 72))
ang = torch.einsum("p,f->pf", torch.arange(19).float(), freqs).repeat_interleave(2, -1)
cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for _ in range(5):
    ops.attn_temporal(qkv, C, qw, qw, cos, sin, ao, 2, 19, 1024, H)
ts = []
for _ in range(reps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ops.attn_temporal(qkv, C, qw, qw, cos, sin, ao, 2, 19, 1024, H)
    e.record()
    torch.cuda.synchronize()
    ts.append(s.elapsed_time(e))
ts.sort()
print(json.dumps({"prefetch": os.environ.get("VSYS_T3_PREFETCH", "1") != "0", "ms_min": round(ts[0], 4), "ms_med": round(ts[len(ts) // 2], 4),
                  "GBps_med": round((269.0 + 89.7) / ts[len(ts) // 2], 1), "checksum": float(ao.float().sum())}))
