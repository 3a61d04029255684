"""attn_temporal at the config-2 shape (2 x 1024 tokens x 19 frames x 16 heads): median / min ms over --reps launches and the
algorithmic GB/s (q|k|v read + output written).  python tools/temporal_probe.py [reps [T S]]; VSYS_FV=4 / 9: the VALU / online kernels."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosys_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
C, H = 1152, 16
T_, S_ = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (19, 1024)     # (38, 3600: the 720p x 128f shape)
N = 2 * T_ * S_
qkv = torch.randn(N, 3 * C, generator=g).to(torch.bfloat16).to(dev)
qw = (torch.randn(72, generator=g) * 0.1 + 1).to(torch.bfloat16).to(dev)
ao = torch.empty(N, C, dtype=torch.bfloat16, device=dev)
freqs = 1.0 / (10000 ** (torch.arange(0, 72, 2).float() / 72))
ang = torch.einsum("p,f->pf", torch.arange(T_).float(), freqs).repeat_interleave(2, -1)
cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
from videosys_amd import _lib as _l
_l.load().vsys_tune_flash_variant(int(os.environ.get("VSYS_FV", "0")))   # 4 = the VALU kernel, 9 = online softmax
for _ in range(5):
    ops.attn_temporal(qkv, C, qw, qw, cos, sin, ao, 2, T_, S_, H)
ts = []
for _ in range(reps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ops.attn_temporal(qkv, C, qw, qw, cos, sin, ao, 2, T_, S_, H)
    e.record()
    torch.cuda.synchronize()
    ts.append(s.elapsed_time(e))
ts.sort()
from videosys_amd import _lib
fv = int(os.environ.get("VSYS_FV", "0"))
print(json.dumps({"frames": T_, "tokens": S_, "flash_variant": fv, "ms_min": round(ts[0], 4), "ms_med": round(ts[len(ts) // 2], 4),
                  "GBps_med": round(N * 4 * C * 2 / 1e6 / ts[len(ts) // 2], 1), "checksum": float(ao.float().sum())}))
