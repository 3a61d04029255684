"""How long does the HOST need to enqueue one config-2 denoise step (Python + ctypes + hipLaunch), compared with the GPU time
of the step?  At N GPUs the device time shrinks ~N-fold, the enqueue time does not: it bounds the strong-scaling efficiency.
    python tools/issue_time.py [--shard 8]   (--shard P: run on 1/P of the pixels, the per-rank token count of P-way DSP)"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shard", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    from videosys_amd import ops
    from videosys_amd.stdit3 import STDiT3, STDiT3Config, synth_state_dict

    dev = torch.device("cuda:0")
    cfg = STDiT3Config()
    model = STDiT3(cfg, device=dev)
    model.load_state_dict(synth_state_dict(cfg, seed=1234))
    g = torch.Generator().manual_seed(0)
    H = 64 // args.shard if args.shard > 1 else 64
    z = torch.randn(1, 4, 19, H, 64, generator=g).to(torch.bfloat16).float().to(dev)
    y = (torch.randn(1, 1, 300, cfg.caption_channels, generator=g) * 0.1).to(torch.bfloat16)
    y = torch.cat([y, model.y_embedder.y_embedding[None, None].cpu().to(y.dtype)], 0).to(dev)
    mask = torch.ones(1, 300, dtype=torch.long)
    kw = dict(mask=mask, fps=torch.tensor([24.0, 24.0]), height=torch.tensor([512.0] * 2), width=torch.tensor([512.0] * 2))
    t = torch.tensor([500.0, 500.0])

    def step():
        out = model(torch.cat([z, z], 0), t, y, **kw)
        ops.cfg_euler_step(z, out, 7.0, 0.01)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    issue, total = [], []
    for _ in range(args.steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        issue.append(t1 - t0)
        total.append(t2 - t0)
    print(json.dumps({"shard": args.shard, "tokens": 2 * 19 * (H // 2) * 32, "host_enqueue_ms": round(1e3 * min(issue), 2),
                      "step_ms": round(1e3 * min(total), 2)}))


if __name__ == "__main__":
    main()
