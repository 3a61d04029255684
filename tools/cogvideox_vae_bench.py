"""Time CogVideoXVAE.decode at BASELINE config 5 (latent [1, 16, 13, 60, 90] -> 49 frames of 480 x 720, tiled 3 x 3), synthetic
weights.   python tools/cogvideox_vae_bench.py [--no-tiling] [--shard P]
--shard P: every rank's share of the tiled decode with its tiles shared out over P ranks (gather stubbed: tools/local_group.StubGroup)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-tiling", action="store_true")
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--shard", type=int, default=0)
    args = ap.parse_args()
    from videosys_amd.vae_cogvideox import CogVideoXVAE, synth_state_dict

    dev = torch.device("cuda:0")
    vae = CogVideoXVAE(synth_state_dict(0), device=dev, use_tiling=not args.no_tiling)
    z = torch.randn(1, 16, 13, 60, 90, generator=torch.Generator().manual_seed(0)).to(dev)
    v = vae.decode(z)
    torch.cuda.synchronize()
    assert torch.isfinite(v.float()).all()
    ts = []
    for _ in range(args.iters):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        vae.decode(z)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    rec = {"workload": f"CogVideoXVAE.decode latent [1,16,13,60,90] -> {list(v.shape)}, tiling={not args.no_tiling}",
           "sec_per_decode": round(min(ts), 4), "max_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}
    if args.shard > 1 and not args.no_tiling:
        from tools.local_group import StubGroup

        per_rank = []
        for r in range(args.shard):
            grp = StubGroup(args.shard, r)
            vae.decode(z, grp)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            vae.decode(z, grp)
            torch.cuda.synchronize(); per_rank.append(round(time.perf_counter() - t0, 4))
        rec.update(shard=args.shard, sec_per_rank=per_rank, slowest_rank_s=max(per_rank),
                   note="tiles dealt by latent area (CogVideoXVAE._deal_tiles); the all-gather is a device copy of the same bytes (results meaningless by construction)")
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
