"""Time CogVideoXVAE.decode at BASELINE config 5 (latent [1, 16, 13, 60, 90] -> 49 frames of 480 x 720, tiled 3 x 3), synthetic
weights.   python tools/cogvideox_vae_bench.py [--no-tiling]"""
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
    print(json.dumps({"workload": f"CogVideoXVAE.decode latent [1,16,13,60,90] -> {list(v.shape)}, tiling={not args.no_tiling}",
                      "sec_per_decode": round(min(ts), 4), "max_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))


if __name__ == "__main__":
    main()
