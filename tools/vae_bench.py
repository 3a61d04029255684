"""Time OpenSoraVAE.decode at the BASELINE config-2 size (latent [1,4,19,64,64] -> 64 frames of 512x512), synthetic weights.
    python tools/vae_bench.py [--frames 64] [--hw 64] [--iters 3] [--fpl 16]
Prints one JSON line: seconds per decode, algorithmic conv TFLOP and the achieved rate."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from videosys_amd.vae_open_sora import OpenSoraVAE, decoder_param_shapes, synth_state_dict  # noqa: E402


def conv_flops(num_frames, H, W, micro=17):
    """2*M*K*N over every conv / linear of the decode path (interior positions only; attention products included)."""
    shp = decoder_param_shapes()
    fl = 0.0
    # temporal VAE: per micro-batch of Tz latent frames
    left, Tzs = num_frames, []
    while left > 0:
        nf = min(micro, left); Tzs.append((nf + 3) // 4); left -= micro
    for Tz in Tzs:
        T = {"conv1": Tz, "res_blocks": Tz, "block_res_blocks.3": Tz, "conv_blocks.2": Tz, "block_res_blocks.2": 2 * Tz, "conv_blocks.1": 2 * Tz,
             "block_res_blocks.1": 4 * Tz, "block_res_blocks.0": 4 * Tz, "conv_out": 4 * Tz}
        for k, s in shp.items():
            if not k.startswith("temporal_vae.decoder.") or not k.endswith("conv.weight"):
                continue
            for tag, t in T.items():
                if "decoder." + tag in k:
                    co, ci = s[0], s[1]
                    taps = s[2] * s[3] * s[4]
                    fl += 2.0 * t * H * W * co * ci * taps
                    break
    # 2-D decoder per frame
    per = 0.0
    res = {"mid_block": 1, "up_blocks.0.resnets": 1, "up_blocks.0.upsamplers": 2, "up_blocks.1.resnets": 2, "up_blocks.1.upsamplers": 4,
           "up_blocks.2.resnets": 4, "up_blocks.2.upsamplers": 8, "up_blocks.3.resnets": 8, "conv_in": 1, "conv_out": 8}
    for k, s in shp.items():
        if not k.startswith("spatial_vae.module.decoder.") or not k.endswith(".weight") or len(s) < 2:
            continue
        for tag, f in res.items():
            if "decoder." + tag in k:
                taps = s[2] * s[3] if len(s) == 4 else 1
                per += 2.0 * (H * f) * (W * f) * s[0] * s[1] * taps
                break
    L = H * W
    per += 2 * 2.0 * L * L * 512  # QK^T and PV
    return fl + per * num_frames


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--hw", type=int, default=64)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--fpl", type=int, default=16)
    ap.add_argument("--encode", type=int, default=0, help="N > 0: time OpenSoraVAE.encode of N frames (image / video conditioning) instead")
    ap.add_argument("--shard", type=int, default=0, help="P > 1: time every rank's share of decode_sharded (wire stubbed: the gather is a "
                                                         "device copy of the same bytes, tools/local_group.StubGroup) and report the slowest")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if args.shard > 1:
        from tools.local_group import StubGroup

        vae = OpenSoraVAE(synth_state_dict(0), device=dev, frames_per_launch=args.fpl)
        Tz = vae.get_latent_size((args.frames, args.hw * 8, args.hw * 8))[0]
        z = torch.randn(1, 4, Tz, args.hw, args.hw, generator=torch.Generator().manual_seed(0)).to(dev)
        per_rank = []
        for r in range(args.shard):
            grp = StubGroup(args.shard, r)
            vae.decode_sharded(z, args.frames, grp)
            ts = []
            for _ in range(args.iters):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                vae.decode_sharded(z, args.frames, grp)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            per_rank.append(round(min(ts), 4))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        vae.decode(z, args.frames)
        torch.cuda.synchronize(); whole = time.perf_counter() - t0
        print(json.dumps({"workload": f"OpenSoraVAE.decode_sharded latent [1,4,{Tz},{args.hw},{args.hw}] -> {args.frames} frames, "
                                      f"P = {args.shard}, wire stubbed", "sec_per_rank": per_rank, "slowest_rank_s": max(per_rank),
                          "frames_per_rank": [list(v) for v in vae.frame_shards(args.frames, args.shard)],
                          "unsharded_s": round(whole, 4), "gathered_bytes": args.frames * (args.hw * 8) ** 2 * 3}))
        return
    if args.encode:
        vae = OpenSoraVAE(synth_state_dict(0, encoder=True), device=dev, frames_per_launch=args.fpl)
        x = (torch.rand(1, 3, args.encode, args.hw * 8, args.hw * 8, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
        noise = lambda shape: torch.zeros(shape)      # (the posterior draw itself is not what is timed)
        zz = vae.encode(x, noise_fn=noise)
        torch.cuda.synchronize()
        assert torch.isfinite(zz).all()
        ts = []
        for _ in range(args.iters):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            vae.encode(x, noise_fn=noise)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(json.dumps({"workload": f"OpenSoraVAE.encode {args.encode} frames {args.hw * 8}x{args.hw * 8} -> latent {list(zz.shape)}",
                          "sec_per_encode": round(min(ts), 4), "all": [round(t, 4) for t in ts],
                          "max_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))
        return
    vae = OpenSoraVAE(synth_state_dict(0), device=dev, frames_per_launch=args.fpl)
    Tz = vae.get_latent_size((args.frames, args.hw * 8, args.hw * 8))[0]
    z = torch.randn(1, 4, Tz, args.hw, args.hw, generator=torch.Generator().manual_seed(0)).to(dev)
    vid = vae.decode(z, args.frames)
    torch.cuda.synchronize()
    assert torch.isfinite(vid.float()).all()
    ts = []
    for _ in range(args.iters):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        vae.decode(z, args.frames)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    fl = conv_flops(args.frames, args.hw, args.hw)
    best = min(ts)
    print(json.dumps({"workload": f"OpenSoraVAE.decode latent [1,4,{Tz},{args.hw},{args.hw}] -> {args.frames} frames {args.hw * 8}x{args.hw * 8}",
                      "sec_per_decode": round(best, 4), "all": [round(t, 4) for t in ts], "algorithmic_tflop": round(fl / 1e12, 2),
                      "tflops": round(fl / best / 1e12, 1), "frames_per_launch": args.fpl,
                      "max_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))


if __name__ == "__main__":
    main()
