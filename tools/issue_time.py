"""How long does the HOST need to enqueue one config-2 denoise step (Python + ctypes + hipLaunch), compared with the GPU time
of the step?  At N GPUs the device time shrinks ~N-fold, the enqueue time does not: it bounds the strong-scaling efficiency.
    python tools/issue_time.py [--shard 8]   (--shard P: run on 1/P of the pixels, the per-rank token count of P-way DSP)
    python tools/issue_time.py --dsp-rank 8 [--rank 0] [--scatter flat|sample] [--no-overlap] [--geometry 720p128f]
        the REAL per-rank sequence of a P-way DSP run (full-size patch embed, split, S-shard at rest, padded T-shard around the
        spatial attention, HIP pack / unpack launches, both side streams, gather, final layer) with the wire stubbed
        (tools/local_group.StubGroup: recv <- send as a device copy): per-rank device time and host enqueue time on ONE GPU."""
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
    ap.add_argument("--dsp-rank", type=int, default=0, help="P: emulate one rank of a P-way DSP group (collectives stubbed)")
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--scatter", default=None, choices=["flat", "sample"])
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--cp", action="store_true", help="with --dsp-rank P: the reference's enable_cp decomposition — this rank holds ONE sample of the "
                                                      "CFG pair and splits its sequence over P / 2 ranks (bench.py's default for an even N)")
    ap.add_argument("--geometry", default="512x512x64f", choices=["512x512x64f", "720p128f"])
    ap.add_argument("--depth", type=int, default=28)
    ap.add_argument("--sweep", action="store_true", help="with --dsp-rank: every (scatter, overlap) setting in one process")
    ap.add_argument("--program", type=int, default=-1, help="1 / 0: force the recorded launch program on / off (default: model default)")
    ap.add_argument("--x-mask", type=int, default=0, help="N > 0: the conditioned step — the first N latent frames are conditioning frames (x_mask False)")
    args = ap.parse_args()
    from videosys_amd import ops
    from videosys_amd.stdit3 import STDiT3, STDiT3Config, synth_state_dict

    dev = torch.device("cuda:0")
    cfg = STDiT3Config(depth=args.depth)
    model = STDiT3(cfg, device=dev)
    model.load_state_dict(synth_state_dict(cfg, seed=1234))
    if args.program >= 0 and hasattr(model, "use_programs"):
        model.use_programs = bool(args.program)
    g = torch.Generator().manual_seed(0)
    T, HH, WW = (19, 64, 64) if args.geometry == "512x512x64f" else (38, 90, 160)
    H = HH // args.shard if args.shard > 1 else HH
    z = torch.randn(1, 4, T, H, WW, generator=g).to(torch.bfloat16).float().to(dev)
    if args.dsp_rank > 1:
        from types import SimpleNamespace

        from tools.local_group import StubGroup

        P = args.dsp_rank
        if args.cp:
            assert P % 2 == 0
            sp = P // 2
            pm = SimpleNamespace(sp_size=sp, cp_size=2, dp_size=1, dp_rank=0, sp_rank=args.rank % sp, cp_rank=0,
                                 sp_group=StubGroup(sp, args.rank % sp) if sp > 1 else None, cp_group=StubGroup(2, 0))
        else:
            pm = SimpleNamespace(sp_size=P, cp_size=1, dp_size=1, dp_rank=0, sp_rank=args.rank, cp_rank=0,
                                 sp_group=StubGroup(P, args.rank), cp_group=None)
        model.enable_parallel(parallel_mgr=pm, overlap=not args.no_overlap)
        if args.scatter:
            model._scatter = args.scatter
    y = (torch.randn(1, 1, 300, cfg.caption_channels, generator=g) * 0.1).to(torch.bfloat16)
    y = torch.cat([y, model.y_embedder.y_embedding[None, None].cpu().to(y.dtype)], 0).to(dev)
    mask = torch.ones(1, 300, dtype=torch.long)
    kw = dict(mask=mask, fps=torch.tensor([24.0, 24.0]), height=torch.tensor([HH * 8.0] * 2), width=torch.tensor([WW * 8.0] * 2))
    t = torch.tensor([500.0, 500.0])
    if args.x_mask:
        xm = torch.ones(2, T, dtype=torch.bool)
        xm[:, :args.x_mask] = False
        kw["x_mask"] = xm

    def step():
        out = model(torch.cat([z, z], 0), t, y, **kw)
        ops.cfg_euler_step(z, out, 7.0, 0.01)

    def measure():
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
        rec = {"shard": args.shard, "geometry": args.geometry, "depth": args.depth, "tokens": 2 * T * (H // 2) * (WW // 2),
               "host_enqueue_ms": round(1e3 * min(issue), 2), "step_ms": round(1e3 * min(total), 2)}
        if args.dsp_rank > 1:
            from videosys_amd import dsp

            rec.update(dsp_rank=args.dsp_rank, rank=args.rank, scatter=model._scatter, overlap=model._overlap, cfg_parallel=2 if args.cp else 1,
                       frames_on_this_rank_padded=(dsp.frames_per_rank(1, T, args.dsp_rank // 2, model._scatter) if args.cp and args.dsp_rank > 2
                                                   else (T if args.cp else dsp.frames_per_rank(2, T, args.dsp_rank, model._scatter))),
                       ideal_frames=round(2 * T / args.dsp_rank, 3),
                       exchange=("one-kernel peer-to-peer (vsys_p2p_exchange; every peer folded onto this rank: the same bytes, the same "
                                 "flag traffic)" if model._sp is not None and model._sp.p2p is not None
                                 else "pack + all_to_all_single + unpack, wire stubbed (device copy recv <- send)"))
        if hasattr(model, "use_programs"):
            rec["launch_program"] = bool(model.use_programs) and not args.x_mask
        if args.x_mask:
            rec["x_mask_conditioning_frames"] = args.x_mask
        print(json.dumps(rec), flush=True)

    if args.sweep and args.dsp_rank > 1:
        for scatter in ("sample", "flat"):
            for overlap in (False, True):
                model.enable_parallel(parallel_mgr=pm, overlap=overlap)
                model._scatter = scatter
                measure()
    else:
        measure()


if __name__ == "__main__":
    main()
