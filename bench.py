#!/usr/bin/env python
"""bench.py — BASELINE metric on MI355X: videos/min and sec/denoise-step for Open-Sora v1.2 STDiT3-XL/2 at
512x512x64f (latent [4,19,64,64] -> 19,456 tokens, CFG batch 2 -> 38,912 token rows, 300 text tokens), bf16.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is ONE denoise step: STDiT3 forward on the CFG-doubled batch (28 spatial + 28 temporal blocks, all GEMMs,
attention, AdaLN, final layer) + the CFG/Euler latent update, inputs resident in HBM, synthetic latents / text
embeddings / random-init weights of the real geometry (no datasets or checkpoints offline).  N > 1 shards ONE video
over the N GPUs with Dynamic Sequence Parallelism (strong scaling).  ``value`` = 60 / (30 * sec_per_step): videos per
minute of the 30-step denoising loop (DiT only — T5 text encode and VAE decode are the "next" rows of SURVEY.md §8f and
are not in the number; said so in config).

Extra objects on the JSON line:
  roofline      dominant kernel = the bf16 MFMA GEMM family (91 % of the step's FLOPs): algorithmic FLOPs per launch /
                average launch duration measured with HIP events on the launch stream in an instrumented replay of
                the same steps; peak 2500 TFLOP/s dense bf16 (MI355X_MICROARCH.md).
  cpu_baseline  the CPU oracle (a port of the reference's fp32 PyTorch path) timed on this box's host cores at the full
                token count: valid_depth 1 and 2, warm-up + 3 repeats, thread-count sweep, fitted to depth 28.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STEPS_PER_VIDEO = 30
PEAK_BF16_TFLOPS = 2500.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--depth", type=int, default=28, help="debug only; the reported config is depth 28")
    ap.add_argument("--pab", action="store_true", help="BASELINE config 3: attention-only PAB (reported as a separate workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vae", action="store_true", help="skip the (untimed-region) VAE decode measurement")
    ap.add_argument("--no-t5", action="store_true", help="skip the (untimed-region) T5 encode measurement")
    ap.add_argument("--text-len", type=int, default=300)
    ap.add_argument("--no-cp", action="store_true",
                    help="N = 2: plain sequence parallelism over both ranks (the reference pipeline's default) instead of the reference's "
                         "enable_cp decomposition (the CFG pair on the two ranks, no exchange inside the blocks)")
    ap.add_argument("--cp", action="store_true",
                    help="even N > 2: force enable_cp (CFG pair on two rank groups x sequence parallelism over N / 2 ranks); not the default "
                         "there: xGMI is point-to-point, and half the group size doubles the bytes every link carries per exchange")
    ap.add_argument("--geometry", default="512x512x64f", choices=["512x512x64f", "720p128f"],
                    help="512x512x64f = BASELINE config 2 (the contract's workload); 720p128f = BASELINE configs[3] geometry "
                         "(1280x720, 128 frames: 273 600 token rows), reported as its own workload, DiT step only")
    ap.add_argument("--gemm-variant", type=int, default=0, help="lab: vsys_tune_gemm_variant id (0 = shipped shape dispatch)")
    ap.add_argument("--flash-variant", type=int, default=0, help="lab: vsys_tune_flash_variant id (0 = shipped)")
    return ap.parse_args()


def self_launch(n: int) -> int:
    """``python bench.py --gpus N`` with N > 1 and no launcher around it: start the N ranks ourselves (one process per GPU under
    ``torch.distributed.run``, rendezvous on 127.0.0.1 at a free port) with this command line, let rank 0's JSON line pass
    through on stdout, and return the launcher's exit code.  The reference's engine spawns its own workers the same way
    (videosys/core/engine/engine.py:40-72)."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ and "LOCAL_RANK" not in os.environ:
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1 and args.gpus == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X devices"
    # dry-run switch for the 1-GPU test box (tests/test_gpu_sp.py): every rank on device 0, gloo instead of RCCL
    one_gpu = os.environ.get("VSYS_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import __graft_entry__ as ge

    if rank == 0:
        ge.build()
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        dist.barrier()

    from videosys_amd import _lib, ops, pab

    if args.gemm_variant:
        assert _lib.load().vsys_tune_gemm_variant(args.gemm_variant) == 0
    if args.flash_variant:
        assert _lib.load().vsys_tune_flash_variant(args.flash_variant) == 0
    from videosys_amd.pipeline_open_sora import OpenSoraPABConfig, get_latent_size
    from videosys_amd.rflow import RFLOW
    from videosys_amd.stdit3 import STDiT3, STDiT3Config, synth_state_dict

    # ---- workload: BASELINE config 2 (config 3 with --pab)
    frames, height, width = (64, 512, 512) if args.geometry == "512x512x64f" else (128, 720, 1280)
    base_geo = args.geometry == "512x512x64f"
    if not base_geo:
        args.no_cpu_baseline = args.no_vae = args.no_t5 = True
    T, Hl, Wl = get_latent_size(frames, height, width)
    cfg = STDiT3Config(depth=args.depth)
    model = STDiT3(cfg, device=dev)
    model.load_state_dict(synth_state_dict(cfg, seed=1234))
    if world > 1:
        # enable_cp (open_sora_transformer_3d.py:466-482): the conditional and the unconditional sample of the CFG pair go to two rank groups and
        # the sequence is split over N / 2 ranks inside each — the same results (tests/test_gpu_sp.py).  Default at N = 2 only, where it
        # removes every exchange inside the blocks (one small all-gather of the output per step: 49.1 vs 54.0-56.5 ms per rank, wire
        # stubbed).  For N >= 4 the group stays all N ranks: a rank's rows spread over N - 1 point-to-point xGMI links instead of N / 2 - 1,
        # i.e. half the bytes per link and exchange (and with the wire stubbed cp 2 x sp 2 is 5 % slower than sp 4 anyway).
        use_cp = world % 2 == 0 and not args.no_cp and (world == 2 or args.cp)
        model.enable_parallel(dp_size=1, sp_size=world, enable_cp=use_cp)
    cp_size = getattr(model.parallel_manager, "cp_size", 1) or 1
    sp_size = getattr(model.parallel_manager, "sp_size", 1) or 1
    if args.pab:
        pab.set_pab_manager(OpenSoraPABConfig(mlp_broadcast=False))
        pab.update_steps(STEPS_PER_VIDEO)
    g = torch.Generator().manual_seed(0)
    z = torch.randn(1, 4, T, Hl, Wl, generator=g).to(torch.bfloat16).float().to(dev)
    L = args.text_len
    y = (torch.randn(1, 1, 300, cfg.caption_channels, generator=g) * 0.1).to(torch.bfloat16)
    mask = torch.zeros(1, 300, dtype=torch.long)
    mask[:, :L] = 1
    y_null = model.y_embedder.y_embedding[None, None].cpu()
    yy = torch.cat([y, y_null.to(y.dtype)], 0).to(dev)
    sched = RFLOW(num_sampling_steps=STEPS_PER_VIDEO, cfg_scale=7.0, use_timestep_transform=True)
    margs = dict(height=torch.tensor([float(height)]), width=torch.tensor([float(width)]),
                 num_frames=torch.tensor([float(frames)]))
    timesteps = sched.prepare_timesteps(1, margs)
    kw = dict(mask=mask, fps=torch.tensor([24.0, 24.0]), height=torch.tensor([float(height)] * 2),
              width=torch.tensor([float(width)] * 2))

    if args.pab:   # the sampler hands its schedule to the model (rflow.py:65, scheduling_rflow_open_sora.py sample()): slabs nobody reads are not kept
        kw["all_timesteps"] = [int(t.to(torch.bfloat16)[0]) for t in timesteps]

    def one_step(i):
        t = timesteps[i % STEPS_PER_VIDEO]
        out = model(torch.cat([z, z], 0), torch.cat([t, t], 0), yy, **kw)
        nxt = timesteps[(i + 1) % STEPS_PER_VIDEO] if (i % STEPS_PER_VIDEO) < STEPS_PER_VIDEO - 1 else torch.zeros(1)
        dt = float(t[0] - nxt[0]) / 1000.0
        ops.cfg_euler_step(z, out, 7.0, dt)

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    barrier()
    dt_s = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist

        tt = torch.tensor([dt_s], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_s = float(tt.item())
    step_s = dt_s / args.steps

    # ---- roofline of the dominant kernel family (instrumented replay; HIP events on the launch stream)
    roof = None
    if rank == 0:
        ev = []
        real_gemm = ops.gemm

        def timed_gemm(x, w, bias=None, **k):
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            s.record()
            r = real_gemm(x, w, bias, **k)
            e.record()
            ev.append((s, e, 2.0 * x.shape[0] * w.shape[0] * x.shape[1], k.get("epilogue", 0)))
            return r

        # the AdaLN-folded forms of the same GEMMs (qkv / fc1 with LayerNorm + modulation in the epilogue: "epilogue" 3 / 4;
        # cross-proj / fc2 that also emit the row statistics: 5)
        real_gemm_ln, real_gemm_stats = ops.gemm_ln, ops.gemm_stats

        def timed_gemm_ln(x, wp, cs, cv, stats, **k):
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            s.record()
            r = real_gemm_ln(x, wp, cs, cv, stats, **k)
            e.record()
            ev.append((s, e, 2.0 * x.shape[0] * wp.shape[0] * x.shape[1], 4 if k.get("gelu") else 3))
            return r

        def timed_gemm_stats(x, w, bias, stats, **k):
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            s.record()
            r = real_gemm_stats(x, w, bias, stats, **k)
            e.record()
            ev.append((s, e, 2.0 * x.shape[0] * w.shape[0] * x.shape[1], 5))
            return r

        real_gemm_gra = ops.gemm_gate_res_add   # (--pab) gate + residual GEMMs that carry the broadcasts behind them: "epilogue" 6

        def timed_gemm_gra(x, w, bias, **k):
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            s.record()
            r = real_gemm_gra(x, w, bias, **k)
            e.record()
            ev.append((s, e, 2.0 * x.shape[0] * w.shape[0] * x.shape[1], 6))
            return r

    barrier()
    nrep = min(args.steps, 3)
    if rank == 0:
        ops.gemm_gate_res_add = timed_gemm_gra
        ops.gemm, ops.gemm_ln, ops.gemm_stats = timed_gemm, timed_gemm_ln, timed_gemm_stats  # every rank replays the steps (DSP collectives); only rank 0 is instrumented
    was_prog = model.use_programs
    model.use_programs = False   # the instrumented replay issues every launch from Python (a recorded launch program would bypass
    try:                         # the event brackets); the TIMED region above ran the product default
        for i in range(nrep):
            one_step(i)
        torch.cuda.synchronize()
    finally:
        model.use_programs = was_prog
        if rank == 0:
            ops.gemm, ops.gemm_ln, ops.gemm_stats = real_gemm, real_gemm_ln, real_gemm_stats
            ops.gemm_gate_res_add = real_gemm_gra
    if rank == 0:
        # drop the once-per-prompt kv_linear launches (none after warm-up) and aggregate
        tot_ms = sum(s.elapsed_time(e) for s, e, _, _ in ev)
        tot_fl = sum(f for _, _, f, _ in ev)
        n = len(ev)
        by_epi = {}
        for s, e, f, epi in ev:
            a = by_epi.setdefault(int(epi), [0.0, 0.0, 0])
            a[0] += s.elapsed_time(e)
            a[1] += f
            a[2] += 1
        achieved = tot_fl / (tot_ms * 1e-3) / 1e12
        # HBM/fabric bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on the same
        # kernels and shapes, tools/gemm_traffic.py -> profiles/r01_gemm_traffic.json), config-2 launch mix
        traffic = None
        tname = next((n for n in ("r06_gemm_traffic.json", "r05_gemm_traffic.json", "r04_gemm_traffic.json", "r03_gemm_traffic.json", "r01_gemm_traffic.json")
                      if os.path.exists(os.path.join(ROOT, "profiles", n))), None)
        if tname is not None and world == 1:
            with open(os.path.join(ROOT, "profiles", tname)) as fh:
                traffic = round(json.load(fh)["avg_bytes_per_launch_config2_mix"])
        roof = {
            "bound": "mfma", "kernel": "vsys::gemm_kernel<EPI, 8, 256, 1, 0, 1> + vsys::gemm2_kernel<EPI, 2, 0, 1> (256x192 tile, bf16 MFMA 16x16x32, shape-dispatched, all epilogues)",
            "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic if base_geo else None,
            "traffic_source": f"profiles/{tname} (committed rocprofv3 PMC passes of the same kernels and shapes with the shipped dispatch; "
                              "NOT re-measured by this run)",
            "annotations_not_measured_by_this_run": {
                "power_limited_mfma_ceiling_tflops": 1816.0, "source": "profiles/r01_mfma_power_ceiling.json",
                "note": "a bare v_mfma_f32_32x32x16_bf16 loop with operands changing every instruction sustains 1816-1830 TFLOP/s at the 1400 W cap "
                        "(2465 with constant operands), the 16x16x32 form these kernels use since round 6 2045-2060 (profiles/r06_kernel_bench_mf16.txt); "
                        "profiles/r02_gemm_yardstick.json: the round-2 family reached 1219 TFLOP/s on random and 1744 on zero operands at "
                        "8192x3072x4096 (vendor library 1307 / 1726)"},
            "traffic_unit": "bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE, fabric side incl. Infinity-Cache hits; algorithmic "
                            "operand + output (+ residual) bytes per launch: 305e6)",
            "launches_per_step": n // nrep, "avg_launch_ms": round(tot_ms / n, 4),
            "algorithmic_gflop_per_launch": round(tot_fl / n / 1e9, 2),
            "gemm_ms_per_step": round(tot_ms / nrep, 2),
            "per_epilogue_tflops": {str(k): round(v[1] / (v[0] * 1e-3) / 1e12, 1) for k, v in by_epi.items()},
        }
    if world > 1:
        barrier()

    # ---- DSP communication (N > 1): HIP-event brackets around every collective of a rank in short replays — all-to-all time
    # per step and rank with the exchanges serialized on the main stream (overlap off), the step time both ways, and the
    # fraction of the serialized communication time the default two-stream overlap hides
    dsp_info = None
    if world > 1:
        import torch.distributed as dist

        from videosys_amd import dsp as vdsp

        def replay(overlap):
            model._overlap = bool(overlap) and model._side is not None
            model.use_programs = False   # eager: the event brackets around the exchanges (collectives or vsys_p2p_exchange launches) must run
            one_step(0)   # settle buffers of this mode
            vdsp.COMM_TIMER.reset()
            vdsp.COMM_TIMER.enabled = True
            barrier()
            t0_ = time.perf_counter()
            for i in range(nrep):
                one_step(i)
            barrier()
            dt_ = (time.perf_counter() - t0_) / nrep
            vdsp.COMM_TIMER.enabled = False
            rep = vdsp.COMM_TIMER.report()
            return dt_, rep["comm_ms"] / nrep, rep["collectives"] // nrep

        was = model._overlap
        seq_par = model._sp is not None   # (N = 2 with enable_cp: sp = 1, the blocks exchange nothing — only the output all-gather is timed)
        if seq_par and model._side is None:
            model._side = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        was_prog2 = model.use_programs
        try:
            t_off, comm_off, ncoll = replay(False)
            t_on, comm_on, _ = replay(True) if seq_par else (t_off, comm_off, ncoll)
        finally:
            model.use_programs = was_prog2
        model._overlap = was
        if seq_par:
            vdsp.check_exchange(model)   # a timed-out peer-to-peer exchange must fail the bench, not shape its number
        mine = torch.tensor([comm_off, comm_on, t_off * 1e3, t_on * 1e3], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        if rank == 0:
            comm_off_all = [round(float(v[0]), 3) for v in allr]
            t_off_ms, t_on_ms = max(float(v[2]) for v in allr), max(float(v[3]) for v in allr)
            hidden = max(0.0, t_off_ms - t_on_ms)
            xi = model._sp.exchange_info if seq_par else {}
            dsp_info = {
                "collectives_per_step": ncoll, "comm_ms_per_step_per_rank_serialized": comm_off_all,
                "comm_ms_per_step_per_rank_overlapped_streams": [round(float(v[1]), 3) for v in allr],
                "step_ms_overlap_off": round(t_off_ms, 3), "step_ms_overlap_on": round(t_on_ms, 3),
                "overlap_fraction": round(min(1.0, hidden / max(max(comm_off_all), 1e-9)), 4),
                "cfg_parallel": cp_size, "sequence_parallel": sp_size,
                "overlap_default": bool(was), "switch_order": model._switch_order(2 // cp_size, T, (Hl // 2) * (Wl // 2)) if seq_par else None,
                # which exchange the group runs on and why: the one-time guarded trial of the one-kernel peer-to-peer exchange at
                # enable_parallel (dsp.p2p_selftest: patterned payload, both paths timed on a config-2-sized message, the faster one wins;
                # any failure -> RCCL on every rank)
                "exchange_path": xi.get("exchange_path", "none: cfg-parallel only (one all-gather of the output per step)"),
                "p2p_selftest": xi.get("selftest"), "p2p_ms_per_exchange": xi.get("p2p_ms"), "rccl_ms_per_exchange": xi.get("rccl_ms"),
                "selftest_message_mb_per_peer": xi.get("message_mb_per_peer"),
                "note": "event brackets include the wait for the slowest peer; overlap_fraction = (step_off - step_on) / serialized comm; these replays issue every launch from Python (eager) so that the brackets run",
            }

    # ---- CPU baseline (rank 0, N == 1 only): oracle = fp32 port of the reference path at the full token count
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.pab:
        try:
            cpu = cpu_baseline(cfg, T, Hl, Wl, L)
        except Exception as e:  # a reported baseline, never a reason to lose the measurement
            cpu = {"error": f"{type(e).__name__}: {e}"[:300]}

    # ---- VAE decode of the final latent (rank 0, N == 1; outside the timed region; row a14): one warm-up + one timed call
    vae = None
    if rank == 0 and world == 1 and not args.no_vae:
        try:
            from videosys_amd.vae_open_sora import OpenSoraVAE, synth_state_dict as vae_synth

            dec = OpenSoraVAE(vae_synth(0), device=dev)
            zb = z[:1].to(torch.bfloat16)
            dec.decode(zb, frames)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            vid = dec.decode(zb, frames)
            torch.cuda.synchronize()
            vae_s = time.perf_counter() - t0
            vae = {"sec_per_video": round(vae_s, 4), "output": list(vid.shape),
                   "videos_per_min_dit_plus_vae": round(60.0 / (STEPS_PER_VIDEO * step_s + vae_s), 4)}
            del dec, vid
        except Exception as e:  # the extra terms must never cost the bench line
            vae = {"error": f"{type(e).__name__}: {e}"[:300]}

    # ---- N > 1: the VAE decode sharded over the ranks by output frame + one all-gather of uint8 frames (outside the timed region;
    # the reference decodes the whole video on every rank).  Every rank takes part; the slowest rank's time is reported.
    if world > 1 and not args.no_vae:
        import torch.distributed as dist

        try:
            from videosys_amd.vae_open_sora import OpenSoraVAE, synth_state_dict as vae_synth

            dec = OpenSoraVAE(vae_synth(0), device=dev)
            zb = z[:1].to(torch.bfloat16)
            dec.decode_sharded(zb, frames, dist.group.WORLD)
            barrier()
            t0 = time.perf_counter()
            vid = dec.decode_sharded(zb, frames, dist.group.WORLD)
            torch.cuda.synchronize()
            tt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            vae_s = float(tt.item())
            vae = {"sec_per_video": round(vae_s, 4), "output": list(vid.shape), "sharded_over_ranks": world,
                   "frames_per_rank": [b_ - a_ for a_, b_ in dec.frame_shards(frames, world)], "gathered": "uint8 [B, F, H, W, 3], one all-gather",
                   "videos_per_min_dit_plus_vae": round(60.0 / (STEPS_PER_VIDEO * step_s + vae_s), 4)}
            del dec, vid
        except Exception as e:
            vae = {"error": f"{type(e).__name__}: {e}"[:300]}
        barrier()

    # ---- T5-v1.1-XXL prompt encode (rank 0, N == 1; outside the timed region): 300-token prompt, device-generated weights
    t5 = None
    if rank == 0 and world == 1 and not args.no_t5:
        try:
            from videosys_amd.t5 import T5Encoder

            enc = T5Encoder(device=dev).init_random_(0)
            ids = torch.randint(0, enc.config.vocab_size, (1, 300), generator=torch.Generator().manual_seed(0))
            tmask = torch.zeros(1, 300, dtype=torch.long)
            tmask[:, :120] = 1
            enc(ids, tmask)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            enc(ids, tmask)
            torch.cuda.synchronize()
            t5 = {"sec_per_prompt": round(time.perf_counter() - t0, 5), "model": "T5-v1.1-XXL encoder geometry, 300 tokens"}
            del enc
            if vae is not None and "sec_per_video" in vae:
                vae["videos_per_min_t5_dit_vae"] = round(60.0 / (STEPS_PER_VIDEO * step_s + vae["sec_per_video"] + t5["sec_per_prompt"]), 4)
        except Exception as e:
            t5 = {"error": f"{type(e).__name__}: {e}"[:300]}

    if rank == 0:
        vpm = 60.0 / (STEPS_PER_VIDEO * step_s)
        line = {
            "metric": (f"videos/min (Open-Sora {'512x512x64f' if base_geo else '1280x720x128f'}, 30 denoise steps, DiT denoising only) "
                       "+ sec/denoise-step"),
            "value": round(vpm, 4), "unit": "videos/min", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_s * 1e3, 3), "sec_per_denoise_step": round(step_s, 5), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {
                "workload": ((f"open-sora-v1.2 STDiT3-XL/2 {'512x512x64f' if base_geo else '1280x720x128f (BASELINE configs[3] geometry)'}, "
                              f"latent [4,{T},{Hl},{Wl}], CFG batch 2 = {2 * T * (Hl // 2) * (Wl // 2)} token rows, ") +
                             f"{L} text tokens, depth {args.depth}" + (", PAB attention-only (config 3)" if args.pab else "")),
                "steps_per_video": STEPS_PER_VIDEO, "parallelism": (f"dsp{world}" if cp_size == 1 else f"cfg-parallel 2 x dsp{sp_size} (the reference's enable_cp; --no-cp: dsp{world})"),
                "not_included": "value is DiT denoising only; vae_decode / t5_encode report the other two terms of the metric beside it",
                "algorithmic_tflop_per_step": 89.4 if args.depth == 28 and L == 300 and base_geo else None,
                "launch_path": ("recorded launch program: " + json.dumps(model.program_stats)) if model.use_programs else "eager (every launch from Python)",
                "dsp_layout": None if world == 1 else {"scatter": model._scatter, "overlap": str(model._overlap), "switch": model._switch},
            },
            "step_tflops": round(89.4 / step_s, 1) if args.depth == 28 and L == 300 and not args.pab and base_geo else None,
            "roofline": roof, "cpu_baseline": cpu, "vae_decode": vae, "t5_encode": t5, "dsp": dsp_info,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(cfg, T, Hl, Wl, L, budget_s=80.0):
    """The reference's CPU PyTorch path on this box's host cores, as SURVEY.md §8(d) prescribes: the oracle (an fp32 port of the
    reference STDiT3, ``kind: "port"``) on the FULL config-2 token count, timed with ``valid_depth`` 1 and 2 (what the reference
    itself honours, open_sora_transformer_3d.py:608), warm-up + 3 repeats each, and fitted as fixed + depth x per-pair — after a
    thread-count sweep over {physical/16 .. physical, logical} cores on a 2-frame slice (a 256-thread pool on a cold process is 200x slower
    than the same cores used well).  The host's fp32 GEMM rate is measured beside it so the figure can be sanity-checked:
    a CPU step cannot beat 89.4 TFLOP / that rate.  Frames are reduced (and said so) only if a full-token pair would not fit
    the time budget."""
    from oracle import stdit3_oracle as O

    t_start = time.perf_counter()
    O.FUSED_SDPA = True   # F.scaled_dot_product_attention at the reference's SDPA call sites, as its own CPU run would
    logical = os.cpu_count() or 1
    try:
        import psutil

        physical = psutil.cpu_count(logical=False) or logical
    except Exception:
        physical = logical
    cands = sorted({max(1, physical // 16), max(1, physical // 8), max(1, physical // 4), max(1, physical // 2), physical, logical})
    c = dict(depth=2, hidden_size=cfg.hidden_size, num_heads=cfg.num_heads, caption_channels=cfg.caption_channels,
             model_max_length=300)
    sd = O.synth_state_dict(**c, seed=1)
    m = O.STDiT3Oracle(sd, 2, cfg.hidden_size, cfg.num_heads)
    kind = "port"
    try:   # where the reference tree exists (the build container; never the GPU box) its OWN STDiT3 class is what gets timed
        from oracle import ref_loader

        if ref_loader.reference_available():
            ref_model = ref_loader.build_reference_stdit3(c, sd)

            class _Ref:    # same call shape as the oracle: forward(x, t, y, valid_depth=, mask=, fps=, height=, width=)
                def forward(self, x, t, y, **kw_):
                    return ref_model(x, t, y, **kw_)

            m, kind = _Ref(), "reference"
    except Exception:
        kind = "port"
    g = torch.Generator().manual_seed(0)
    y = torch.randn(2, 1, 300, cfg.caption_channels, generator=g) * 0.1
    mask = torch.zeros(1, 300, dtype=torch.long)
    mask[:, :L] = 1
    kw = dict(mask=mask, fps=torch.tensor([24.0, 24.0]), height=torch.tensor([512.0, 512.0]), width=torch.tensor([512.0, 512.0]))
    t = torch.tensor([500.0, 500.0])

    def run(x, depth):
        with torch.no_grad():
            t0 = time.perf_counter()
            m.forward(x, t, y, valid_depth=depth, **kw)
            return time.perf_counter() - t0

    # thread-count sweep on a 2-frame slice (the pool gets slower beyond ~physical/4 on this class of host), then the measurement
    # on as many of the 19 frames as the budget allows: 1 warm-up + 2 x (depth 1 + depth 2) = 7 single-pair passes
    probe = torch.randn(2, 4, 2, Hl, Wl, generator=g)
    sweep = {}
    for n in cands:
        torch.set_num_threads(n)
        run(probe, 1)                  # warm-up at this pool size
        sweep[n] = run(probe, 1)
        if sweep[n] > 1.5 * min(sweep.values()) or time.perf_counter() - t_start > budget_s * 0.2:
            break   # past the knee or out of sweep time
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    per_frame = 1.3 * sweep[best] / 2     # (a full-size pass is a little slower per frame than the 2-frame slice)
    left = budget_s * 0.85 - (time.perf_counter() - t_start)
    Ts = max(2, min(T, int(left / (per_frame * 9.5))))
    x = torch.randn(2, 4, Ts, Hl, Wl, generator=g)
    # The 2-frame slice under-feeds a big pool (16 of 128 cores won there in round 5, and the 19-frame measurement then ran at a
    # quarter of the host's GEMM rate): one FULL-SIZE depth-1 pass at the sweep's winner AND at the next larger pool, keep the faster.
    full = {}
    bigger = [n for n in cands if n > best]
    for n in [best] + bigger[:1]:
        torch.set_num_threads(n)
        run(x, 1)                      # warm-up at this pool size and problem size
        full[n] = run(x, 1)
    best = min(full, key=full.get)
    torch.set_num_threads(best)
    reps = 2
    t1 = sorted(run(x, 1) for _ in range(reps))
    t2 = sorted(run(x, 2) for _ in range(reps))
    pair = max(t2[0] - t1[0], 1e-9)
    fixed = max(t1[0] - pair, 0.0)
    scale = T / Ts
    step_s = (fixed + cfg.depth * pair) * scale
    # host fp32 GEMM rate at the same pool size
    a = torch.randn(4096, 4096)
    torch.mm(a, a)
    t0 = time.perf_counter()
    for _ in range(3):
        torch.mm(a, a)
    gemm_tf = 3 * 2 * 4096**3 / (time.perf_counter() - t0) / 1e12
    O.FUSED_SDPA = False
    return {
        "value": round(60.0 / (STEPS_PER_VIDEO * step_s), 6), "unit": "videos/min", "cores": physical, "threads": best,
        "logical_cpus": logical, "kind": kind,
        "sec_per_denoise_step": round(step_s, 2), "fit": {"fixed_s": round(fixed * scale, 3), "per_block_pair_s": round(pair * scale, 3),
                                                            "depth": cfg.depth},
        "valid_depth_1_s": [round(v, 3) for v in t1], "valid_depth_2_s": [round(v, 3) for v in t2],
        "thread_sweep_2_frames_depth1_s": {str(k): round(v, 3) for k, v in sweep.items()},
        "full_size_depth1_s_by_threads": {str(k): round(v, 3) for k, v in full.items()},
        "cpu_step_tflops": round(89.4 / step_s, 3) if cfg.depth == 28 and L == 300 else None,
        "host_fp32_gemm_tflops": round(gemm_tf, 3),
        "lower_bound_s_at_gemm_rate": round(89.4 / gemm_tf, 2),
        "sample_seconds": round(time.perf_counter() - t_start, 1),
        "sample": (("the reference's own STDiT3 class (videosys/models/transformers/open_sora_transformer_3d.py through oracle/ref_loader.py, fp32, CPU)"
                    if kind == "reference" else "CPU oracle (fp32 PyTorch port of the reference STDiT3)") +
                   f" on {Ts} of {T} latent frames x 1024 tokens, CFG batch 2, "
                   f"{L} text tokens: valid_depth 1 and 2, warm-up + {reps} repeats each (minimum taken), fitted fixed + 28 x pair"
                   + ("" if Ts == T else f", scaled x{scale:.2f} in tokens")),
    }


if __name__ == "__main__":
    main()
