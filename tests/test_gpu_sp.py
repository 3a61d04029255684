"""GPU: the model-level sequence-parallel paths end to end on real hardware.  The GPU box has ONE device, so the two ranks
share cuda:0 and talk over gloo (which stages CUDA tensors through the host); everything else — the HIP pack/unpack kernels,
the padded layouts, the per-rank shapes of every kernel launch — is the product path that runs under RCCL on an 8-GPU node.
The sharded result must equal the single-process result bit for bit (the partition is over whole attention problems)."""
import os
import socket
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _stdit3_worker(rank, world, port, outdir, T, HW, p2p=False):
    import traceback

    import torch.distributed as dist

    try:
        if p2p:     # the one-kernel exchange between PROCESSES: tensors and flags mapped through HIP IPC, flags polled on the device
            os.environ["VSYS_DSP_P2P"] = "1"
            os.environ["VSYS_P2P_TIMEOUT_S"] = "5"
        from oracle import stdit3_oracle as O
        from videosys_amd import dsp
        from videosys_amd.stdit3 import STDiT3, STDiT3Config

        torch.cuda.set_device(0)
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
        cfg = dict(depth=2, hidden_size=576, num_heads=8, caption_channels=64, model_max_length=16)
        sd = O.synth_state_dict(**cfg, seed=31)
        sd = {k: (v if k == "rope.freqs" else v.to(torch.bfloat16).float()) for k, v in sd.items()}
        g = torch.Generator().manual_seed(9)
        x = torch.randn(2, 4, T, HW, HW, generator=g).to(torch.bfloat16).float()
        y = torch.randn(2, 1, 16, 64, generator=g).to(torch.bfloat16).float()
        mask = torch.zeros(1, 16, dtype=torch.long)
        mask[:, :11] = 1
        kw = dict(mask=mask, fps=torch.tensor([24.0, 24.0]), height=torch.tensor([float(HW * 8)] * 2),
                  width=torch.tensor([float(HW * 8)] * 2))
        t = torch.tensor([500.0, 500.0])
        model = STDiT3(STDiT3Config(**cfg), device="cuda:0")
        model.load_state_dict(sd)
        ref_full = model(x, t, y, **kw).float().cpu()
        # a rank whose MODULATED ACTIVATIONS travel (the reference's order) computes the spatial qkv site with the separate AdaLN pass
        # and folds every other site; the single-process model does exactly that with fold_spatial_qkv = False
        model.fold_spatial_qkv = False
        ref = model(x, t, y, **kw).float().cpu()
        model.fold_spatial_qkv = True
        assert (ref - ref_full).abs().max().item() <= 2e-2 * ref.abs().max().item()
        model.enable_parallel(1, world, False, overlap=False)   # batched path (no side streams)
        assert not model._overlap
        out = model(x, t, y, **kw).float().cpu()
        out2 = model(x, t, y, **kw).float().cpu()
        # the default: comm/compute overlap — the two CFG samples on two side streams, collectives issued A1, B1, A2, B2
        model.enable_parallel(1, world, False, overlap=True)
        assert model._overlap is True and model._side is not None and model._switch_order(2, T, (HW // 2) ** 2) == "activations"
        out3 = model(x, t, y, **kw).float().cpu()
        out4 = model(x, t, y, **kw).float().cpu()
        torch.cuda.synchronize()
        # the other order of the exchange: qkv GEMM at rest on the un-padded shard, the 3C-wide q|k|v travels
        model._switch = "qkv"
        out6 = model(x, t, y, **kw).float().cpu()
        model._switch = "auto"
        torch.cuda.synchronize()
        assert torch.equal(out6, ref_full), "qkv-first exchange (the folded qkv GEMM runs at rest: the single-GPU arithmetic)"
        p2p_launches = 0
        if model._sp is not None and model._sp.p2p is not None:
            model._sp.p2p.check()          # no exchange timed out waiting for the other process
            p2p_launches = model._sp.p2p.launches
        # enable_cp: the CFG pair split over the two ranks (cp = 2, sp = 1), outputs gathered along the batch
        model.enable_parallel(1, world, True)
        assert model.parallel_manager.cp_size == 2 and model.parallel_manager.sp_size == 1 and model._sp is None
        out5 = model(x, t, y, **kw).float().cpu()
        torch.cuda.synchronize()
        ok = (torch.equal(out, ref) and torch.equal(out, out2) and torch.equal(out3, ref) and torch.equal(out4, ref)
              and torch.equal(out5, ref_full))
        if p2p:
            assert p2p_launches > 0, "VSYS_DSP_P2P=1 did not take the peer-to-peer path"
        err = (out - ref).abs().max().item()
        with open(os.path.join(outdir, f"r{rank}.txt"), "w") as f:
            f.write("ok" if ok else f"mismatch max|diff| {err} of {ref.abs().max().item()}")
    except Exception:
        with open(os.path.join(outdir, f"r{rank}.txt"), "w") as f:
            f.write(traceback.format_exc())
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _stdit3_cp_sp_worker(rank, world, port, outdir, T, HW):
    """enable_cp with sequence parallelism inside each CFG group (world 4 = cp 2 x sp 2): what bench.py runs for an even N."""
    import traceback

    import torch.distributed as dist

    try:
        from oracle import stdit3_oracle as O
        from videosys_amd.stdit3 import STDiT3, STDiT3Config

        torch.cuda.set_device(0)
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
        cfg = dict(depth=2, hidden_size=576, num_heads=8, caption_channels=64, model_max_length=16)
        sd = O.synth_state_dict(**cfg, seed=31)
        sd = {k: (v if k == "rope.freqs" else v.to(torch.bfloat16).float()) for k, v in sd.items()}
        g = torch.Generator().manual_seed(9)
        x = torch.randn(2, 4, T, HW, HW, generator=g).to(torch.bfloat16).float()
        y = torch.randn(2, 1, 16, 64, generator=g).to(torch.bfloat16).float()
        mask = torch.zeros(1, 16, dtype=torch.long)
        mask[:, :11] = 1
        kw = dict(mask=mask, fps=torch.tensor([24.0, 24.0]), height=torch.tensor([float(HW * 8)] * 2),
                  width=torch.tensor([float(HW * 8)] * 2))
        t = torch.tensor([500.0, 500.0])
        model = STDiT3(STDiT3Config(**cfg), device="cuda:0")
        model.load_state_dict(sd)
        model.fold_spatial_qkv = False      # (what a rank whose modulated activations travel computes: see _stdit3_worker)
        ref = model(x, t, y, **kw).float().cpu()
        model.fold_spatial_qkv = True
        outs = []
        for overlap in (False, True):
            model.enable_parallel(1, world, True, overlap=overlap)
            pm = model.parallel_manager
            assert pm.cp_size == 2 and pm.sp_size == world // 2 and model._sp is not None
            outs.append(model(x, t, y, **kw).float().cpu())
            outs.append(model(x, t, y, **kw).float().cpu())     # the replayed launch program
        torch.cuda.synchronize()
        if model._sp.p2p is not None:
            model._sp.p2p.check()
        ok = all(torch.equal(o, ref) for o in outs)
        with open(os.path.join(outdir, f"r{rank}.txt"), "w") as f:
            f.write("ok" if ok else f"mismatch max|diff| {max((o - ref).abs().max().item() for o in outs)} of {ref.abs().max().item()}")
    except Exception:
        with open(os.path.join(outdir, f"r{rank}.txt"), "w") as f:
            f.write(traceback.format_exc())
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _run(worker, args, world=2, timeout=300):
    ctx = mp.get_context("spawn")
    port = _free_port()
    with tempfile.TemporaryDirectory() as d:
        procs = [ctx.Process(target=worker, args=(r, world, port, d) + args) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=timeout)
        for p in procs:
            if p.is_alive():
                p.kill()
        res = []
        for r in range(world):
            fn = os.path.join(d, f"r{r}.txt")
            res.append(open(fn).read() if os.path.exists(fn) else "no result (crashed or timed out)")
    for r, s in enumerate(res):
        assert s == "ok", f"rank {r}: {s}"


def _latte_worker(rank, world, port, outdir, p2p=False):
    import traceback

    import torch.distributed as dist

    try:
        if p2p:
            os.environ["VSYS_DSP_P2P"] = "1"
            os.environ["VSYS_P2P_TIMEOUT_S"] = "5"
        from conftest import load_golden
        from oracle import latte_oracle as LO
        from videosys_amd import pab
        from videosys_amd.latte import LatteT2V

        torch.cuda.set_device(0)
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
        pab.set_pab_manager(None)
        fx = load_golden("latte_fwd_small.pt")
        cfg = fx["cfg"]
        sd = LO.synth_state_dict(cfg["num_layers"], cfg["num_attention_heads"], cfg["attention_head_dim"],
                                 caption_channels=cfg["caption_channels"], seed=fx["seed"])
        sd = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
        m = LatteT2V(**cfg, device="cuda:0")
        m.load_state_dict(sd)
        g = torch.Generator().manual_seed(4)
        res = []
        for frames in (4, 3):   # 3 frames over 2 ranks: one zero-padded frame on the last rank
            x = fx["x"][:, :, :frames].contiguous()
            call = lambda: m(x, timestep=fx["t"], encoder_hidden_states=fx["y"].clone(), encoder_attention_mask=fx["mask"],
                             return_dict=False)[0].float().cpu()
            m._sp = None            # single-process path
            ref = call()
            m.enable_parallel(1, world, False)
            out = call()
            res.append((frames, torch.equal(out, ref), (out - ref).abs().max().item(), ref.abs().max().item()))
            if p2p:
                assert m._sp.p2p is not None and m._sp.p2p.launches > 0, "VSYS_DSP_P2P=1 did not take the peer-to-peer path"
                m._sp.p2p.check()
        torch.cuda.synchronize()
        bad = [r for r in res if not r[1]]
        with open(os.path.join(outdir, f"r{rank}.txt"), "w") as f:
            f.write("ok" if not bad else f"mismatch (frames, equal, max|diff|, max|ref|): {bad}")
    except Exception:
        with open(os.path.join(outdir, f"r{rank}.txt"), "w") as f:
            f.write(traceback.format_exc())
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_latte_sp_two_ranks_equals_single():
    """Latte frame-sharded sequence parallelism (latte_transformer_3d.py:826-843,1300-1308,1428-1429,1469-1482)."""
    _run(_latte_worker, ())


def _cogvideox_worker(rank, world, port, outdir, p2p=False):
    import traceback

    import torch.distributed as dist

    try:
        if p2p:
            os.environ["VSYS_DSP_P2P"] = "1"
            os.environ["VSYS_P2P_TIMEOUT_S"] = "5"
        from conftest import load_golden
        from oracle import cogvideox_oracle as CO
        from videosys_amd import pab
        from videosys_amd.cogvideox import CogVideoXTransformer3DModel

        torch.cuda.set_device(0)
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
        pab.set_pab_manager(None)
        fx = load_golden("cogvideox_fwd_small.pt")
        res = []
        for key in ("sincos", "rope"):
            cfg = dict(fx[key]["cfg"], num_attention_heads=6)   # 6 heads: 3 per rank
            sd = {k: v.to(torch.bfloat16).float() for k, v in CO.synth_state_dict(
                cfg["num_layers"], cfg["num_attention_heads"], text_embed_dim=cfg["text_embed_dim"], seed=fx["seed"]).items()}
            m = CogVideoXTransformer3DModel(**cfg, device="cuda:0")
            m.load_state_dict(sd)
            if key == "rope":
                # 3 frames x (3 x 5) patches = 45 video tokens: 23 + 22 over two ranks, one zero-padded row on the last
                x = fx["x"][:, :, :, :6, :10].contiguous()
                rope = CO.rope_3d(64, CO.crop_region((3, 5), 45, 30), (3, 5), 3)
            else:
                x, rope = fx["x"], None
            call = lambda: m(x, fx["y"], fx["t"], image_rotary_emb=rope, return_dict=False)[0].float().cpu()
            ref = call()
            m.enable_parallel(1, world, False)
            out = call()
            res.append((key, torch.equal(out, ref), (out - ref).abs().max().item(), ref.abs().max().item()))
            if p2p:
                assert m._sp.p2p is not None and m._sp.p2p.launches > 0, "VSYS_DSP_P2P=1 did not take the peer-to-peer path"
                m._sp.p2p.check()
        torch.cuda.synchronize()
        bad = [r for r in res if not r[1]]
        with open(os.path.join(outdir, f"r{rank}.txt"), "w") as f:
            f.write("ok" if not bad else f"mismatch (scheme, equal, max|diff|, max|ref|): {bad}")
    except Exception:
        with open(os.path.join(outdir, f"r{rank}.txt"), "w") as f:
            f.write(traceback.format_exc())
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_cogvideox_ulysses_two_ranks_equals_single():
    """CogVideoX Ulysses (cogvideox_transformer_3d.py:45-86,112-123,160-165,530-532,569-570): heads split 2 ways, the video rows
    sharded at rest with the text rows replicated."""
    _run(_cogvideox_worker, ())


@pytest.mark.parametrize("T,HW", [(5, 16), (4, 12), (1, 12)])
def test_stdit3_dsp_two_ranks_equals_single(T, HW):
    """Open-Sora DSP (open_sora_transformer_3d.py:288-315,598-619): T = 5 needs the temporal zero-pad (5 -> 6 over 2 ranks),
    HW = 12 -> S = 36 tokens per frame split 18 / 18; T = 1 is the image case (the CFG batch is scattered instead of the frame)."""
    _run(_stdit3_worker, (T, HW))


@pytest.mark.parametrize("T,HW", [(5, 16), (4, 12)])
def test_stdit3_cfg_parallel_times_dsp_four_processes_equals_single(T, HW):
    """enable_cp (open_sora_transformer_3d.py:466-482,545-557,621) on four processes: the CFG pair on two rank groups, DSP over two
    ranks inside each, outputs gathered along the batch — bit-identical to the single-process model, eager and replayed."""
    _run(_stdit3_cp_sp_worker, (T, HW), world=4, timeout=600)


def test_latte_and_cogvideox_two_processes_peer_to_peer_over_ipc():
    """The same one-kernel exchange under Latte's frame-sharded sequence parallelism (the two switches around every temporal block,
    latte_transformer_3d.py:826-843,1300-1308) and CogVideoX's Ulysses exchange (heads <-> sequence, two problems per peer on the way
    back, cogvideox_transformer_3d.py:45-86,112-165), two processes over HIP IPC: bit-identical to the single-process outputs."""
    _run(_latte_worker, (True,))
    _run(_cogvideox_worker, (True,))


@pytest.mark.parametrize("T,HW", [(5, 16), (4, 12)])
def test_stdit3_dsp_two_processes_peer_to_peer_over_ipc(T, HW):
    """The one-kernel peer-to-peer exchange between two PROCESSES sharing the test box's GPU (VSYS_DSP_P2P=1): destination tensors
    mapped through torch's CUDA IPC, flag arrays through vsys_p2p_ipc_export / _open, sequence flags polled inside the kernel —
    the path a rank per GPU takes over xGMI.  Same assertions as the RCCL-shaped test above: every layout bit-identical to the
    single-process output (comm.py:104-141,282-304; open_sora_transformer_3d.py:288-315)."""
    _run(_stdit3_worker, (T, HW, True))


def test_cogvideox_pab_ulysses_four_ranks_in_process_and_tiled_decode():
    """The combination BASELINE configs[4] names — CogVideoX + PAB (spatial broadcast) + DSP degree 4 + the 3-D VAE's tiled decode —
    in one run on a small geometry: four Ulysses ranks as threads of this process (tools/local_group; 12 heads = 3 per rank, 48 video
    tokens = 12 per rank, text rows replicated), the PAB schedule of the reference fixture (broadcast steps skip the attention AND its
    two exchanges on every rank alike), every step's output of every rank BIT-identical to the single-process run; the ranks then
    decode the final latents with the tiled decode, its 9 tiles shared out over the four ranks and gathered once, and get the bits
    of the unsharded decode of the single-process latents
    (cogvideox_transformer_3d.py:45-86,112-165; pipeline_cogvideox.py:33-44; autoencoder_kl_cogvideox.py:1161-1239)."""
    from conftest import load_golden
    from oracle import cogvideox_oracle as CO
    from tools.local_group import LocalWorld
    from videosys_amd import pab
    from videosys_amd.cogvideox import CogVideoXTransformer3DModel
    from videosys_amd.vae_cogvideox import CogVideoXVAE, synth_state_dict as vae_synth

    fx = load_golden("cogvideox_pab_small.pt")
    cfg = dict(fx["cfg"], num_attention_heads=12)     # hidden 768 = 4 GEMM column tiles; 3 heads per rank
    sd = {k: v.to(torch.bfloat16).float() for k, v in CO.synth_state_dict(cfg["num_layers"], 12, text_embed_dim=cfg["text_embed_dim"],
                                                                          seed=fx["seed"]).items()}
    rope = CO.rope_3d(64, CO.crop_region((4, 6), 45, 30), (4, 6), 3)
    P = 4

    def run_schedule(m):
        m.reset_pab_state()
        return [m(fx["x"], fx["y"], torch.tensor([t, t]), image_rotary_emb=rope, return_dict=False)[0].float().cpu()
                for t in fx["timesteps"]]

    pab.set_pab_manager(pab.PABConfig(spatial_broadcast=True, **fx["pab"]))
    try:
        pab.update_steps(fx["steps"])
        single = CogVideoXTransformer3DModel(**cfg, device="cuda:0")
        single.load_state_dict(sd)
        want = run_schedule(single)
        assert any(not torch.equal(a, b) for a, b in zip(want, want[1:]))

        def rank_fn(r, group):
            torch.cuda.set_device(0)
            m = CogVideoXTransformer3DModel(**cfg, device="cuda:0")
            m.load_state_dict(sd)
            m.enable_parallel(parallel_mgr=_rank_manager(group, P, r))
            assert m._sp is not None and m._sp.P == P and m._sp.p2p is not None   # in process: the one-kernel Ulysses exchange
            outs = run_schedule(m)
            torch.cuda.synchronize()
            assert m._sp.p2p.launches > 0
            m._sp.p2p.check()
            # the tiled decode with its 9 tiles shared out over the four ranks (tiles r, r + 4, ...) and gathered once
            vae_r = CogVideoXVAE(vae_sd, device="cuda:0", sample_height=64, sample_width=96, use_tiling=True)
            px_r = vae_r.decode_latents(outs[-1][:1].to(torch.bfloat16).to("cuda:0"), group=group)
            torch.cuda.synchronize()
            return outs, px_r.cpu()

        vae_sd = vae_synth(13)
        both = LocalWorld(P, timeout=300).run(rank_fn)
        per_rank, px_ranks = [b[0] for b in both], [b[1] for b in both]
    finally:
        pab.set_pab_manager(None)
    for r, outs in enumerate(per_rank):
        for i, (o, w) in enumerate(zip(outs, want)):
            assert torch.equal(o, w), f"rank {r}, step {i} (t = {fx['timesteps'][i]}): max|diff| {float((o - w).abs().max()):.3e}"
    # rank 0 decodes what it sampled: [B, F, C, H, W] latents -> pixels through the tiled decode (tile = half the sample size)
    vae = CogVideoXVAE(vae_sd, device="cuda:0", sample_height=64, sample_width=96, use_tiling=True)
    lat = per_rank[0][-1][:1].to(torch.bfloat16).to("cuda:0")
    assert lat.shape[-2] > vae.tile_latent_min_height and lat.shape[-1] > vae.tile_latent_min_width      # the decode really tiles
    px = vae.decode_latents(lat)
    px_single = vae.decode_latents(want[-1][:1].to(torch.bfloat16).to("cuda:0"))
    # (the tile arithmetic of tiled_decode, :1180-1239: 3 x 3 tiles of 4 x 6 latent rows / columns stepping 3 / 4, each cropped to 27 x 39
    #  pixels, the last to what it has — 27 + 27 + 16 by 39 + 39 + 32; at the real 60 x 90 latent the same rule gives 480 x 720)
    assert px.shape[0] == 1 and px.shape[1] == 3 and tuple(px.shape[-2:]) == (70, 110) and torch.isfinite(px.float()).all()
    assert torch.equal(px, px_single)
    for r, pr in enumerate(px_ranks):          # every rank's tile-sharded decode: the same pixels
        assert torch.equal(pr, px.cpu()), f"rank {r}: tile-sharded decode differs"


@pytest.mark.parametrize("nproc,cp", [(2, False), (2, True), (4, True), (4, False)])
def test_bench_two_ranks_dry_run(nproc, cp):
    """bench.py's N > 1 path (rank-0 build, barriers, DSP model, max-over-ranks timing, one JSON line from rank 0), launched the
    way the driver launches it, with every rank on the one GPU of the test box over gloo (VSYS_BENCH_ONE_GPU=1), depth 2.
    cp = the reference's enable_cp (the CFG pair on two rank groups, sequence parallelism over N / 2 ranks inside each): the default
    at N = 2 (no exchange inside the blocks; --no-cp = plain DSP), forced with --cp at N = 4 (default there: DSP over all N ranks)."""
    import json
    import subprocess
    import sys

    from conftest import ROOT

    env = dict(os.environ, VSYS_BENCH_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "2", "--warmup", "1",
           "--depth", "2", "--no-cpu-baseline", "--no-vae", "--no-t5"] + (["--cp"] if cp and nproc > 2 else ([] if cp or nproc > 2 else ["--no-cp"]))
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == nproc and j["steps"] == 2 and j["value"] > 0
    assert j["roofline"]["frac"] > 0
    d = j["dsp"]   # per-rank collective time and the overlap report the first multi-GPU run is meant to deliver
    sp = nproc // 2 if cp else nproc
    assert j["config"]["parallelism"].startswith(f"cfg-parallel 2 x dsp{sp}" if cp else f"dsp{nproc}"), j["config"]["parallelism"]
    assert d["cfg_parallel"] == (2 if cp else 1) and d["sequence_parallel"] == sp
    assert len(d["comm_ms_per_step_per_rank_serialized"]) == nproc and d["step_ms_overlap_off"] > 0 and d["step_ms_overlap_on"] > 0
    if sp == 1:   # the blocks exchange nothing: one all-gather of the output per step
        assert d["collectives_per_step"] == 1 and d["exchange_path"].startswith("none") and d["switch_order"] is None, d
        return
    assert d["collectives_per_step"] >= 2 * 2 + 1   # 2 spatial blocks x 2 exchanges + the final gather
    assert d["overlap_default"] is True and d["switch_order"] == "activations" and 0.0 <= d["overlap_fraction"] <= 1.0
    # the guarded trial of the one-kernel peer-to-peer exchange ran at enable_parallel and its verdict + both timings are reported
    assert d["p2p_selftest"] == "pass" and d["exchange_path"] in ("p2p", "rccl"), d
    assert d["p2p_ms_per_exchange"] > 0 and d["rccl_ms_per_exchange"] > 0 and d["selftest_message_mb_per_peer"] > 1.0


def _selftest_worker(rank, world, port, outdir, fault):
    import traceback

    import torch.distributed as dist

    try:
        if fault:
            os.environ["VSYS_P2P_SELFTEST_FAULT"] = fault
        os.environ["VSYS_P2P_SELFTEST_TIMEOUT_S"] = "2"
        from videosys_amd import dsp

        torch.cuda.set_device(0)
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
        info = dsp.p2p_selftest(dist.group.WORLD, rows=96, C=576, rounds=3, timing_iters=5)
        sp = dsp.SequenceParallel(dist.group.WORLD)          # VSYS_DSP_P2P unset = auto: the trial decides
        if fault:
            ok = info["selftest"].startswith("fail") and info["exchange_path"] == "rccl" and sp.p2p is None \
                and sp.exchange_info["selftest"].startswith("fail")
            if fault == "mismatch":
                ok = ok and "differ from what it sent" in info["selftest"]
            else:
                ok = ok and ("never received" in info["selftest"] or "differ" in info["selftest"])
        else:
            ok = info["selftest"] == "pass" and info["p2p_ms"] > 0 and info["rccl_ms"] > 0 and info["exchange_path"] in ("p2p", "rccl") \
                and (sp.p2p is not None) == (sp.exchange_info["exchange_path"] == "p2p")
            if sp.p2p is not None:                           # and the chosen path really exchanges: a [B, T, Sl, C] switch and back
                x = torch.arange(2 * 3 * 4 * 64, dtype=torch.float32, device="cuda:0").reshape(2, 3, 4, 64).to(torch.bfloat16) + rank
                out = torch.zeros(2, 2, 8, 64, dtype=torch.bfloat16, device="cuda:0")
                sp.to_temporal_shard(x, 8, out=out)
                torch.cuda.synchronize()
                sp.p2p.check()
                ok = ok and sp.p2p.launches == 1
        with open(os.path.join(outdir, f"r{rank}.txt"), "w") as f:
            f.write("ok" if ok else f"unexpected: {info} / {sp.exchange_info}")
    except Exception:
        with open(os.path.join(outdir, f"r{rank}.txt"), "w") as f:
            f.write(traceback.format_exc())
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("fault", ["", "mismatch", "timeout"])
def test_p2p_selftest_two_processes_pass_and_forced_fail(fault):
    """dsp.p2p_selftest over two PROCESSES (HIP IPC mappings, flags polled on the device; both on the box's one GPU): it passes,
    times both paths and picks one; with a forced payload mismatch on rank 0 or a skipped exchange on rank 1 (the peer's flag never
    arrives: the kernel's wall-clock bound ends the wait) EVERY rank falls back to the RCCL path and says why — nothing hangs."""
    _run(_selftest_worker, (fault,), world=2, timeout=240)


def test_bench_plain_form_launches_its_own_ranks():
    """``python bench.py --gpus 2`` WITHOUT torchrun around it (how the driver launches the N = 1 bench): bench.py starts its own
    ranks and still prints exactly one JSON line (engine.py:40-72 spawns the reference's workers the same way)."""
    import json
    import subprocess
    import sys

    from conftest import ROOT

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["VSYS_BENCH_ONE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--depth", "2",
           "--no-cpu-baseline", "--no-t5"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["config"]["parallelism"].startswith("cfg-parallel 2 x dsp1")
    v = j["vae_decode"]      # N > 1: the decode is sharded by output frame and gathered once
    assert v.get("sharded_over_ranks") == 2 and v["frames_per_rank"] == [32, 32] and v["output"] == [1, 64, 512, 512, 3], v


def test_batched_copy_executor_matches_plan_semantics():
    """hip_copy_executor (one vsys_copy_4d_batch launch per plan) against the element-wise definition of a copy plan, on the
    pack / unpack plans of a 4-way DSP switch and of the Ulysses exchange (padding on the last shard in both)."""
    from test_host_cpu import torch_copy_executor
    from videosys_amd import dsp

    g = torch.Generator().manual_seed(2)
    B, T, S, C, P = 2, 5, 10, 16, 4
    Sl = -(-S // P)
    pack, unpack, sshape, oshape = dsp.plan_switch_to_temporal_shard(B, T, Sl, S, C, P)
    x = torch.randn(B, T, Sl, C, generator=g).to(torch.bfloat16)
    for plan, src_shape, dst_shape in ((pack, (B, T, Sl, C), sshape), (unpack, sshape, oshape)):
        src = torch.randn(src_shape, generator=g).to(torch.bfloat16)
        want = torch.full(dst_shape, 7.0, dtype=torch.bfloat16)
        torch_copy_executor(src, want, plan)
        got = torch.full(dst_shape, 7.0, dtype=torch.bfloat16, device="cuda:0")
        dsp.hip_copy_executor(src.to("cuda:0"), got, plan)
        assert torch.equal(got.cpu(), want)
    Lt, Lv, Cc = 3, 9, 64
    Lvl = -(-Lv // P)
    pk, ul, ur, sshape, oshape = dsp.plan_heads_scatter(B, Lt, Lvl, Lv, Cc, P, 1)
    src = torch.randn(B, Lt + Lvl, 3 * Cc, generator=g).to(torch.bfloat16)
    want = torch.zeros(sshape, dtype=torch.bfloat16)
    torch_copy_executor(src, want, pk)
    got = torch.zeros(sshape, dtype=torch.bfloat16, device="cuda:0")
    dsp.hip_copy_executor(src.to("cuda:0"), got, pk)
    assert torch.equal(got.cpu(), want)


# ------------------------------------------------------------------------------------------------ P = 8 on one GPU, in process
def _rank_manager(group, P, r):
    from types import SimpleNamespace

    return SimpleNamespace(sp_size=P, cp_size=1, dp_size=1, dp_rank=0, sp_rank=r, cp_rank=0, sp_group=group, cp_group=None)


@pytest.mark.parametrize("frames,hl,wl", [(128, 90, 160), (64, 64, 64)])
def test_dsp_eight_ranks_in_process_equals_single_and_oracle(frames, hl, wl):
    """BASELINE configs[3] (Open-Sora 720p x 128f, DSP degree 8) and configs[1] geometry sharded 8 ways, ALL EIGHT RANKS on the one
    GPU of the box (tools/local_group.py: a rank = a thread, a collective = event-ordered device copies): per rank the real padded
    T-shard shapes (T = 38 -> 40 frames: 5 per sample per rank, S = 3600 / 8 = 450 at rest; T = 19, S = 128), the HIP pack / unpack
    launches, both side streams and the A1 B1 A2 B2 collective order.  Every layout — scatter "sample" (the reference's) and
    "flat", exchange order "activations" and "qkv", overlap on and off — must reproduce the single-process output BIT FOR BIT on
    every rank's gathered result; that output is then held against the fp32 oracle at the floor tolerance.  Two block pairs."""
    import fulldepth_util as U
    from oracle import stdit3_oracle as O
    from tools.local_group import LocalWorld
    from videosys_amd import pab
    from videosys_amd.pipeline_open_sora import get_latent_size
    from videosys_amd.stdit3 import STDiT3, STDiT3Config

    pab.set_pab_manager(None)
    P, depth = 8, 2
    T, Hl, Wl = get_latent_size(frames, hl * 8, wl * 8)
    assert (Hl, Wl) == (hl, wl)
    sd = U.bf16_round(O.synth_state_dict(depth, 1152, 16, seed=4321))
    g = torch.Generator().manual_seed(5)
    z = torch.randn(1, 4, T, Hl, Wl, generator=g).to(torch.bfloat16).float()
    y = (torch.randn(1, 1, 300, 4096, generator=g) * 0.1).to(torch.bfloat16).float()
    mask = torch.ones(1, 300, dtype=torch.long)
    x = torch.cat([z, z], 0)
    yy = torch.cat([y, sd["y_embedder.y_embedding"][None, None]], 0)
    t = torch.tensor([600.0, 600.0]).to(torch.bfloat16).float()
    kw = dict(mask=mask, fps=torch.tensor([24.0, 24.0]), height=torch.tensor([hl * 8.0] * 2), width=torch.tensor([wl * 8.0] * 2))

    single = STDiT3(STDiT3Config(depth=depth), device="cuda:0")
    single.load_state_dict(sd)
    out_single = single(x, t, yy, **kw)
    # (AdaLN fold) ranks whose modulated activations travel keep the separate AdaLN pass at the spatial qkv site and fold the rest
    single.fold_spatial_qkv = False
    out_single_act = single(x, t, yy, **kw)
    single.fold_spatial_qkv = True
    torch.cuda.synchronize()

    # (scatter, what travels, overlap, one-kernel peer-to-peer exchange or pack + all_to_all_single + unpack)
    variants = [("flat", "activations", True, True), ("flat", "activations", False, True), ("sample", "activations", True, True),
                ("sample", "qkv", False, True), ("flat", "qkv", False, True),
                ("flat", "activations", True, False), ("sample", "activations", False, False)]
    if frames == 128:      # (273 600 token rows per variant: the large geometry runs one variant of every axis, the small one all seven)
        variants = [variants[0], variants[4], variants[5], variants[6]]
    world = LocalWorld(P, timeout=300)

    def rank_fn(r, group):
        torch.cuda.set_device(0)
        m = STDiT3(STDiT3Config(depth=depth), device="cuda:0")
        m.load_state_dict(sd)
        res = []
        for scatter, order, overlap, p2p in variants:
            m.enable_parallel(parallel_mgr=_rank_manager(group, P, r), overlap=overlap)
            m._scatter, m._switch = scatter, order
            assert m._overlap == overlap
            assert m._sp.p2p is not None      # in-process groups run the one-kernel exchange (vsys_p2p_exchange) by default
            if not p2p:
                m._sp.p2p = None
            before = 0 if m._sp.p2p is None else m._sp.p2p.launches
            out = m(x, t, yy, **kw)        # recorded (launch program, program.py) ...
            out2 = m(x, t, yy, **kw)       # ... and replayed: collectives and cross-stream events re-issued from the log
            torch.cuda.synchronize()
            assert m.program_stats["replayed"] >= 1
            if p2p:
                assert m._sp.p2p.launches > before
                m._sp.p2p.check()
            want = out_single if order == "qkv" else out_single_act
            res.append(bool(torch.equal(out, want)) and bool(torch.equal(out2, want)))
        S_full = (Hl // 2) * (Wl // 2)
        nfr = {v: __import__("videosys_amd.dsp", fromlist=["x"]).frames_per_rank(2, T, P, v) for v in ("flat", "sample")}
        return res, nfr, m._switch_order(1, 2 * T, S_full)

    results = world.run(rank_fn)
    for r, (res, nfr, _) in enumerate(results):
        bad = [v for v, ok in zip(variants, res) if not ok]
        assert not bad, f"rank {r}: sharded output differs from the single-process output for {bad}"
    print(f"\n[dsp x8 in process] T={T} S={(Hl // 2) * (Wl // 2)}: frames on the busiest rank {results[0][1]}, auto order {results[0][2]}")

    ref = O.STDiT3Oracle(sd, depth, 1152, 16, device="cuda:0", dtype=torch.float32)
    floor = O.STDiT3Oracle(sd, depth, 1152, 16, device="cuda:0", dtype=torch.bfloat16)
    out_ref = ref.forward(x, t, yy, **kw)
    out_floor = floor.forward(x, t, yy, **kw)
    r = dict(out_hip=U.stats(out_single, out_ref), out_floor=U.stats(out_floor, out_ref))
    why = U.verdict(r["out_hip"], r["out_floor"])
    assert not why, f"sharded geometry vs fp32 oracle: {why}"
    del single, ref, floor
    torch.cuda.empty_cache()
