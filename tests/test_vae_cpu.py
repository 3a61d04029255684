"""CPU: the VAE-decode oracle (oracle/vae_oracle.py) against the golden minted from the reference's VideoAutoencoderPipeline
(oracle/make_golden_vae.py), the decode-side parameter inventory against the reference's own state_dict (when
/root/reference is present), and the host-side size helpers against the values SURVEY.md §8d derives from the reference."""
import os

import pytest
import torch

from conftest import load_golden


@pytest.fixture(scope="module")
def gold():
    return load_golden("opensora_vae_small.pt")


def test_vae_oracle_matches_reference_golden(gold):
    from oracle import vae_oracle as VO
    from oracle.make_golden import sd_checksum
    from videosys_amd.vae_open_sora import synth_state_dict

    sd = synth_state_dict(gold["seed"])
    assert sd_checksum(sd) == gold["sd_checksum"]
    out = VO.decode(sd, gold["z"], gold["num_frames"])
    ref = gold["out_fp32"].float()
    assert out.shape == ref.shape == (1, 3, 22, 96, 64)
    # the fixture stores the fp32 reference rounded to fp16 (|x| < 4: 2^-10 absolute)
    assert (out - ref).abs().max().item() < 3e-3
    xz = VO.temporal_decode(sd, gold["z"][:, :, :5] * torch.tensor(VO.SCALE)[None, :, None, None, None]
                            + torch.tensor(VO.SHIFT)[None, :, None, None, None], 17)
    assert xz.shape == gold["x_z_first"].shape == (1, 4, 17, 12, 8)
    assert (xz - gold["x_z_first"]).abs().max().item() < 1e-4


def test_decoder_param_inventory_matches_reference():
    from oracle import ref_loader

    if not ref_loader.reference_available():
        pytest.skip("reference tree not present on this box")
    from videosys_amd.vae_open_sora import decoder_param_shapes

    model = ref_loader.build_reference_opensora_vae()
    ref = {k: tuple(v.shape) for k, v in model.state_dict().items()
           if "encoder" not in k and ".quant_conv" not in k and k not in ("scale", "shift")}
    assert decoder_param_shapes() == ref


def test_latent_size_helpers():
    """get_latent_size of VideoAutoencoderPipeline (autoencoder_kl_open_sora.py:704-716): 64 x 512 x 512 -> [19, 64, 64];
    128 x 720 x 1280 -> [38, 90, 160]."""
    from videosys_amd.vae_open_sora import OpenSoraVAE

    v = OpenSoraVAE.__new__(OpenSoraVAE)
    v.micro_frame_size = 17
    assert v.get_latent_size((64, 512, 512)) == [19, 64, 64]
    assert v.get_latent_size((128, 720, 1280)) == [38, 90, 160]
    assert v.get_latent_size((17, 256, 256)) == [5, 32, 32]
    assert v.get_latent_size((1, 512, 512)) == [1, 64, 64]


def test_vae_needs_gpu():
    from videosys_amd.vae_open_sora import OpenSoraVAE

    with pytest.raises(RuntimeError):
        OpenSoraVAE({}, device="cpu")


def test_vae_oracle_matches_live_reference_class():
    """oracle/vae_oracle.py against the reference's VideoAutoencoderPipeline run here (another seed, another latent size, three
    micro-batches incl. a 1-latent-frame tail) when /root/reference is present."""
    from oracle import ref_loader

    if not ref_loader.reference_available():
        pytest.skip("reference tree not present on this box")
    from oracle import vae_oracle as VO
    from videosys_amd.vae_open_sora import synth_state_dict

    sd = synth_state_dict(21)
    model = ref_loader.build_reference_opensora_vae(sd)
    g = torch.Generator().manual_seed(4)
    z = torch.randn(1, 4, 11, 4, 6, generator=g)       # 5 + 5 + 1 latent frames -> 17 + 17 + 4 = 38 frames
    with torch.no_grad():
        ref = model.decode(z, num_frames=38)
    out = VO.decode(sd, z, 38)
    assert out.shape == ref.shape == (1, 3, 38, 32, 48)
    assert (out - ref).abs().max().item() < 1e-4


# ---------------------------------------------------------------------------------------------------- encode side
def test_vae_encode_oracle_matches_reference_golden():
    """oracle/vae_oracle.py encode against the fixture minted from the reference's VideoAutoencoderPipeline.encode
    (oracle/make_golden_vae_encode.py): the two encoders' distribution parameters and the sampled, normalised latents (same
    seeded noise stream)."""
    from oracle import vae_oracle as VO
    from oracle.make_golden import sd_checksum
    from videosys_amd.vae_open_sora import synth_state_dict

    gold = load_golden("opensora_vae_encode_small.pt")
    sd = synth_state_dict(gold["seed"], encoder=True)
    assert sd_checksum(sd) == gold["sd_checksum"]
    fr = gold["x"][0].permute(1, 0, 2, 3)
    assert (VO.spatial_encode_moments(sd, fr[:4]) - gold["frames4_moments"]).abs().max().item() < 1e-4
    assert (VO.temporal_encode_moments(sd, gold["xz17"]) - gold["xz17_moments"]).abs().max().item() < 1e-4
    torch.manual_seed(gold["noise_seed"])
    z = VO.encode(sd, gold["x"])
    assert z.shape == gold["z_fp32"].shape == (1, 4, 6, 4, 6)
    assert (z - gold["z_fp32"]).abs().max().item() < 1e-4


def test_encoder_param_inventory_matches_reference():
    from oracle import ref_loader

    if not ref_loader.reference_available():
        pytest.skip("reference tree not present on this box")
    from videosys_amd.vae_open_sora import decoder_param_shapes, encoder_param_shapes, synth_state_dict

    model = ref_loader.build_reference_opensora_vae()
    ref = {k: tuple(v.shape) for k, v in model.state_dict().items() if "encoder" in k or ".quant_conv" in k}
    assert encoder_param_shapes() == ref
    a, b = synth_state_dict(5), synth_state_dict(5, encoder=True)
    assert set(b) == set(decoder_param_shapes()) | set(encoder_param_shapes())
    assert all(torch.equal(a[k], b[k]) for k in a)          # decode-side weights do not depend on the flag


def test_vae_encode_oracle_matches_live_reference_class():
    from oracle import ref_loader

    if not ref_loader.reference_available():
        pytest.skip("reference tree not present on this box")
    from oracle import vae_oracle as VO
    from videosys_amd.vae_open_sora import synth_state_dict

    sd = synth_state_dict(21, encoder=True)
    model = ref_loader.build_reference_opensora_vae(sd)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 5, 16, 24, generator=g).clamp(-1, 1)     # two samples, one 5-frame micro batch (3 pad frames)
    torch.manual_seed(99)
    with torch.no_grad():
        ref = model.encode(x)
    torch.manual_seed(99)
    out = VO.encode(sd, x)
    assert out.shape == ref.shape == (2, 4, 2, 2, 3)
    assert (out - ref).abs().max().item() < 1e-4


def test_vae_encode_host_flow_on_emulated_kernels():
    """The host composition of OpenSoraVAE.encode — grids, zero borders, the stride-1-conv-then-sample form of the five strided
    convolutions, the padded 8-channel heads, micro batching, noise order — run on CPU with every kernel wrapper replaced by a
    torch restatement of its contract (tests/vae_cpu_emul.py), against the oracle: the moments of both encoders at bf16-storage
    accuracy, the sampled latents on the same seeded noise."""
    from oracle import vae_oracle as VO
    from vae_cpu_emul import cpu_vae, emulated_vae_ops
    from videosys_amd.vae_open_sora import synth_state_dict

    gold = load_golden("opensora_vae_encode_small.pt")
    sd = synth_state_dict(gold["seed"], encoder=True)
    rel = lambda a, b: ((a.float() - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
    with emulated_vae_ops():
        vae = cpu_vae(sd)
        fr = gold["x"][0].permute(1, 0, 2, 3)[:4]
        m2d = vae._spatial_encode(fr.permute(1, 0, 2, 3).to(torch.bfloat16).contiguous()).float().permute(1, 0, 2, 3)
        assert rel(m2d, gold["frames4_moments"]) <= 1.5 * rel(gold["frames4_moments_bf16"], gold["frames4_moments"]) + 1e-3
        mt = vae._temporal_encode(gold["xz17"][0].to(torch.bfloat16).contiguous()).float()[None]
        assert rel(mt, gold["xz17_moments"]) <= 1.5 * rel(gold["xz17_moments_bf16"], gold["xz17_moments"]) + 1e-3
        torch.manual_seed(gold["noise_seed"])
        z = vae._encode(gold["x"], lambda shape: torch.randn(shape))
    assert z.shape == gold["z_fp32"].shape
    assert rel(z, gold["z_fp32"]) <= 0.06, rel(z, gold["z_fp32"])
    cos = torch.nn.functional.cosine_similarity(z.flatten(), gold["z_fp32"].flatten(), dim=0).item()
    assert cos >= 0.998, cos


def _fake_decoders(vae):
    """Stand-ins for the two decoder stages that keep their data dependences (a temporal chunk's frames depend on the whole latent
    chunk, a pixel frame on its own latent frame only), so the frame-range / shard bookkeeping of decode() can be checked on CPU."""
    def temporal(z4, num_frames, out, f0):
        base = z4.float().mean(dim=1, keepdim=True)                                 # [4, 1, H, W]: every frame sees the chunk
        for i in range(num_frames):
            out[:, f0 + i] = (base[:, 0] + 0.25 * i + z4.float()[:, min(i // 4, z4.shape[1] - 1)]).to(out.dtype)
        return num_frames

    def spatial(xz, out, f0, in_scale=1.0):
        up = xz.float().repeat_interleave(8, dim=2).repeat_interleave(8, dim=3)     # [4, F, 8H, 8W]
        out[:, f0:f0 + xz.shape[1]] = torch.tanh(up[:3] * 0.7 + up[3:4] * 0.1).to(out.dtype)

    vae._temporal_decode, vae._spatial_decode = temporal, spatial


def test_decode_frame_ranges_and_rank_shards_equal_the_full_decode():
    """OpenSoraVAE.decode(frames=(f0, f1)) runs the temporal VAE for the micro-frame chunks that hold the wanted frames and the 2-D
    decoder for those frames only; decode_sharded gives every rank of a group a contiguous block of frames and gathers uint8 frames
    once.  Host bookkeeping checked on CPU with stand-in decoder stages (the kernels' bit-identity is the GPU test's job): every
    range equals the slice of the full decode, every rank of 2 / 3 / 8 returns the full video (22 and 64 frames: 17-frame chunks)."""
    from tools.local_group import LocalWorld
    from vae_cpu_emul import cpu_vae
    from videosys_amd.vae_open_sora import pixels_to_uint8, synth_state_dict

    vae = cpu_vae(synth_state_dict(7), encoder=False)
    vae.frames_per_launch = 3
    assert vae.frame_shards(64, 8) == [(0, 9), (9, 17), (17, 26), (26, 34), (34, 43), (43, 51), (51, 58), (58, 64)]
    _fake_decoders(vae)
    g = torch.Generator().manual_seed(3)
    for frames, tz in ((22, 7), (64, 19), (17, 5)):
        z = torch.randn(2, 4, tz, 3, 2, generator=g)
        full = vae.decode(z, frames)
        assert tuple(full.shape) == (2, 3, frames, 24, 16)
        for f0, f1 in ((0, frames), (0, 1), (16, min(18, frames)), (min(17, frames), frames), (5, 5), (frames - 1, frames), (3, min(21, frames))):
            part = vae.decode(z, frames, frames=(f0, f1))
            assert torch.equal(part, full[:, :, f0:f1]), (frames, f0, f1)
        with pytest.raises(ValueError):
            vae.decode(z, frames, frames=(2, frames + 1))
        want = pixels_to_uint8(full)
        for P in (2, 3, 8):
            shards = vae.frame_shards(frames, P)
            assert shards[0][0] == 0 and shards[-1][1] == frames and all(a[1] == b[0] for a, b in zip(shards, shards[1:]))
            assert shards == [vae.frame_shard(frames, P, r) for r in range(P)]
            if P >= -(-frames // 17):      # as many ranks as 17-frame chunks: no rank's block crosses a chunk border
                assert all(a // 17 == (b - 1) // 17 for a, b in shards if b > a), shards
            outs = LocalWorld(P, timeout=60).run(lambda r, grp: vae.decode_sharded(z, frames, grp))
            for o in outs:
                assert o.dtype == torch.uint8 and torch.equal(o, want), (frames, P)
            outs = LocalWorld(P, timeout=60).run(lambda r, grp: vae.decode_sharded(z, frames, grp, to_uint8=False))
            assert all(torch.equal(o, full) for o in outs)


def _sharded_decode_gloo_worker(rank, world, port, ret):
    """Both sharded decodes through a REAL torch.distributed group (gloo, CPU tensors, stand-in decoder stages): the gathers are
    dist.all_gather_into_tensor calls with the shapes and dtypes the product issues (uint8 frames stacked along dim 0; an int64 header
    and bf16 tiles for the CogVideoX tiles)."""
    try:
        import torch.distributed as dist

        from oracle import cogvideox_vae_oracle as CV
        from vae_cpu_emul import cpu_vae
        from videosys_amd import ops
        from videosys_amd.vae_cogvideox import CogVideoXVAE
        from videosys_amd.vae_open_sora import pixels_to_uint8, synth_state_dict

        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
        torch.manual_seed(0)
        vae = cpu_vae(synth_state_dict(7), encoder=False)
        vae.frames_per_launch = 4
        _fake_decoders(vae)
        g = torch.Generator().manual_seed(3)
        z = torch.randn(1, 4, 7, 3, 2, generator=g)
        want = pixels_to_uint8(vae.decode(z, 22))
        got = vae.decode_sharded(z, 22, dist.group.WORLD)
        assert got.dtype == torch.uint8 and torch.equal(got, want), "OpenSoraVAE.decode_sharded over gloo"
        # CogVideoX tiles
        cv = CogVideoXVAE.__new__(CogVideoXVAE)
        cv.tile_sample_min_height, cv.tile_sample_min_width = 32, 48
        cv.tile_latent_min_height, cv.tile_latent_min_width = 4, 6
        cv.tile_overlap_factor_height, cv.tile_overlap_factor_width = 1 / 6, 1 / 5
        cv._decode_tile = lambda zt: torch.tanh(zt.float()[:3].repeat_interleave(8, dim=2).repeat_interleave(8, dim=3)).to(torch.bfloat16).contiguous()

        def torch_blend(a, b, ext, axis):
            f = CV.blend_v if axis == 0 else CV.blend_h
            b.copy_(f(a.float()[None], b.float()[None], ext)[0].to(b.dtype))
            return b

        ops.blend_edge = torch_blend
        zb = torch.randn(16, 2, 9, 14, generator=g).to(torch.bfloat16)
        want = cv._tiled(zb.clone())
        got = cv._tiled(zb.clone(), dist.group.WORLD)
        assert torch.equal(got, want), "CogVideoXVAE._tiled(group) over gloo"
        ret.put((rank, "ok"))
    except Exception:  # noqa
        import traceback

        ret.put((rank, traceback.format_exc()))
    finally:
        import torch.distributed as dist

        if dist.is_initialized():
            dist.destroy_process_group()


def test_sharded_decodes_over_a_two_rank_gloo_group():
    """World size 2 over gloo on the CPU: OpenSoraVAE.decode_sharded (frames) and CogVideoXVAE._tiled(group) (tiles) through
    torch.distributed itself, every rank returning the unsharded result (autoencoder_kl_open_sora.py:672-695 and
    autoencoder_kl_cogvideox.py:1161-1239 decode everything on every rank)."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_sharded_decode_gloo_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    res = [ret.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"
