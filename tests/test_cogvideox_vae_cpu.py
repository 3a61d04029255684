"""CPU: the CogVideoX-VAE oracle against the golden minted from the reference's AutoencoderKLCogVideoX (plain, even-frame and tiled
decodes), the parameter inventory against the reference class, and the tile geometry at the published sample size."""
import pytest
import torch

from conftest import load_golden


def test_cogvideox_vae_oracle_matches_reference_golden():
    from oracle import cogvideox_vae_oracle as CV
    from oracle.make_golden import sd_checksum
    from videosys_amd.vae_cogvideox import synth_state_dict

    gold = load_golden("cogvideox_vae_small.pt")
    sd = synth_state_dict(gold["seed"])
    assert sd_checksum(sd) == gold["sd_checksum"]
    sh, sw = gold["sample"]
    for key, z, tiling in (("even", gold["z_even"], False), ("tiled", gold["z"], True)):
        out = CV.decode(sd, z, sh, sw, tiling=tiling)
        assert (out - gold[key].float()).abs().max().item() < 3e-3, key   # fixture stored as fp16


def test_cogvideox_vae_param_inventory_matches_reference():
    from oracle import ref_loader

    if not ref_loader.reference_available():
        pytest.skip("reference tree not present on this box")
    from videosys_amd.vae_cogvideox import decoder_param_shapes

    m = ref_loader.build_reference_cogvideox_vae()
    ref = {k: tuple(v.shape) for k, v in m.state_dict().items() if k.startswith("decoder.")}
    assert decoder_param_shapes() == ref


def test_tile_geometry_at_published_size():
    """480 x 720 samples (autoencoder_kl_cogvideox.py:983-995,1183-1189): latent tiles 30 x 45, strides 25 x 36, cross-fades
    40 x 72 pixels, kept 200 x 288 pixels per tile -> 3 x 3 tiles for the 60 x 90 latent of config 5."""
    from oracle import cogvideox_vae_oracle as CV

    g = CV.tile_geometry(480, 720)
    assert (g["tl_h"], g["tl_w"], g["ov_h"], g["ov_w"], g["be_h"], g["be_w"], g["lim_h"], g["lim_w"]) == (30, 45, 25, 36, 40, 72, 200, 288)
    assert len(range(0, 60, g["ov_h"])) == 3 and len(range(0, 90, g["ov_w"])) == 3
    assert CV._frame_batches(13) == [(0, 3), (3, 5), (5, 7), (7, 9), (9, 11), (11, 13)]


def test_host_frame_batching_matches_reference_decode_shapes():
    """CogVideoXVAE's host-side frame batching / output frame count (no GPU needed) against the shapes the reference produced in
    the golden: 5 latent frames -> 17, 4 -> 16; and the published 13 -> 49."""
    from videosys_amd.vae_cogvideox import CogVideoXVAE

    v = CogVideoXVAE.__new__(CogVideoXVAE)
    gold = load_golden("cogvideox_vae_small.pt")
    assert v._batches(5) == [(0, 3), (3, 5)] and v._out_frames(5) == gold["plain"].shape[2] == 17
    assert v._batches(4) == [(0, 2), (2, 4)] and v._out_frames(4) == gold["even"].shape[2] == 16
    assert v._out_frames(13) == 49


def test_vae_grid_arithmetic():
    """ops.VaeGrid: rows, guards and the conv-output grid of a padded causal input (pure host arithmetic)."""
    from videosys_amd.ops import VaeGrid

    g = VaeGrid(2, 5, 12, 8, pad=1, tf=2)
    assert (g.Hp, g.Wp, g.plane, g.sample_rows, g.rows, g.guard) == (14, 10, 140, 7 * 140, 2 * 7 * 140, 11)
    o = g.conv_out()
    assert (o.T, o.tf, o.pad, o.rows) == (5, 0, 1, 2 * 5 * 140)
    d = VaeGrid(3, 1, 4, 4, sample_rows=128)
    assert d.rows == 384 and d.guard == 0
    with pytest.raises(AssertionError):
        VaeGrid(1, 1, 4, 4, sample_rows=8)


def test_cogvideox_vae_oracle_matches_live_reference_class():
    """oracle/cogvideox_vae_oracle.py against the reference's AutoencoderKLCogVideoX run here (another seed, 7 latent frames =
    frame batches 3 + 2 + 2, tiled 2 x 2) when /root/reference is present: bit-exact."""
    from oracle import ref_loader

    if not ref_loader.reference_available():
        pytest.skip("reference tree not present on this box")
    from oracle import cogvideox_vae_oracle as CV
    from videosys_amd.vae_cogvideox import synth_state_dict

    sd = synth_state_dict(5)
    m = ref_loader.build_reference_cogvideox_vae(sd, sample_height=96, sample_width=160)
    m.enable_tiling()
    g = torch.Generator().manual_seed(6)
    z = torch.randn(1, 16, 7, 8, 12, generator=g)
    with torch.no_grad():
        ref = m.decode(z).sample
    out = CV.decode(sd, z, 96, 160, tiling=True)
    assert out.shape == ref.shape == (1, 3, 25, 64, 96)
    assert torch.equal(out, ref)


def test_cogvideox_vae_needs_gpu():
    from videosys_amd.vae_cogvideox import CogVideoXVAE
    from videosys_amd.vae_open_sora import AutoencoderKLDecoder

    with pytest.raises(RuntimeError):
        CogVideoXVAE({}, device="cpu")
    with pytest.raises(RuntimeError):
        AutoencoderKLDecoder({}, device="cpu")


def test_tiled_decode_with_tiles_shared_out_over_ranks_host_bookkeeping():
    """CogVideoXVAE._tiled(group=...): tiles r, r + P, ... decoded by rank r, padded to the full tile's pixel shape, gathered once,
    cut back to their own shapes and blended in the reference's order (autoencoder_kl_cogvideox.py:1161-1239) — on CPU with a stand-in
    tile decoder and a torch blend (the oracle's), 2 / 3 / 4 ranks as threads: every rank returns the unsharded result, incl. the
    ragged last tiles of a 9 x 14 latent and more ranks than tiles in a row."""
    from oracle import cogvideox_vae_oracle as CV
    from tools.local_group import LocalWorld
    from videosys_amd import ops
    from videosys_amd.vae_cogvideox import CogVideoXVAE

    vae = CogVideoXVAE.__new__(CogVideoXVAE)
    vae.tile_sample_min_height, vae.tile_sample_min_width = 32, 48
    vae.tile_latent_min_height, vae.tile_latent_min_width = 4, 6
    vae.tile_overlap_factor_height, vae.tile_overlap_factor_width = 1 / 6, 1 / 5

    def fake_tile(zt):       # [16, T, h, w] -> [3, 4T - 3, 8h, 8w]: a deterministic function of the latent tile alone
        x = zt.float()[:3].repeat_interleave(8, dim=2).repeat_interleave(8, dim=3)
        x = torch.cat([x[:, :1]] + [x[:, 1:].repeat_interleave(4, dim=1)], dim=1) if x.shape[1] > 1 else x
        return torch.tanh(x + 0.05 * zt.float()[3:6].mean()).to(torch.bfloat16).contiguous()

    def torch_blend(a, b, ext, axis):
        f = CV.blend_v if axis == 0 else CV.blend_h
        b.copy_(f(a.float()[None], b.float()[None], ext)[0].to(b.dtype))
        return b

    # config 5: 3 x 3 tiles of 30 x 45 latent pixels stepping 25 / 36 over 60 x 90 -> the last row / column are 10 / 18 wide
    areas = [min(30, 60 - i) * min(45, 90 - j) for i in (0, 25, 50) for j in (0, 36, 72)]
    owner, slot, per = CogVideoXVAE._deal_tiles(areas, 4)
    loads = [sum(a for a, o in zip(areas, owner) if o == r) for r in range(4)]
    assert sum(areas) == 7560 and max(loads) == 1980 and per == 3 and owner[0] == 0
    assert sorted((o, s_) for o, s_ in zip(owner, slot)) == sorted(set((o, s_) for o, s_ in zip(owner, slot)))    # no slot used twice
    vae._decode_tile = fake_tile
    saved = ops.blend_edge
    ops.blend_edge = torch_blend
    try:
        g = torch.Generator().manual_seed(4)
        for (H, W) in ((8, 12), (9, 14)):
            zb = torch.randn(16, 3, H, W, generator=g).to(torch.bfloat16)
            want = vae._tiled(zb.clone())
            for P in (2, 3, 4):
                outs = LocalWorld(P, timeout=60).run(lambda r, grp: vae._tiled(zb.clone(), grp))
                for r, o in enumerate(outs):
                    assert o.shape == want.shape and torch.equal(o, want), (H, W, P, r)
    finally:
        ops.blend_edge = saved
