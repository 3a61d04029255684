"""GPU: CogVideoXVAE (causal 3-D VAE decode with conv caches, spatial norm, time-doubling upsampling, tiled decode) against the
golden minted from the reference's AutoencoderKLCogVideoX; kernel checks for the pieces it adds."""
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def bfr(t):
    return t.to(torch.bfloat16).float()


def test_blend_edge_matches_reference_loops():
    from oracle import cogvideox_vae_oracle as CV
    from videosys_amd import ops

    g = torch.Generator().manual_seed(1)
    a = torch.randn(1, 3, 4, 12, 20, generator=g).to(torch.bfloat16)
    b = torch.randn(1, 3, 4, 10, 20, generator=g).to(torch.bfloat16)
    want = CV.blend_v(a.clone(), b.clone(), 6)
    got = ops.blend_edge(a[0].to(dev()).contiguous(), b[0].to(dev()).contiguous(), 6, 0)
    assert torch.equal(got.cpu(), want[0])
    b2 = torch.randn(1, 3, 4, 12, 9, generator=g).to(torch.bfloat16)
    want = CV.blend_h(a.clone(), b2.clone(), 5)
    got = ops.blend_edge(a[0].to(dev()).contiguous(), b2[0].to(dev()).contiguous(), 5, 1)
    assert torch.equal(got.cpu(), want[0])


@pytest.mark.parametrize("T,zT", [(3, 3), (5, 3), (9, 3), (2, 2), (8, 2)])
def test_spatial_norm_silu_matches_torch(T, zT):
    from videosys_amd import ops

    g = torch.Generator().manual_seed(T * 10 + zT)
    C, H, W, zH, zW = 128, 8, 12, 4, 6
    x = bfr(torch.randn(1, C, T, H, W, generator=g) * 1.3 + 0.4)
    zq = bfr(torch.randn(1, 16, zT, zH, zW, generator=g))
    gamma = bfr(1 + 0.1 * torch.randn(C, generator=g)); beta = bfr(0.1 * torch.randn(C, generator=g))
    wy = bfr(torch.randn(C, 16, generator=g) * 0.1); by = bfr(1 + 0.1 * torch.randn(C, generator=g))
    wb = bfr(torch.randn(C, 16, generator=g) * 0.1); bb = bfr(0.1 * torch.randn(C, generator=g))
    # reference math (CogVideoXSpatialNorm3D.forward) in fp32 with the bf16 rounding points of a bf16 run
    if T > 1 and T % 2 == 1:
        zi = torch.cat([F.interpolate(zq[:, :, :1], size=(1, H, W)), F.interpolate(zq[:, :, 1:], size=(T - 1, H, W))], 2)
    else:
        zi = F.interpolate(zq, size=(T, H, W))
    nf = bfr(F.group_norm(x, 32, gamma, beta, 1e-6))
    Y = bfr(torch.einsum("oc,bcthw->bothw", wy, zi) + by[None, :, None, None, None])
    Bv = bfr(torch.einsum("oc,bcthw->bothw", wb, zi) + bb[None, :, None, None, None])
    ref = F.silu(bfr(bfr(nf * Y) + Bv))
    gs = ops.VaeGrid(1, T, H, W, 1, 0)
    xb, xr = gs.alloc(C, dev(), zero=True)
    xr.view(T, H + 2, W + 2, C)[:, 1:-1, 1:-1] = x[0].permute(1, 2, 3, 0).to(torch.bfloat16).to(dev())
    gd = ops.VaeGrid(1, T, H, W, 1, 2)
    yb_, yr = gd.alloc(C, dev(), zero=True)
    zrows = torch.zeros(zT * zH * zW, 32, dtype=torch.bfloat16, device=dev())
    zrows[:, :16] = zq[0].permute(1, 2, 3, 0).reshape(-1, 16).to(torch.bfloat16).to(dev())
    w_yb = torch.zeros(2 * C, 32, dtype=torch.bfloat16, device=dev())
    w_yb[:C, :16] = wy.to(torch.bfloat16).to(dev()); w_yb[C:, :16] = wb.to(torch.bfloat16).to(dev())
    b_yb = torch.cat([by, bb]).to(torch.bfloat16).to(dev())
    yb = ops.gemm128(zrows, w_yb, b_yb)
    ops.spatial_norm_silu(xr, gs, yr, gd, C, gamma.to(torch.bfloat16).to(dev()), beta.to(torch.bfloat16).to(dev()), yb, (zT, zH, zW))
    got = yr.view(T + 2, H + 2, W + 2, C)[2:, 1:-1, 1:-1].permute(3, 0, 1, 2).float().cpu()
    err = (got - ref[0]).abs().max().item()
    assert err <= 2 ** -6 * ref.abs().max().item(), err


def test_regrid_time_modes():
    from videosys_amd import ops

    g = torch.Generator().manual_seed(4)
    C, H, W = 64, 3, 5
    for T, tmode in ((3, 2), (2, 1), (5, 2), (4, 1)):
        x = bfr(torch.randn(1, C, T, H, W, generator=g))
        if tmode == 2:
            want = torch.cat([F.interpolate(x[:, :, 0], scale_factor=2.0)[:, :, None], F.interpolate(x[:, :, 1:], scale_factor=2.0)], 2)
        else:
            want = F.interpolate(x, scale_factor=2.0)
        gs = ops.VaeGrid(1, T, H, W, 1, 0)
        _, xr = gs.alloc(C, dev(), zero=True)
        xr.view(T, H + 2, W + 2, C)[:, 1:-1, 1:-1] = x[0].permute(1, 2, 3, 0).to(torch.bfloat16).to(dev())
        T2 = want.shape[2]
        gd = ops.VaeGrid(1, T2, 2 * H, 2 * W, 1, 0)
        _, yr = gd.alloc(C, dev(), zero=True)
        ops.regrid(xr, gs, yr, gd, C, up=1, tmode=tmode)
        got = yr.view(T2, 2 * H + 2, 2 * W + 2, C)[:, 1:-1, 1:-1].permute(3, 0, 1, 2).float().cpu()
        assert torch.equal(got, want[0])


def test_cogvideox_vae_decode_matches_reference_golden():
    from videosys_amd.vae_cogvideox import CogVideoXVAE, synth_state_dict

    gold = load_golden("cogvideox_vae_small.pt")
    sh, sw = gold["sample"]
    vae = CogVideoXVAE(synth_state_dict(gold["seed"]), device=dev(), sample_height=sh, sample_width=sw, use_tiling=False)
    rms = lambda t: t.pow(2).mean().sqrt().item()
    ref_t = gold["tiled"].float()
    floor = rms(gold["tiled_bf16"].float() - ref_t) / rms(ref_t)
    for key, z, tiling in (("plain", gold["z"], False), ("even", gold["z_even"], False), ("tiled", gold["z"], True)):
        vae.use_tiling = tiling
        out = vae.decode(z.to(dev())).float().cpu()
        ref = gold[key].float()
        assert out.shape == ref.shape, (key, out.shape, ref.shape)
        assert torch.isfinite(out).all()
        mine = rms(out - ref) / rms(ref)
        cos = F.cosine_similarity(out.flatten(), ref.flatten(), dim=0).item()
        assert mine <= 1.5 * floor + 1e-3 and cos >= 0.999, f"{key}: rel rms {mine:.4f} (reference-bf16 floor {floor:.4f}), cosine {cos:.5f}"
    # decode_latents: pipeline layout [B, T, C, H, W] and the 1 / scaling_factor factor (pipeline_cogvideox.py:359-364)
    lat = (gold["z"] * vae.config.scaling_factor).permute(0, 2, 1, 3, 4).to(torch.bfloat16)
    fr = vae.decode_latents(lat.to(dev())).float().cpu()
    assert fr.shape == gold["tiled"].shape
    assert F.cosine_similarity(fr.flatten(), ref_t.flatten(), dim=0).item() >= 0.998


def test_cogvideox_pipeline_latents_to_uint8_video():
    """CogVideoXPipeline.generate end to end: embeddings -> DDIM denoising (small transformer) -> CogVideoXVAE -> uint8 video."""
    from videosys_amd import CogVideoXConfig, CogVideoXPipeline
    from videosys_amd.vae_cogvideox import CogVideoXVAE

    fx = load_golden("cogvideox_sample_small.pt")
    pipe = CogVideoXPipeline(CogVideoXConfig(model_path=f"THUDM/CogVideoX-5b@synthetic:{fx['seed']}", transformer_config=fx["cfg"]),
                             device=dev())
    assert isinstance(pipe.vae_decoder, CogVideoXVAE) and pipe.vae_decoder.config.scaling_factor == 0.7
    kw = dict(prompt_embeds=fx["pos"], negative_prompt_embeds=fx["neg"], latents=fx["latents"], height=64, width=96, num_frames=9,
              num_inference_steps=2, guidance_scale=fx["guidance"], use_dynamic_cfg=True)
    video = pipe.generate(**kw).video
    assert video.dtype == torch.uint8 and tuple(video.shape) == (1, 9, 64, 96, 3) and video.device.type == "cpu"
    lat = pipe.generate(output_type="latent", **kw).video
    fr = pipe.vae_decoder(lat.to(torch.bfloat16))
    ref = ((fr.float() / 2.0 + 0.5).clamp(0, 1) * 255).round().permute(0, 2, 3, 4, 1).to("cpu", torch.uint8)
    assert torch.equal(video, ref) and video.float().std().item() > 1.0
