"""CPU: pins oracle/cogvideox_oracle.py against fixtures minted from the reference's CogVideoX classes
(oracle/make_golden_cogvideox.py), and the host-side scheduler / RoPE / position tables against the oracle."""
import math

import pytest
import torch

from conftest import load_golden
from oracle import cogvideox_oracle as CO


def _oracle(fx, cfg):
    sd = {k: v.to(torch.bfloat16).float() for k, v in CO.synth_state_dict(cfg["num_layers"], cfg["num_attention_heads"],
                                                                          text_embed_dim=cfg["text_embed_dim"], seed=fx["seed"]).items()}
    return CO.CogVideoXOracle(sd, cfg["num_layers"], cfg["num_attention_heads"], max_text_seq_length=cfg["max_text_seq_length"],
                              sample_width=cfg["sample_width"], sample_height=cfg["sample_height"],
                              sample_frames=cfg["sample_frames"],
                              use_rotary_positional_embeddings=cfg["use_rotary_positional_embeddings"]), sd


def test_forward_matches_reference_golden_both_position_schemes():
    fx = load_golden("cogvideox_fwd_small.pt")
    for key in ("sincos", "rope"):
        o, sd = _oracle(fx, fx[key]["cfg"])
        assert abs(float(sum(v.double().abs().sum() for v in sd.values())) - fx["sd_checksum"]) < 1e-3 * fx["sd_checksum"]
        rope = (fx["rope_cos"], fx["rope_sin"]) if key == "rope" else None
        torch.testing.assert_close(o(fx["x"], fx["y"], fx["t"], rope), fx[key]["out"], rtol=2e-4, atol=2e-4)
    cos, sin = CO.rope_3d(64, fx["crops"], (4, 6), 3)
    torch.testing.assert_close(cos, fx["rope_cos"])
    torch.testing.assert_close(sin, fx["rope_sin"])
    assert (fx["rope"]["out"] - fx["sincos"]["out"]).abs().max() > 1e-2


def test_sampling_matches_reference_golden():
    fx = load_golden("cogvideox_sample_small.pt")
    o, _ = _oracle(fx, fx["cfg"])
    steps = fx["steps"]
    assert CO.ddim_timesteps(steps) == fx["timesteps"]
    ac = CO.ddim_alphas(snr_shift_scale=1.0)
    rope = CO.rope_3d(64, CO.crop_region((4, 6), 45, 30), (4, 6), 3)
    z = fx["latents"].clone()
    emb = torch.cat([fx["neg"], fx["pos"]], 0)
    for i, t in enumerate(fx["timesteps"]):
        v = o(torch.cat([z, z]), emb, torch.tensor([t, t]), rope)
        g = CO.dynamic_cfg(fx["guidance"], t, steps)
        assert math.isclose(g, fx["guidance_per_step"][i], rel_tol=1e-6)
        unc, txt = v.chunk(2)
        c_z, c_v = CO.ddim_coeffs_v(t, steps, ac)
        z = (c_z * z + c_v * (unc + g * (txt - unc))).to(torch.bfloat16).float()
    torch.testing.assert_close(z, fx["out"], rtol=2e-2, atol=2e-2)  # bf16 re-rounding of the latents every step


def test_host_scheduler_and_tables_mirror_oracle():
    from videosys_amd.cogvideox import cogvideox_pos_embed_3d
    from videosys_amd.pipeline_cogvideox import (CogVideoXConfig, CogVideoXDDIMScheduler, CogVideoXPABConfig, CogVideoXPipeline,
                                                 get_3d_rotary_pos_embed, get_resize_crop_region_for_grid)

    for shift in (3.0, 1.0):
        s = CogVideoXDDIMScheduler(snr_shift_scale=shift)
        ac = CO.ddim_alphas(snr_shift_scale=shift)
        torch.testing.assert_close(s.alphas_cumprod, ac)
        for steps in (4, 7, 50):
            s.set_timesteps(steps)
            assert s.timesteps == CO.ddim_timesteps(steps)
            for t in s.timesteps:
                a, b = s.coeffs(t)
                oa, ob = CO.ddim_coeffs_v(t, steps, ac)
                assert math.isclose(a, oa, rel_tol=1e-9) and math.isclose(b, ob, rel_tol=1e-9, abs_tol=1e-12)
    crops = get_resize_crop_region_for_grid((30, 45), 45, 30)
    assert crops == CO.crop_region((30, 45), 45, 30) == ((0, 0), (30, 45))
    cos, sin = get_3d_rotary_pos_embed(64, crops, (30, 45), 2)
    oc, os_ = CO.rope_3d(64, crops, (30, 45), 2)
    assert torch.equal(cos, oc) and torch.equal(sin, os_)
    torch.testing.assert_close(cogvideox_pos_embed_3d(192, 6, 4, 3, 1.875, 1.0), CO.sincos_3d(192, 6, 4, 3, 1.875, 1.0))
    c = CogVideoXConfig()
    assert (c.model_path, c.num_gpus, c.vae_tiling, c.enable_pab) == ("THUDM/CogVideoX-2b", 1, True, False)
    assert c.pipeline_cls is CogVideoXPipeline
    p = CogVideoXPABConfig()
    assert (p.spatial_threshold, p.spatial_range, p.spatial_broadcast) == ([100, 850], 2, True)


def test_oracle_vs_live_reference():
    from oracle import ref_loader

    if not ref_loader.reference_available():
        pytest.skip("reference tree not present (GPU box)")
    fx = load_golden("cogvideox_fwd_small.pt")
    cfg = fx["rope"]["cfg"]
    o, sd = _oracle(fx, cfg)
    model = ref_loader.build_reference_cogvideox(cfg, sd)
    g = torch.Generator().manual_seed(3)
    x, y = torch.randn(1, 2, 16, 8, 8, generator=g), torch.randn(1, 7, 128, generator=g)
    rope = CO.rope_3d(64, ((0, 0), (4, 4)), (4, 4), 2)
    t = torch.tensor([321])
    with torch.no_grad():
        ref = model(x, y, t, image_rotary_emb=rope, return_dict=False)[0]
    torch.testing.assert_close(o(x, y, t, rope), ref, rtol=2e-4, atol=2e-4)
