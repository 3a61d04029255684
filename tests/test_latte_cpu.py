"""CPU: pins oracle/latte_oracle.py against the fixtures minted from the reference's LatteT2V class
(oracle/make_golden_latte.py) and, when /root/reference is present, against the live reference; checks the host-side
DDIM scheduler mirror against the oracle / the diffusers restatement."""
import math

import pytest
import torch

from conftest import load_golden
from oracle import latte_oracle as LO


def _oracle(fx):
    cfg = fx["cfg"]
    sd = LO.synth_state_dict(cfg["num_layers"], cfg["num_attention_heads"], cfg["attention_head_dim"],
                             caption_channels=cfg["caption_channels"], seed=fx["seed"])
    sd = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    if "sd_checksum" in fx:
        assert abs(float(sum(v.double().abs().sum() for v in sd.values())) - fx["sd_checksum"]) < 1e-3 * fx["sd_checksum"]
    return LO.LatteOracle(sd, cfg["num_layers"], cfg["num_attention_heads"], cfg["attention_head_dim"],
                          sample_size=cfg["sample_size"], video_length=cfg["video_length"]), sd


def test_latte_forward_matches_reference_golden():
    fx = load_golden("latte_fwd_small.pt")
    o, _ = _oracle(fx)
    torch.testing.assert_close(o(fx["x"], fx["t"], fx["y"], fx["mask"]), fx["out"], rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(o(fx["x"], fx["t"], fx["y"], None), fx["out_nomask"], rtol=2e-4, atol=2e-4)
    assert (fx["out"] - fx["out_nomask"]).abs().max() > 1e-3  # the mask matters in this fixture


def test_latte_sampling_matches_reference_golden():
    fx = load_golden("latte_sample_small.pt")
    o, _ = _oracle(fx)
    assert LO.ddim_timesteps(fx["steps"]) == fx["timesteps"]
    z = LO.latte_sample(o, fx["latents"], fx["pos"], fx["neg"], fx["pmask"], fx["nmask"], fx["steps"], fx["guidance"])
    torch.testing.assert_close(z, fx["out"], rtol=5e-4, atol=5e-4)


def test_ddim_mirror_matches_oracle_and_stub():
    from oracle.diffusers_stub import DDIMScheduler as StubDDIM
    from videosys_amd.pipeline_latte import DDIMScheduler

    ac = LO.ddim_tables()
    for steps in (4, 20, 50):
        s = DDIMScheduler()
        s.set_timesteps(steps)
        st = StubDDIM()
        st.set_timesteps(steps)
        assert s.timesteps == LO.ddim_timesteps(steps) == [int(v) for v in st.timesteps]
        g = torch.Generator().manual_seed(steps)
        z, eps = torch.randn(2, 3, generator=g), torch.randn(2, 3, generator=g)
        for t in s.timesteps:
            c_z, c_eps = s.coeffs(t)
            oz, oe = LO.ddim_coeffs(t, steps, ac)
            assert math.isclose(c_z, oz, rel_tol=1e-6) and math.isclose(c_eps, oe, rel_tol=1e-6, abs_tol=1e-7)
            torch.testing.assert_close(c_z * z + c_eps * eps, st.step(eps, t, z)[0], rtol=1e-5, atol=1e-5)


def test_latte_config_defaults_mirror_reference():
    from videosys_amd import LatteConfig, LattePABConfig, LattePipeline

    c = LatteConfig()
    assert (c.model_path, c.num_gpus, c.beta_start, c.beta_end, c.beta_schedule, c.variance_type) == \
        ("maxin-cn/Latte-1", 1, 0.0001, 0.02, "linear", "learned_range")
    assert c.pipeline_cls is LattePipeline and c.enable_pab is False
    p = LattePABConfig()
    assert (p.spatial_threshold, p.spatial_range, p.temporal_range, p.cross_range) == ([100, 800], 2, 3, 6)
    assert sorted(p.mlp_spatial_broadcast_config) == [400, 480, 560, 640, 720] and p.mlp_broadcast


def test_latte_oracle_vs_live_reference():
    from oracle import ref_loader

    if not ref_loader.reference_available():
        pytest.skip("reference tree not present (GPU box)")
    fx = load_golden("latte_fwd_small.pt")
    o, sd = _oracle(fx)
    model = ref_loader.build_reference_latte(fx["cfg"], sd)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 4, 4, 12, 20, generator=g)  # non-square grid, not the sample_size grid
    y = torch.randn(2, 9, 64, generator=g)
    t = torch.tensor([801, 801])
    with torch.no_grad():
        ref = model(x, timestep=t, encoder_hidden_states=y, encoder_attention_mask=None,
                    added_cond_kwargs={"resolution": None, "aspect_ratio": None}, enable_temporal_attentions=True,
                    return_dict=False)[0]
    torch.testing.assert_close(o(x, t, y, None), ref, rtol=2e-4, atol=2e-4)
