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


def _fake_velocity(z, t):
    return torch.tanh(z * 0.7 + 0.001 * t) * 0.9 - 0.1 * z.roll(1, dims=-1)        # oracle/make_golden_dpm.py fake_model


def _drive(sched, steps, seed, shape):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(tuple(shape), generator=g)
    sched.set_timesteps(steps)
    ts = [int(t) for t in sched.timesteps]
    old, traj, x0s = None, [], []
    for i, t in enumerate(ts):
        z, old = sched.step(_fake_velocity(z, t), old, t, ts[i - 1] if i > 0 else None, z, generator=g, return_dict=False)
        traj.append(z.clone())
        x0s.append(old.clone())
    return ts, traj, x0s


def test_dpm_scheduler_matches_the_reference_class():
    """videosys_amd.pipeline_cogvideox.CogVideoXDPMScheduler against the fixture minted from the reference's own class
    (schedulers/scheduling_dpm_cogvideox.py:119-483 driven as pipeline_cogvideox.py:679-721 drives it: x0 of the previous step and its
    timestep handed back in, one seeded CPU generator): timesteps, every step's latents and x0 prediction for the 2b / 5b settings and
    a leading-spacing schedule without zero terminal SNR — including the infinite log-SNR of the first (zero-SNR) and last steps and
    the ORDER of the noise draws (a second-order step draws twice and uses the second).  Then the same against the live class."""
    from videosys_amd.pipeline_cogvideox import CogVideoXDDIMScheduler, CogVideoXDPMScheduler

    fx = load_golden("cogvideox_dpm_small.pt")
    for name, case in fx["cases"].items():
        sched = CogVideoXDPMScheduler(**case["kwargs"])
        assert isinstance(sched, CogVideoXDDIMScheduler)
        ts, traj, x0s = _drive(sched, case["steps"], case["seed"], fx["shape"])
        assert ts == case["timesteps"], name
        for i, (a, b) in enumerate(zip(traj, case["traj"])):
            assert torch.isfinite(a).all() and (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item()), (name, i)
        for i, (a, b) in enumerate(zip(x0s, case["x0"])):
            assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item()), (name, "x0", i)
        # the multipliers of the first step of a zero-terminal-SNR schedule: h = inf, pure x0 + fresh noise; of the last: x0 exactly
        sa, sb, m1, m2, m3, m4, mn, second = sched.multipliers(ts[-1], ts[-2])
        if ts[-1] - 1000 // case["steps"] < 0:      # the step lands on "alpha = 1" (:395-399): a first-order step onto x0 itself
            assert not second and m1 == 0.0 and m2 == -1.0 and mn == 0.0
        else:
            assert second and m3 > 1.0 and m4 > 0.0
        if case["kwargs"]["rescale_betas_zero_snr"]:
            sa, sb, m1, m2, m3, m4, mn, second = sched.multipliers(ts[0], None)
            assert sa == 0.0 and sb == 1.0 and m1 == 0.0 and m3 is None and abs(m2 * m2 + mn * mn - 1.0) < 1e-12
    from oracle import ref_loader

    if ref_loader.reference_available():
        kw = dict(prediction_type="v_prediction", timestep_spacing="trailing", rescale_betas_zero_snr=True, snr_shift_scale=1.0)
        ref = ref_loader.load_reference_cogvideox_dpm_scheduler(**kw)
        a = _drive(ref, 7, 99, (1, 2, 16, 4, 4))
        b = _drive(CogVideoXDPMScheduler(**kw), 7, 99, (1, 2, 16, 4, 4))
        assert a[0] == b[0]
        assert all((x - y).abs().max().item() <= 2e-5 for x, y in zip(a[1], b[1]))
