"""TEST INFRASTRUCTURE ONLY — mints tests/golden/latte_*.pt from the reference's LatteT2V class.

    python oracle/make_golden_latte.py       (build container only: needs /root/reference)

The reference's own latte_transformer_3d.py (blocks, forward, PAB hooks, final layer, unpatchify) runs on CPU in fp32
over the restated diffusers==0.30.0 leaves of oracle/diffusers_stub.py (diffusers is pinned by the reference but not
vendored / installable here).  Inputs and weights are rounded to bf16-representable values first, so what is pinned is
"reference fp32 on bf16-rounded inputs".  The sampling fixture drives the reference transformer with the denoising loop
of pipeline_latte.py:798-876 (CFG batch [negative | prompt], learned-sigma half dropped) and the DDIM stub.
"""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import diffusers_stub, ref_loader  # noqa: E402
from oracle import latte_oracle as LO  # noqa: E402
from oracle.make_golden import OUT, bf16r, pack, sd_checksum  # noqa: E402

# C = 8 heads x 72 = 576 (GEMM-friendly: 576 = 3 x 192 = 9 x 64), two (spatial, temporal) block pairs
LATTE_CFG = dict(num_attention_heads=8, attention_head_dim=72, in_channels=4, out_channels=8, num_layers=2,
                 cross_attention_dim=576, attention_bias=True, sample_size=16, patch_size=2,
                 activation_fn="gelu-approximate", norm_type="ada_norm_single", norm_elementwise_affine=False, norm_eps=1e-6,
                 caption_channels=64, video_length=4)
LATTE_SEED = 99
COND = {"resolution": None, "aspect_ratio": None}


def weights():
    sd = LO.synth_state_dict(LATTE_CFG["num_layers"], 8, 72, caption_channels=64, seed=LATTE_SEED)
    return {k: bf16r(v) for k, v in sd.items()}


def inputs(seed=3, B=2, L=12):
    g = torch.Generator().manual_seed(seed)
    x = bf16r(torch.randn(B, 4, 4, 16, 16, generator=g))
    y = bf16r(torch.randn(B, L, 64, generator=g))
    mask = torch.ones(B, L, dtype=torch.long)
    mask[0, 7:] = 0  # the two CFG halves carry prompts of different length
    return x, y, mask


def ref_call(model, x, t, y, mask, all_timesteps=None):
    return model(x, timestep=t, all_timesteps=all_timesteps, encoder_hidden_states=y, encoder_attention_mask=mask,
                 added_cond_kwargs=COND, enable_temporal_attentions=True, return_dict=False)[0]


def make_fwd():
    sd = weights()
    model = ref_loader.build_reference_latte(LATTE_CFG, sd)
    x, y, mask = inputs()
    t = torch.tensor([437, 437])
    with torch.no_grad():
        out = ref_call(model, x, t, y, mask)
        out_nomask = ref_call(model, x, t, y, None)
    fix = dict(cfg=LATTE_CFG, seed=LATTE_SEED, sd_checksum=sd_checksum(sd), x=x, t=t, y=y, mask=mask, out=out,
               out_nomask=out_nomask)
    torch.save(pack(fix), os.path.join(OUT, "latte_fwd_small.pt"))
    print("latte_fwd_small.pt", tuple(out.shape), float(out.abs().max()))


def make_sample(steps=4, guidance=7.5):
    sd = weights()
    model = ref_loader.build_reference_latte(LATTE_CFG, sd)
    x, y, mask = inputs(seed=11, B=2)
    z = x[:1].clone()
    neg, pos = y[:1], y[1:]
    nmask, pmask = mask[:1], mask[1:]
    sched = diffusers_stub.DDIMScheduler()
    sched.set_timesteps(steps)
    emb = torch.cat([neg, pos], 0)
    m = torch.cat([nmask, pmask], 0)
    traj = []
    with torch.no_grad():
        for t in sched.timesteps:
            zin = torch.cat([z] * 2)
            tt = t[None].expand(2)
            noise = ref_call(model, zin, tt, emb, m, all_timesteps=sched.timesteps)
            unc, txt = noise.chunk(2)
            noise = unc + guidance * (txt - unc)
            noise = noise.chunk(2, dim=1)[0]
            z = sched.step(noise, t, z)[0]
            traj.append(z.clone())
    fix = dict(cfg=LATTE_CFG, seed=LATTE_SEED, steps=steps, guidance=guidance, latents=x[:1], neg=neg, pos=pos, nmask=nmask,
               pmask=pmask, timesteps=[int(v) for v in sched.timesteps], out=z, first=traj[0])
    torch.save(pack(fix), os.path.join(OUT, "latte_sample_small.pt"))
    print("latte_sample_small.pt", tuple(z.shape), float(z.abs().max()))


def make_pab(steps=6):
    """PAB broadcast schedule through the reference blocks: the attention / cross / mlp flags per step and the output."""
    import importlib

    sd = weights()
    model = ref_loader.build_reference_latte(LATTE_CFG, sd)
    pab_mgr = importlib.import_module("videosys.core.pab.pab_mgr")
    cfgp = pab_mgr.PABConfig(spatial_broadcast=True, spatial_threshold=[100, 900], spatial_range=2, temporal_broadcast=True,
                             temporal_threshold=[100, 900], temporal_range=3, cross_broadcast=True,
                             cross_threshold=[100, 900], cross_range=4, mlp_broadcast=True,
                             mlp_spatial_broadcast_config={664: {"block": [0, 1], "skip_count": 2}},
                             mlp_temporal_broadcast_config={664: {"block": [1], "skip_count": 2}})
    pab_mgr.set_pab_manager(cfgp)
    pab_mgr.update_steps(steps)
    x, y, mask = inputs(seed=21)
    sched = diffusers_stub.DDIMScheduler()
    sched.set_timesteps(steps)
    outs = []
    with torch.no_grad():
        for t in sched.timesteps:
            outs.append(ref_call(model, x, t[None].expand(2), y, mask, all_timesteps=sched.timesteps))
    pab_mgr.PAB_MANAGER = None
    fix = dict(cfg=LATTE_CFG, seed=LATTE_SEED, steps=steps, x=x, y=y, mask=mask, timesteps=[int(v) for v in sched.timesteps],
               outs=outs,
               pab=dict(spatial_threshold=[100, 900], spatial_range=2, temporal_threshold=[100, 900], temporal_range=3,
                        cross_threshold=[100, 900], cross_range=4,
                        mlp_spatial_broadcast_config={664: {"block": [0, 1], "skip_count": 2}},
                        mlp_temporal_broadcast_config={664: {"block": [1], "skip_count": 2}}))
    torch.save(pack(fix), os.path.join(OUT, "latte_pab_small.pt"))
    print("latte_pab_small.pt", len(outs), [float((outs[i] - outs[i - 1]).abs().max()) for i in range(1, len(outs))])


if __name__ == "__main__":
    which = sys.argv[1:] or ["fwd", "sample", "pab"]
    for w in which:
        {"fwd": make_fwd, "sample": make_sample, "pab": make_pab}[w]()
