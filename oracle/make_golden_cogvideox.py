"""TEST INFRASTRUCTURE ONLY — mints tests/golden/cogvideox_*.pt from the reference's CogVideoX classes.

    python oracle/make_golden_cogvideox.py       (build container only: needs /root/reference)

The reference's own cogvideox_transformer_3d.py (processor, block, forward), modules/normalization.py,
modules/embeddings.py (patch embed, 3-D RoPE) and schedulers/scheduling_ddim_cogvideox.py run on CPU in fp32 over the
restated diffusers==0.30.0 leaves of oracle/diffusers_stub.py.  Inputs / weights are bf16-rounded first.  The sampling
fixture drives them with the loop of pipeline_cogvideox.py:675-723 (CFG [negative | prompt], dynamic CFG, v-prediction).
"""
from __future__ import annotations

import importlib
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import cogvideox_oracle as CO  # noqa: E402
from oracle import ref_loader  # noqa: E402
from oracle.make_golden import OUT, bf16r, pack, sd_checksum  # noqa: E402

SEED = 555


def cfg_for(rope: bool):
    return dict(num_attention_heads=3, attention_head_dim=64, in_channels=16, out_channels=16, time_embed_dim=512,
                text_embed_dim=128, num_layers=2, sample_width=12, sample_height=8, sample_frames=9, patch_size=2,
                max_text_seq_length=10, use_rotary_positional_embeddings=rope)


def weights():
    return {k: bf16r(v) for k, v in CO.synth_state_dict(2, 3, text_embed_dim=128, seed=SEED).items()}


def inputs(seed=1, B=2):
    g = torch.Generator().manual_seed(seed)
    return bf16r(torch.randn(B, 3, 16, 8, 12, generator=g)), bf16r(torch.randn(B, 10, 128, generator=g))


def rope_tables():
    emb = importlib.import_module("videosys.models.modules.embeddings")
    crops = CO.crop_region((4, 6), 720 // 16, 480 // 16)  # the pipeline's own call shape (:456-470) on a 64x96 px video
    return emb.get_3d_rotary_pos_embed(64, crops, (4, 6), 3), crops


def make_fwd():
    sd = weights()
    fix = dict(seed=SEED, sd_checksum=sd_checksum(sd))
    x, y = inputs()
    t = torch.tensor([713, 713])
    for rope in (False, True):
        model = ref_loader.build_reference_cogvideox(cfg_for(rope), sd)
        (cos, sin), crops = rope_tables()
        with torch.no_grad():
            out = model(x, y, t, image_rotary_emb=(cos, sin) if rope else None, return_dict=False)[0]
        fix["rope" if rope else "sincos"] = dict(cfg=cfg_for(rope), out=out)
        fix["crops"] = crops
        fix["rope_cos"], fix["rope_sin"] = cos, sin
        print("fwd", rope, tuple(out.shape), float(out.abs().max()))
    fix.update(x=x, y=y, t=t)
    torch.save(pack(fix), os.path.join(OUT, "cogvideox_fwd_small.pt"))


def make_sample(steps=4, guidance=6.0):
    sd = weights()
    model = ref_loader.build_reference_cogvideox(cfg_for(True), sd)
    sched = ref_loader.load_reference_cogvideox_scheduler(prediction_type="v_prediction", timestep_spacing="trailing",
                                                          rescale_betas_zero_snr=True, clip_sample=False, snr_shift_scale=1.0)
    sched.set_timesteps(steps)
    x, y = inputs(seed=9)
    z = x[:1].clone()
    neg, pos = y[:1], y[1:]
    emb = torch.cat([neg, pos], 0)
    (cos, sin), _ = rope_tables()
    gs = []
    with torch.no_grad():
        for t in sched.timesteps:
            zin = torch.cat([z] * 2)
            v = model(zin, emb, t.expand(2), image_rotary_emb=(cos, sin), return_dict=False)[0].float()
            g = 1 + guidance * ((1 - math.cos(math.pi * ((steps - t.item()) / steps) ** 5.0)) / 2)  # use_dynamic_cfg
            gs.append(g)
            unc, txt = v.chunk(2)
            v = unc + g * (txt - unc)
            z = sched.step(v, t, z, return_dict=False)[0]
            z = bf16r(z)  # latents.to(prompt_embeds.dtype) (:723)
    fix = dict(cfg=cfg_for(True), seed=SEED, steps=steps, guidance=guidance, latents=x[:1], neg=neg, pos=pos,
               timesteps=[int(v) for v in sched.timesteps], guidance_per_step=gs, out=z)
    torch.save(pack(fix), os.path.join(OUT, "cogvideox_sample_small.pt"))
    print("sample", tuple(z.shape), float(z.abs().max()), fix["timesteps"])


def make_pab(steps=6):
    sd = weights()
    model = ref_loader.build_reference_cogvideox(cfg_for(True), sd)
    pab_mgr = importlib.import_module("videosys.core.pab.pab_mgr")
    pab_mgr.set_pab_manager(pab_mgr.PABConfig(spatial_broadcast=True, spatial_threshold=[100, 900], spatial_range=3))
    pab_mgr.update_steps(steps)
    x, y = inputs(seed=17)
    (cos, sin), _ = rope_tables()
    ts = CO.ddim_timesteps(steps)
    outs = []
    with torch.no_grad():
        for t in ts:
            outs.append(model(x, y, torch.tensor([t, t]), image_rotary_emb=(cos, sin), return_dict=False)[0])
    pab_mgr.PAB_MANAGER = None
    fix = dict(cfg=cfg_for(True), seed=SEED, steps=steps, x=x, y=y, timesteps=ts, outs=outs,
               pab=dict(spatial_threshold=[100, 900], spatial_range=3))
    torch.save(pack(fix), os.path.join(OUT, "cogvideox_pab_small.pt"))
    print("pab", ts)


if __name__ == "__main__":
    for w in (sys.argv[1:] or ["fwd", "sample", "pab"]):
        {"fwd": make_fwd, "sample": make_sample, "pab": make_pab}[w]()
