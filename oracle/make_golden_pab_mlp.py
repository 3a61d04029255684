#!/usr/bin/env python
"""TEST INFRASTRUCTURE — mints tests/golden/stdit3_pab_mlp_small.pt: the REAL reference STDiT3 (small geometry) run with the full
Pyramid Attention Broadcast incl. the MLP broadcast (pab_mgr.py:93-174, open_sora_transformer_3d.py:232-280).

The reference's STDiT3.forward never hands ``all_timesteps`` to its blocks (open_sora_transformer_3d.py:608-613), so with
``mlp_broadcast=True`` it raises TypeError inside ``_is_t_in_skip_config``.  The block code itself is complete;
here every block's ``forward`` is wrapped to receive the schedule — the one-line fix the model is missing — and nothing else
is touched.     python oracle/make_golden_pab_mlp.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_loader  # noqa: E402
from oracle import stdit3_oracle as O  # noqa: E402
from oracle.make_golden import OUT, SMALL_CFG, SMALL_SEED, pack, round_sd, small_inputs  # noqa: E402

TIMESTEPS = [980, 900, 800, 704, 640, 400, 96]           # bf16-exact; the schedule the sampler would hand down
MLP_SPATIAL = {900: {"block": [0, 1], "skip_count": 2}, 640: {"block": [1], "skip_count": 1}}
MLP_TEMPORAL = {800: {"block": [0], "skip_count": 2}}
ATTN = dict(spatial=(100, 930, 2), temporal=(100, 930, 2), cross=(100, 930, 3))


def main():
    sd = round_sd(O.synth_state_dict(**SMALL_CFG, seed=SMALL_SEED))
    ref = ref_loader.build_reference_stdit3(SMALL_CFG, sd)
    for blk in list(ref.spatial_blocks) + list(ref.temporal_blocks):
        inner = blk.forward
        blk.forward = (lambda f: lambda *a, **k: f(*a, all_timesteps=TIMESTEPS, **k))(inner)
    mods = ref_loader.load_reference_modules()
    pab = mods["pab_mgr"]
    cfg = pab.PABConfig(
        cross_broadcast=True, cross_threshold=list(ATTN["cross"][:2]), cross_range=ATTN["cross"][2],
        spatial_broadcast=True, spatial_threshold=list(ATTN["spatial"][:2]), spatial_range=ATTN["spatial"][2],
        temporal_broadcast=True, temporal_threshold=list(ATTN["temporal"][:2]), temporal_range=ATTN["temporal"][2],
        mlp_broadcast=True, mlp_spatial_broadcast_config=MLP_SPATIAL, mlp_temporal_broadcast_config=MLP_TEMPORAL)
    pab.set_pab_manager(cfg)
    pab.update_steps(len(TIMESTEPS))
    inp = small_inputs()
    outs = []
    with torch.no_grad():
        for t in TIMESTEPS:
            tt = torch.tensor([float(t), float(t)])
            outs.append(ref(inp["x"], tt, inp["y"], mask=inp["mask"], fps=inp["fps"], height=inp["height"], width=inp["width"]))
    left = (len(cfg.mlp_spatial_outputs), len(cfg.mlp_temporal_outputs))
    pab.PAB_MANAGER = None
    torch.save(pack(dict(cfg=SMALL_CFG, seed=SMALL_SEED, inputs=inp, timesteps=TIMESTEPS, steps=len(TIMESTEPS), pab=ATTN,
                         mlp_spatial=MLP_SPATIAL, mlp_temporal=MLP_TEMPORAL, stored_left=left, outs=torch.stack(outs))),
               os.path.join(OUT, "stdit3_pab_mlp_small.pt"))
    print("stdit3_pab_mlp_small", len(outs), "entries left in the stores:", left)


if __name__ == "__main__":
    main()
