"""TEST INFRASTRUCTURE ONLY — mints tests/golden/stdit3_xmask_small.pt from the REAL reference (needs /root/reference):

    python oracle/make_golden_xmask.py

Image / video conditioning of Open-Sora (the callers either side of the denoise step):
  * STDiT3.forward with ``x_mask`` [B, T] (open_sora_transformer_3d.py:181-184,198-200,220-222,262-273,578-582,622 and
    T2IFinalLayer :75-87): frames whose mask is False see the timestep-0 modulation;
  * RFLOW.sample with ``mask`` (scheduling_rflow_open_sora.py:215-236,254-255): conditioning frames are held, join the
    denoising when mask * 1000 >= t and are noised once at that step (noise from the global CPU generator, seeded here);
  * apply_mask_strategy / parse_mask_strategy / find_nearest_point (pipeline_open_sora.py:795-854) on a few strategies.
Same small model, seeds and rounding rules as oracle/make_golden.py.
"""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_loader  # noqa: E402
from oracle import stdit3_oracle as O  # noqa: E402
from oracle.make_golden import OUT, SMALL_CFG, SMALL_SEED, bf16r, pack, round_sd, sd_checksum, small_inputs  # noqa: E402

NOISE_SEED = 2024


def main():
    sd = round_sd(O.synth_state_dict(**SMALL_CFG, seed=SMALL_SEED))
    ref = ref_loader.build_reference_stdit3(SMALL_CFG, sd)
    mods = ref_loader.load_reference_modules()
    rf = mods["rflow"]

    # ---- forward with a different mask per sample
    inp = small_inputs()
    x_mask = torch.tensor([[1, 0, 1, 1, 0], [0, 1, 1, 0, 1]], dtype=torch.bool)
    with torch.no_grad():
        out = ref(inp["x"], inp["timestep"], inp["y"], mask=inp["mask"], x_mask=x_mask, fps=inp["fps"],
                  height=inp["height"], width=inp["width"])
        out_all_true = ref(inp["x"], inp["timestep"], inp["y"], mask=inp["mask"], x_mask=torch.ones(2, 5, dtype=torch.bool),
                           fps=inp["fps"], height=inp["height"], width=inp["width"])

    # ---- masked sampling: frame 0 is a held reference frame, frame 1 an "edited" one (ratio 0.6: joins once t <= 600)
    inp2 = small_inputs(seed=11)
    z = inp2["x"][:1].clone()
    y = inp2["y"][:1].clone()
    y_null = bf16r(sd["y_embedder.y_embedding"])[None, None]
    cond = torch.tensor([[0.0, 0.6, 1.0, 1.0, 1.0]])
    sched = rf.RFLOW(num_sampling_steps=5, cfg_scale=7.0, use_timestep_transform=True)
    margs = dict(y=y, mask=inp2["mask"], height=inp2["height"][:1], width=inp2["width"][:1],
                 num_frames=torch.tensor([17.0]), fps=inp2["fps"][:1])
    x_masks = []
    orig = ref.forward

    def spy(*a, **k):
        x_masks.append(k["x_mask"].clone())
        return orig(*a, **k)

    ref.forward = spy
    torch.manual_seed(NOISE_SEED)
    with torch.no_grad():
        z_out = sched.sample(ref, z.clone(), margs, y_null, device="cpu", progress=False, mask=cond.clone())
    ref.forward = orig

    # ---- the mask-strategy helpers of the pipeline (module-level functions: no pipeline instance needed)
    # (the module itself does not import here — ftfy / bs4 / torchvision are absent — so the four definitions are compiled
    # from the reference file where it lies; nothing is copied into the repo)
    import ast
    from types import SimpleNamespace

    path = "/root/reference/videosys/pipelines/open_sora/pipeline_open_sora.py"
    src = open(path).read()
    want = {"MASK_DEFAULT", "parse_mask_strategy", "find_nearest_point", "apply_mask_strategy"}
    ns = {"torch": torch}
    for node in ast.parse(src).body:
        name = getattr(node, "name", None) or (node.targets[0].id if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name) else None)
        if name in want:
            exec(compile(ast.Module([node], []), path, "exec"), ns)
    pl = SimpleNamespace(**{k: ns[k] for k in want})
    g = torch.Generator().manual_seed(5)
    cases = []
    for ms, loop_i, align, Tz, Tref in (("0", 0, 5, 15, 1), ("0,0,0,0,1,0", 0, None, 15, 7), ("0,0,-5,0,5,0.3", 0, 5, 15, 10),
                                        ("0,0,0,0,1;0,1,0,-3,3,0.5", 0, None, 15, 4), ("1,0,-5,0,5", 1, 5, 15, 15),
                                        ("1,0,-5,0,5", 0, 5, 15, 15), ("", 0, 5, 15, 3), ("0,0,2,7,4", 0, 5, 15, 9)):
        zz = torch.randn(1, 4, Tz, 2, 2, generator=g)
        refs = [[torch.randn(4, Tref, 2, 2, generator=g), torch.randn(4, Tref + 2, 2, 2, generator=g)]]
        z_in = zz.clone()
        masks = pl.apply_mask_strategy(zz, refs, [ms], loop_i, align=align)
        cases.append(dict(ms=ms, loop_i=loop_i, align=align, z_in=z_in, refs=refs, z_out=zz.clone(), masks=masks,
                          parsed=pl.parse_mask_strategy(ms)))
    nearest = [(v, p, m, pl.find_nearest_point(v, p, m)) for v in range(0, 16) for p in (5,) for m in (15, 10, 5)]

    fix = dict(cfg=SMALL_CFG, seed=SMALL_SEED, sd_checksum=sd_checksum(sd), inputs=inp, x_mask=x_mask, out=out,
               out_all_true=out_all_true,
               sample=dict(z0=z, y=y, y_null=y_null, mask=inp2["mask"], height=inp2["height"][:1], width=inp2["width"][:1],
                           num_frames=torch.tensor([17.0]), fps=inp2["fps"][:1], steps=5, cfg_scale=7.0, cond_mask=cond,
                           noise_seed=NOISE_SEED, z_out=z_out, x_masks=torch.stack(x_masks),
                           all_timesteps=margs["all_timesteps"]),
               mask_strategy=cases, nearest=nearest)
    torch.save(pack(fix), os.path.join(OUT, "stdit3_xmask_small.pt"))
    print("stdit3_xmask_small", out.shape, float((out - out_all_true).abs().max()), [m.tolist() for m in x_masks][:5],
          float((z_out - z).abs().max()))


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
