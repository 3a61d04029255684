"""TEST INFRASTRUCTURE ONLY — mints tests/golden/cogvideox_dpm_small.pt from the reference's CogVideoXDPMScheduler.

    python oracle/make_golden_dpm.py       (build container only: needs /root/reference)

The reference's own schedulers/scheduling_dpm_cogvideox.py:119-483 (in-tree; only ``randn_tensor`` is a restated diffusers leaf,
oracle/diffusers_stub.py) is driven the way pipeline_cogvideox.py:679-721 drives it: fp32 latents, a seeded CPU generator, the
previous step's x0 prediction and timestep handed back in, for the 2b and the 5b scheduler settings.  The "model" is a fixed
synthetic function of (latents, t) so the fixture needs no network: what is pinned is the scheduler's arithmetic and the ORDER of
its noise draws.
"""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_loader  # noqa: E402
from oracle.make_golden import OUT  # noqa: E402

SHAPE = (2, 3, 16, 4, 6)


def fake_model(z: torch.Tensor, t: int) -> torch.Tensor:
    """A deterministic stand-in for the guided velocity prediction: smooth in z, different at every timestep."""
    return torch.tanh(z * 0.7 + 0.001 * t) * 0.9 - 0.1 * z.roll(1, dims=-1)


def run(sched, steps: int, seed: int):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(SHAPE, generator=g)
    sched.set_timesteps(steps)
    ts = [int(t) for t in sched.timesteps]
    old, traj, x0s = None, [], []
    for i, t in enumerate(ts):
        v = fake_model(z, t)
        z, old = sched.step(v, old, t, ts[i - 1] if i > 0 else None, z, generator=g, return_dict=False)
        traj.append(z.clone())
        x0s.append(old.clone())
    return dict(timesteps=ts, traj=traj, x0=x0s)


def main():
    out = {"shape": SHAPE, "cases": {}}
    for name, kw, steps, seed in (("5b", dict(snr_shift_scale=1.0), 6, 11), ("2b", dict(snr_shift_scale=3.0), 5, 12),
                                  ("leading", dict(snr_shift_scale=3.0, timestep_spacing="leading", rescale_betas_zero_snr=False), 4, 13)):
        base = dict(prediction_type="v_prediction", timestep_spacing="trailing", rescale_betas_zero_snr=True)
        base.update(kw)
        sched = ref_loader.load_reference_cogvideox_dpm_scheduler(**base)
        case = run(sched, steps, seed)
        case.update(kwargs=base, steps=steps, seed=seed)
        out["cases"][name] = case
    torch.save(out, os.path.join(OUT, "cogvideox_dpm_small.pt"))
    print("wrote cogvideox_dpm_small.pt:", {k: (v["timesteps"], float(v["traj"][-1].abs().mean())) for k, v in out["cases"].items()})


if __name__ == "__main__":
    main()
