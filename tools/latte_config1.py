#!/usr/bin/env python
"""BASELINE config 1: Latte geometry 256x256x16f (latent [4,16,32,32], 256 tokens/frame, 120 text tokens), 20 DDIM steps,
guidance 7.5, synthetic weights.  Runs the HIP pipeline on cuda:0 (timed) and the CPU oracle beside it for the first
--cpu-steps steps (latent parity + CPU seconds/step on the host cores).  python tools/latte_config1.py [--depth 28]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=28)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--cpu-steps", type=int, default=1)
    ap.add_argument("--pab", action="store_true")
    a = ap.parse_args()
    import __graft_entry__ as ge

    ge.build()
    from oracle import latte_oracle as LO
    from videosys_amd import LatteConfig, LattePABConfig, LattePipeline

    cfg = dict(num_attention_heads=16, attention_head_dim=72, num_layers=a.depth, caption_channels=4096, sample_size=64,
               video_length=16)
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(1, 4, 16, 32, 32, generator=g)
    pos = (torch.randn(1, 120, 4096, generator=g) * 0.1).to(torch.bfloat16).float()
    neg = (torch.randn(1, 120, 4096, generator=g) * 0.1).to(torch.bfloat16).float()
    pmask = torch.ones(1, 120, dtype=torch.long)
    pmask[:, 77:] = 0
    nmask = torch.ones(1, 120, dtype=torch.long)
    nmask[:, 9:] = 0
    pipe = LattePipeline(LatteConfig(model_path="synthetic:4321", transformer_config=cfg, enable_pab=a.pab,
                                     pab_config=LattePABConfig()), device="cuda:0")
    kw = dict(prompt_embeds=pos, negative_prompt_embeds=neg, prompt_mask=pmask, negative_mask=nmask, latents=lat,
              guidance_scale=7.5, output_type="latent", verbose=False)
    pipe.generate(num_inference_steps=2, **kw)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = pipe.generate(num_inference_steps=a.steps, **kw).video
    torch.cuda.synchronize()
    gpu_s = time.perf_counter() - t0
    res = dict(config="latte 256x256x16f", depth=a.depth, steps=a.steps, pab=a.pab, gpu_seconds=gpu_s,
               gpu_ms_per_step=1e3 * gpu_s / a.steps, finite=bool(torch.isfinite(out).all()))
    if a.cpu_steps > 0 and not a.pab:
        torch.set_num_threads(os.cpu_count())
        sd = LO.synth_state_dict(a.depth, 16, 72, seed=4321)
        sd = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
        orc = LO.LatteOracle(sd, a.depth, 16, 72, sample_size=64, video_length=16)
        # oracle for the FIRST cpu_steps steps of the same 20-step schedule
        ac = LO.ddim_tables()
        z = lat.clone()
        t0 = time.perf_counter()
        for t in LO.ddim_timesteps(a.steps)[: a.cpu_steps]:
            o = orc(torch.cat([z, z]), torch.tensor([t, t]), torch.cat([neg, pos]), torch.cat([nmask, pmask]))
            unc, txt = o.chunk(2)
            eps = (unc + 7.5 * (txt - unc))[:, :4]
            c_z, c_eps = LO.ddim_coeffs(t, a.steps, ac)
            z = c_z * z + c_eps * eps
        cpu_s = time.perf_counter() - t0
        # HIP path for the same number of steps of the same schedule
        pipe.scheduler.set_timesteps(a.steps)
        from videosys_amd import ops

        zg = lat.float().to("cuda:0").contiguous().clone()
        for t in pipe.scheduler.timesteps[: a.cpu_steps]:
            o = pipe.transformer(zg, timestep=torch.full((2,), t), encoder_hidden_states=torch.cat([neg, pos]),
                                 encoder_attention_mask=torch.cat([nmask, pmask]), return_dict=False)[0]
            c_z, c_eps = pipe.scheduler.coeffs(t)
            ops.cfg_linear_step(zg, o, 7.5, c_z, c_eps, cond_first=False)
        d = (zg.cpu() - z)
        res.update(cpu_cores=os.cpu_count(), cpu_steps=a.cpu_steps, cpu_seconds_per_step=cpu_s / a.cpu_steps,
                   parity_max_abs=float(d.abs().max()), parity_ref_max=float(z.abs().max()),
                   parity_cosine=float(torch.nn.functional.cosine_similarity(zg.cpu().flatten(), z.flatten(), dim=0)))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
