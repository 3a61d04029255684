#!/usr/bin/env python
"""CogVideoX transformer step timing on cuda:0 (BASELINE config 5 geometry: 720x480x49f -> latent [13,16,60,90], 17 550 video
+ 226 text tokens, CFG batch 2).  python tools/cogvideox_bench.py [--model 5b|2b] [--layers N] [--steps K] [--pab]"""
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
    ap.add_argument("--model", default="5b")
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--pab", action="store_true")
    ap.add_argument("--flash-variant", type=int, default=0, help="vsys_tune_flash_variant id (A/B; 0 = shipped dispatch)")
    a = ap.parse_args()
    import __graft_entry__ as ge

    ge.build()
    if a.flash_variant:
        from videosys_amd import _lib

        assert _lib.load().vsys_tune_flash_variant(a.flash_variant) == 0
    from videosys_amd import CogVideoXConfig, CogVideoXPABConfig, CogVideoXPipeline

    geo = dict(num_attention_heads=48, num_layers=42, use_rotary_positional_embeddings=True) if a.model == "5b" else \
        dict(num_attention_heads=30, num_layers=30, use_rotary_positional_embeddings=False)
    if a.layers:
        geo["num_layers"] = a.layers
    pipe = CogVideoXPipeline(CogVideoXConfig(model_path=f"THUDM/CogVideoX-{a.model}@synthetic:777", transformer_config=geo,
                                             enable_pab=a.pab, pab_config=CogVideoXPABConfig()), device="cuda:0")
    g = torch.Generator().manual_seed(0)
    pos = (torch.randn(1, 226, 4096, generator=g) * 0.1).to(torch.bfloat16).float()
    neg = (torch.randn(1, 226, 4096, generator=g) * 0.1).to(torch.bfloat16).float()
    kw = dict(prompt_embeds=pos, negative_prompt_embeds=neg, height=480, width=720, num_frames=49, guidance_scale=6.0,
              use_dynamic_cfg=True, seed=0, output_type="latent", verbose=False)
    pipe.generate(num_inference_steps=2, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = pipe.generate(num_inference_steps=a.steps, **kw).video
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    L, C, H = geo["num_layers"], geo["num_attention_heads"] * 64, geo["num_attention_heads"]
    N = 2 * (17550 + 226)
    flops = L * (2 * N * C * (3 * C + C + 8 * C) + 4 * N * (N // 2) * C)
    print(json.dumps(dict(model=a.model, layers=L, steps=a.steps, pab=a.pab, ms_per_step=1e3 * dt / a.steps,
                          tflops_per_step=flops / 1e12, achieved_tflops=flops / (dt / a.steps) / 1e12,
                          finite=bool(torch.isfinite(out).all()))))


if __name__ == "__main__":
    main()
