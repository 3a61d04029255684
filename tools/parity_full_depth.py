#!/usr/bin/env python
"""Full-depth latent parity tables (the numbers tests/test_gpu_fulldepth.py asserts on), written as JSON for profiles/.
    python tools/parity_full_depth.py > gpurun_out/parity_full_depth.json"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import contextlib

    import __graft_entry__ as ge

    with contextlib.redirect_stdout(sys.stderr):   # keep stdout pure JSON
        ge.build()
    import fulldepth_util as U

    res = {"method": U.__doc__.split("Method.")[1].strip()}
    t0 = time.time()
    hip, ref, floor, y_null = U.opensora_models(depth=28, seed=1234)
    if "--720p128f" in sys.argv:   # configs[3] geometry, all 28 pairs, one step: 2 x 38 x 3600 = 273 600 token rows
        z, y, mask, geom = U.opensora_inputs(T=38, HW=(90, 160, 128), L=300)
        res["geometry"] = "720p x 128f: latent [4, 38, 90, 160], CFG batch 2 = 273 600 token rows, 300 text tokens, depth 28"
        r = U.opensora_one_step(hip, ref, floor, y_null, z, y, mask, geom, t_value=700.0)
        res["configs3_geometry_one_step"] = r
        res["configs3_geometry_one_step_verdict"] = U.verdict(r["out_hip"], r["out_floor"]) or "within tolerance"
        res["seconds"] = time.time() - t0
        print(json.dumps(res, indent=1))
        return
    z, y, mask, geom = U.opensora_inputs(T=19, HW=64, L=300)
    z5, y5, mask5, geom5 = U.opensora_inputs(T=5, HW=64, L=120)

    def section(name, fn, a, b):
        import traceback
        try:
            t1 = time.time()
            res[name] = fn()
            res[name]["seconds"] = time.time() - t1
            res[name + "_verdict"] = U.verdict(res[name][a], res[name][b]) or "within tolerance"
        except Exception:
            res[name] = {"error": traceback.format_exc()}

    section("config2_one_step", lambda: U.opensora_one_step(hip, ref, floor, y_null, z, y, mask, geom, t_value=700.0), "out_hip", "out_floor")
    section("config2_rflow3", lambda: U.opensora_rflow(hip, ref, floor, y_null, z, y, mask, geom, steps=3), "z_hip", "z_floor")
    section("config3_pab30_T5", lambda: U.opensora_pab_schedule(hip, ref, floor, y_null, z5, y5, mask5, geom5, steps=30), "z_hip", "z_floor")
    del hip, ref, floor
    torch.cuda.empty_cache()
    section("latte_config1_one_step", lambda: U.latte_config1(depth=28), "out_hip", "out_floor")
    res["seconds"] = time.time() - t0
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
