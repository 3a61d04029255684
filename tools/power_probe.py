"""Sample rocm-smi (clock, power) while one GEMM shape runs back to back for a few seconds.   python tools/power_probe.py [variant]"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    import __graft_entry__ as ge

    ge.build()
    from videosys_amd import _lib, ops

    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    M, N, K = 38912, 3456, 1152
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    b = torch.zeros(N, dtype=torch.bfloat16, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    samples = []
    stop = False

    def poll():
        while not stop:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True)
            try:
                j = json.loads(r.stdout)
                c = j[sorted(j)[0]]
                samples.append({k: v for k, v in c.items() if "sclk" in k.lower() or "power" in k.lower() or "mclk" in k.lower()})
            except Exception:
                samples.append({"raw": r.stdout[:200]})
            time.sleep(0.2)

    lib.vsys_tune_gemm_variant(variant)
    ops.gemm(x, w, b, out=out)
    torch.cuda.synchronize()
    th = threading.Thread(target=poll)
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 4.0:
        for _ in range(200):
            ops.gemm(x, w, b, out=out)
        torch.cuda.synchronize()
        n += 200
    dt = time.perf_counter() - t0
    stop = True
    th.join()
    lib.vsys_tune_gemm_variant(0)
    print(json.dumps({"variant": variant, "tflops_sustained": round(2.0 * M * N * K * n / dt / 1e12, 1), "samples": samples[1:-1][:12]}))


if __name__ == "__main__":
    main()
