"""Per-shape timing of ops.linear_skinny (T5-XXL linears at 300 tokens) over the K-split count, GEMM and reduce separately.
    python tools/skinny_probe.py [--rows 300]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=300)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--narrow", action="store_true", help="the 128-column kernel form (panel shared by three column-tile workgroups)")
    args = ap.parse_args()
    from videosys_amd import _lib, ops

    dev = torch.device("cuda:0")
    M = args.rows
    Mp = (M + 383) // 384 * 384
    g = torch.Generator(device=dev).manual_seed(0)
    for name, N, K in (("qkv", 12288, 4096), ("o", 4096, 4096), ("wi", 20480, 4096), ("wo", 4096, 10240)):
        x = (torch.randn(Mp, K, generator=g, device=dev)).to(torch.bfloat16)
        # several weight copies so that consecutive iterations do not find the weight in the 256 MB Infinity Cache
        ws = [(torch.randn(N, K, generator=g, device=dev) / K ** 0.5).to(torch.bfloat16) for _ in range(6)]
        res = torch.zeros(Mp, N, dtype=torch.bfloat16, device=dev)
        out = torch.empty(Mp, N, dtype=torch.bfloat16, device=dev)
        row = {"shape": f"{name} {M}x{N}x{K}", "form": "128-column kernel" if args.narrow else "256x384 tile", "weight_mb": round(N * K * 2 / 1e6, 1),
               "auto_split": ops.skinny_split(N, Mp, K, wide=not args.narrow)}
        for S in ((1, 2, 4, 8) if args.narrow else (2, 3, 4, 5, 8, 10, 16)):
            if (args.narrow and K % (S * 32)) or K // S < 256:
                continue
            part = torch.empty(S * N * Mp, dtype=torch.float32, device=dev)
            pv = part.view(S, N, Mp)
            Ks = K // S
            tg, tr = [], []
            for it in range(args.iters):
                w = ws[it % len(ws)]
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                st = torch.cuda.current_stream().cuda_stream
                e[0].record()
                if args.narrow:
                    ops.gemm128(w, x, out_f32=pv, batch=S, batch_a=Ks, batch_w=Ks, batch_o=N * Mp, M=N, K=Ks)
                    e[1].record()
                    _lib.check(_lib.load().vsys_splitk_reduce_t(pv.data_ptr(), S, N * Mp, Mp, res.data_ptr(), res.stride(0), out.data_ptr(),
                                                                out.stride(0), M, N, st), "reduce")
                else:
                    _lib.check(_lib.load().vsys_gemm_skinny_slices(w.data_ptr(), K, x.data_ptr(), K, part.data_ptr(), M, Mp, N, K, S, st), "gemm")
                    e[1].record()
                    _lib.check(_lib.load().vsys_splitk_reduce(part.data_ptr(), S, Mp * N, N, res.data_ptr(), res.stride(0), out.data_ptr(),
                                                              out.stride(0), M, N, st), "reduce")
                e[2].record()
                torch.cuda.synchronize()
                tg.append(e[0].elapsed_time(e[1]) * 1e3)
                tr.append(e[1].elapsed_time(e[2]) * 1e3)
            tg, tr = sorted(tg[2:]), sorted(tr[2:])
            row[f"S{S}"] = {"gemm_us": round(tg[len(tg) // 2], 1), "reduce_us": round(tr[len(tr) // 2], 1),
                            "weight_tb_s": round(N * K * 2 / (tg[len(tg) // 2] + tr[len(tr) // 2]) / 1e6, 2)}
        # the direct many-rows form for reference
        td = []
        for it in range(args.iters):
            w = ws[it % len(ws)]
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.gemm128(x[:M], w, res=res[:M])
            b.record()
            torch.cuda.synchronize()
            td.append(a.elapsed_time(b) * 1e3)
        td = sorted(td[2:])
        row["direct_us"] = round(td[len(td) // 2], 1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
