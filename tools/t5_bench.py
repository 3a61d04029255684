"""Time the T5-v1.1-XXL encoder (the reference's "DeepFloyd/t5-v1_1-xxl": 24 layers, d_model 4096, d_ff 10240, 64 heads) on one
prompt of 300 tokens, random weights generated on the device.   python tools/t5_bench.py [--tokens 300] [--batch 1]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=300)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--ab", action="store_true", help="also time the direct (many-rows) GEMM path in the same process, interleaved")
    args = ap.parse_args()
    from videosys_amd.t5 import T5Encoder

    dev = torch.device("cuda:0")
    enc = T5Encoder(device=dev).init_random_(0)
    c = enc.config
    ids = torch.randint(0, c.vocab_size, (args.batch, args.tokens))
    mask = torch.ones(args.batch, args.tokens, dtype=torch.long)
    mask[:, 120:] = 0
    out = enc(ids, mask).last_hidden_state
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    def timed():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        enc(ids, mask)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    ts = []
    modes = {"direct_gemm+valu_attention": (0, False), "weight_streaming+valu_attention": (enc.skinny_rows, False)}
    ab = {k: [] for k in modes}
    keep = (enc.skinny_rows, enc.mfma_attention)
    for _ in range(args.iters):
        ts.append(timed())
        if args.ab:   # the earlier forms in the same process, interleaved
            for k, (rows, mf) in modes.items():
                enc.skinny_rows, enc.mfma_attention = rows, mf
                timed()
                ab[k].append(timed())
            enc.skinny_rows, enc.mfma_attention = keep
            timed()
    if args.ab:
        print(json.dumps({**{k + "_sec": round(min(v), 5) for k, v in ab.items()}, "shipped_sec": round(min(ts), 5)}))
    params = sum(v.numel() for k, v in enc.w.items())
    flops = 2.0 * args.batch * args.tokens * (params - enc.w["emb"].numel())
    print(json.dumps({"workload": f"T5-v1.1-XXL encoder, batch {args.batch} x {args.tokens} tokens", "sec": round(min(ts), 5),
                      "params_b": round(params / 1e9, 2), "weight_gb": round(params * 2 / 1e9, 2),
                      "weight_stream_tb_s": round((params - enc.w["emb"].numel()) * 2 / min(ts) / 1e12, 2),
                      "tflops": round(flops / min(ts) / 1e12, 1)}))


if __name__ == "__main__":
    main()
