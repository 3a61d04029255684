"""CPU study (no GPU needed): what folding AdaLN into the following GEMM does to the result.

The unfused path the product runs:   y = bf16(LN(x) * (1 + s) + b);   out = y @ W^T            (bf16 operands, fp32 accumulation)
The folded path:   W' = bf16(W * (1 + s));  acc = x @ W'^T;  out = acc * rstd - (mu * rstd) * colsum(W') + (b @ W^T)
Both are compared with the same expression in fp64 on the same bf16 inputs.  The folded form multiplies the RAW activations into the
weights, so the rounding of W' (2^-9 relative) is amplified by |x| / sigma instead of |x - mu| / sigma and the row mean is removed only
afterwards, by cancellation: rows whose mean or whose outlier channels are large against their spread lose accuracy.  The sweep below
varies exactly that (a per-row offset, a handful of outlier channels).

    python tools/adaln_fold_numerics.py [--rows 2048] [--out profiles/r03_adaln_fold_numerics.json]"""
import argparse
import json

import torch


def bf(t):
    return t.to(torch.bfloat16).to(torch.float64)


def study(rows, C, N, offset, outliers, outlier_scale, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, C, generator=g, dtype=torch.float64)
    if outliers:
        idx = torch.randperm(C, generator=g)[:outliers]
        x[:, idx] += outlier_scale * (1 + 0.1 * torch.randn(rows, outliers, generator=g, dtype=torch.float64))
    x = x + offset * torch.randn(rows, 1, generator=g, dtype=torch.float64)
    x = bf(x)                                                   # the residual stream is bf16
    W = bf(torch.randn(N, C, generator=g, dtype=torch.float64) / C ** 0.5)
    s = bf(0.3 * torch.randn(C, generator=g, dtype=torch.float64))
    b = bf(0.3 * torch.randn(C, generator=g, dtype=torch.float64))
    mu = x.mean(1, keepdim=True)
    rstd = 1.0 / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-6)
    exact = ((x - mu) * rstd * (1 + s) + b) @ W.T
    # unfused: the modulated row is rounded to bf16 once, the GEMM accumulates exactly (fp32 accumulation of <= 1152 products of bf16
    # pairs is exact to ~2^-24 relative: below everything measured here)
    y = bf((x - mu) * rstd * (1 + s) + b)
    unfused = y @ W.T
    # folded
    Wp = bf(W * (1 + s))
    acc = (x @ Wp.T).to(torch.float32).to(torch.float64)       # fp32 accumulator
    colsum = Wp.sum(1)
    cvec = (b @ W.T)
    folded = acc * rstd - (mu * rstd) * colsum + cvec
    rel = lambda a: float(((a - exact).pow(2).mean() / exact.pow(2).mean()).sqrt())
    # the bf16 store of the result is common to both and bounds what can be seen downstream
    store = rel(bf(exact))
    return dict(offset_sigma=offset, outlier_channels=outliers, outlier_scale_sigma=outlier_scale,
                mean_over_sigma=float((mu.abs() * rstd).mean()), max_abs_x_over_sigma=float((x.abs().amax(1, keepdim=True) * rstd).mean()),
                rel_rms_unfused=rel(unfused), rel_rms_folded=rel(folded), rel_rms_bf16_store=store,
                folded_over_unfused=rel(folded) / rel(unfused))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2048)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    rows = []
    for offset, outliers, scale in ((0, 0, 0), (1, 0, 0), (4, 0, 0), (16, 0, 0), (0, 4, 10), (0, 4, 40), (0, 4, 150), (4, 4, 40)):
        r = study(a.rows, 1152, 3456, offset, outliers, scale, seed=11)
        rows.append(r)
        print(json.dumps(r), flush=True)
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(dict(what="AdaLN folded into the qkv GEMM vs the unfused bf16 path, both against fp64 on the same bf16 inputs "
                                "(C = 1152, N = 3456); rel-rms error of the GEMM output", rows=rows), fh, indent=1)


if __name__ == "__main__":
    main()
