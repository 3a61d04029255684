"""-m gpu: the PyTorch custom-op route (``torch.ops.vsys.launch`` / ``torch.ops.vsys.program_run``, csrc/torch_binding.cpp — the
TORCH_LIBRARY fragment north_star asks the host code to go through) is the PRODUCT path and runs the same kernels as the ctypes
binding of the same extern "C" functions: bit-identical outputs for single launches (tensors handed over as tensors), for entry points
outside the denoise step (VAE), and for a whole recorded denoise step replayed through one dispatcher call per segment."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _routes(fn):
    """Run ``fn`` under the custom-op route and under the ctypes route; returns both results."""
    from videosys_amd import _lib

    tv = _lib.torch_ops()
    assert tv is not None, "libvideosys_torch.so did not load: the product would silently run on the ctypes route"
    a = fn()
    try:
        _lib._torch_ops = None      # (None = "not available": ops._call / Program.run bind through ctypes)
        b = fn()
    finally:
        _lib._torch_ops = tv
    return a, b


def test_single_launches_match_the_ctypes_route():
    from videosys_amd import ops

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16).to(dev)
    M, C, H, S, T_, B = 2 * 4 * 64, 576, 8, 64, 4, 2
    x, w, b = rnd(M, C), rnd(3 * C, C, sc=1 / math.sqrt(C)), rnd(3 * C, sc=0.1)
    mod = rnd(B, 6 * C, sc=0.3)
    qw, kw = rnd(72, sc=0.1) + 1, rnd(72, sc=0.1) + 1
    freqs = 1.0 / (10000 ** (torch.arange(0, 72, 2).float() / 72))
    ang = torch.einsum("p,f->pf", torch.arange(T_).float(), freqs).repeat_interleave(2, -1)
    cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)

    def chain():
        xm = ops.adaln_modulate(x, mod[0, :C], mod[0, C:2 * C], M // B, 6 * C)
        qkv = ops.gemm(xm, w, b)
        kp, vt = ops.alloc_kv_buffers(B * T_, H, S, dev)
        ops.attn_prep_kv(qkv[:, C:2 * C], qkv[:, 2 * C:], kw, kp, vt, B * T_, H, S)
        ao = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
        ops.flash_attn(qkv[:, :C], qw, kp, vt, ao, B * T_, H, S, S)
        at = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
        ops.attn_temporal(qkv, C, qw, kw, cos, sin, at, B, T_, S, H)
        r = x.clone()
        ops.gemm(ao, w[:C], b[:C], epilogue=ops.EPI_GATE_RES, gate=mod[0, 2 * C:3 * C], gate_stride=6 * C, rows_per_sample=M // B, res=r, out=r)
        return xm, qkv, ao, at, r

    for got, want in zip(*_routes(chain)):
        assert torch.equal(got, want)
    # an entry point outside the denoise step (VAE GroupNorm + SiLU: host descriptor arrays travel as integers)
    grid = ops.VaeGrid(1, 2, 8, 8, pad=1)
    buf, rows = grid.alloc(128, dev, zero=True)
    rows.copy_(rnd(grid.rows, 128))
    gamma, beta = rnd(128) + 1, rnd(128)

    def gn():
        out_buf, out_rows = grid.alloc(128, dev, zero=True)
        ops.group_norm(rows, grid, out_rows, grid, 128, gamma, beta, 1e-6, True)
        return out_rows.clone()

    a, b_ = _routes(gn)
    assert torch.equal(a, b_)
    with pytest.raises(ops._lib.VsysError):          # C++ argument validation surfaces as the library's own error type
        ops.gemm(x[:, :100].contiguous(), w[:, :100].contiguous(), None)


def test_recorded_step_replays_through_one_dispatcher_call():
    """A small STDiT3: step 1 records (eager launches through torch.ops.vsys.launch), step 2 replays (torch.ops.vsys.program_run);
    same outputs as the ctypes route, and the replay really is the program path."""
    from oracle import stdit3_oracle as O
    from videosys_amd.stdit3 import STDiT3, STDiT3Config

    cfg = dict(depth=2, hidden_size=576, num_heads=8, caption_channels=64, model_max_length=16)
    sd = O.synth_state_dict(**cfg, seed=7)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 4, 4, 16, 16, generator=g).to(torch.bfloat16).float()
    y = torch.randn(2, 1, 16, 64, generator=g).to(torch.bfloat16).float()
    mask = torch.zeros(1, 16, dtype=torch.long)
    mask[:, :11] = 1
    kw = dict(mask=mask, fps=torch.tensor([24.0, 24.0]), height=torch.tensor([128.0, 128.0]), width=torch.tensor([128.0, 128.0]))
    t = torch.tensor([500.0, 500.0])

    def two_steps():
        model = STDiT3(STDiT3Config(**cfg), device="cuda:0")
        model.load_state_dict(sd)
        o1 = model(x, t, y, **kw).float().cpu()
        o2 = model(x, t, y, **kw).float().cpu()
        return o1, o2, dict(model.program_stats)

    (a1, a2, sa), (b1, b2, sb) = _routes(two_steps)
    assert torch.equal(a1, b1) and torch.equal(a2, b2) and torch.equal(a1, a2)
    assert sa["recorded"] >= 1 and sa["replayed"] >= 1 and sa == sb, (sa, sb)
