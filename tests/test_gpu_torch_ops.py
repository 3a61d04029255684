"""-m gpu: ``torch.ops.videosys_amd.*`` (videosys_amd/torch_ops.py, the PyTorch custom-op registration north_star asks for) runs the
same kernels as the direct C-ABI bindings: bit-identical outputs, in-place semantics (residual stream aliasing ``out``), errors on
CPU tensors."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_custom_ops_match_direct_bindings():
    import videosys_amd.torch_ops as T   # noqa: F401  (registers the namespace)
    from videosys_amd import ops

    tv = torch.ops.videosys_amd
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16).to(dev)
    M, C, H, S, T_, B = 2 * 4 * 64, 576, 8, 64, 4, 2
    x, w, b = rnd(M, C), rnd(3 * C, C, sc=1 / math.sqrt(C)), rnd(3 * C, sc=0.1)
    mod = rnd(B, 6 * C, sc=0.3)
    # adaln + qkv gemm
    xm_a = ops.adaln_modulate(x, mod[0, :C], mod[0, C:2 * C], M // B, 6 * C)
    xm_b = torch.empty_like(x)
    tv.adaln_modulate(x, mod[0, :C], mod[0, C:2 * C], M // B, 6 * C, 1e-6, xm_b)
    assert torch.equal(xm_a, xm_b)
    qkv_a = ops.gemm(xm_a, w, b)
    qkv_b = torch.empty(M, 3 * C, dtype=torch.bfloat16, device=dev)
    tv.gemm(xm_b, w, b, ops.EPI_BIAS, None, 0, 0, None, None, qkv_b)
    assert torch.equal(qkv_a, qkv_b)
    # spatial attention: prep + flash
    qw, kw = rnd(72, sc=0.1) + 1, rnd(72, sc=0.1) + 1
    nf = B * T_
    outs = []
    for route in (0, 1):
        kp, vt = ops.alloc_kv_buffers(nf, H, S, dev)
        ao = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
        if route == 0:
            ops.attn_prep_kv(qkv_a[:, C:2 * C], qkv_a[:, 2 * C:], kw, kp, vt, nf, H, S)
            ops.flash_attn(qkv_a[:, :C], qw, kp, vt, ao, nf, H, S, S)
        else:
            tv.attn_prep_kv(qkv_a[:, C:2 * C], qkv_a[:, 2 * C:], kw, kp, vt, nf, H, S, 1e-6)
            tv.flash_attn(qkv_a[:, :C], qw, kp, vt, ao, nf, H, S, S, 1e-6)
        outs.append(ao)
    assert torch.equal(outs[0], outs[1])
    # temporal attention
    freqs = 1.0 / (10000 ** (torch.arange(0, 72, 2).float() / 72))
    ang = torch.einsum("p,f->pf", torch.arange(T_).float(), freqs).repeat_interleave(2, -1)
    cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
    ta, tb = torch.empty(M, C, dtype=torch.bfloat16, device=dev), torch.empty(M, C, dtype=torch.bfloat16, device=dev)
    ops.attn_temporal(qkv_a, C, qw, kw, cos, sin, ta, B, T_, S, H)
    tv.attn_temporal(qkv_a, C, qw, kw, cos, sin, tb, B, T_, S, H, 1e-6)
    assert torch.equal(ta, tb)
    # projection with gate + residual, in place on the residual stream (res is out), PAB slab written
    wp, bp = rnd(C, C, sc=1 / math.sqrt(C)), rnd(C, sc=0.1)
    ra, rb = x.clone(), x.clone()
    auxa, auxb = torch.empty_like(x), torch.empty_like(x)
    ops.gemm(outs[0], wp, bp, epilogue=ops.EPI_GATE_RES, gate=mod[0, 2 * C:3 * C], gate_stride=6 * C, rows_per_sample=M // B, res=ra,
             aux=auxa, out=ra)
    tv.gemm(outs[1], wp, bp, ops.EPI_GATE_RES, mod[0, 2 * C:3 * C], 6 * C, M // B, rb, auxb, rb)
    assert torch.equal(ra, rb) and torch.equal(auxa, auxb) and not torch.equal(ra, x)
    tv.add_rows(rb, auxb)
    ops.add_rows(ra, auxa)
    assert torch.equal(ra, rb)
    # CFG + Euler update
    z = torch.randn(1, 4, 3, 8, 8, generator=g).to(dev)
    mo = torch.randn(2, 8, 3, 8, 8, generator=g).to(dev)
    za, zb = z.clone(), z.clone()
    ops.cfg_euler_step(za, mo, 7.0, 0.03)
    tv.cfg_euler_step(zb, mo, 7.0, 0.03)
    assert torch.equal(za, zb) and not torch.equal(za, z)
    with pytest.raises(Exception):
        tv.add_rows(torch.zeros(2, 8, dtype=torch.bfloat16), torch.zeros(2, 8, dtype=torch.bfloat16))
