"""GPU: the CogVideoX kernels (head_dim 64 attention, two-segment LayerNorm-modulate / gate epilogue, patch im2col,
unpatchify) against plain torch fp32 on bf16-rounded inputs.  Tolerance: max|err| <= 2^-7 * max|ref| per op."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def check(out, ref, rel=2 ** -7, what=""):
    out, ref = out.float().cpu(), ref.float().cpu()
    err = (out - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= rel * scale, f"{what}: max|err| {err:.4e} vs max|ref| {scale:.3f}"


def bf(t):
    return t.to(torch.bfloat16)


def rope_ref(x, cos, sin):
    """modules/embeddings.py:358-412 apply_rotary_emb (use_real, unbind_dim -1) on [..., S, D]."""
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(-2)
    return x * cos + rot * sin


@pytest.mark.parametrize("B,H,Lt,Lv,norm,rope", [(2, 6, 20, 300, True, True), (1, 3, 226, 130, True, False),
                                                  (2, 4, 7, 64, False, True), (1, 2, 30, 1000, True, True)])
def test_flash_attn_d64_joint_sequence(B, H, Lt, Lv, norm, rope):
    from videosys_amd import ops

    D, L = 64, Lt + Lv
    C = H * D
    g = torch.Generator().manual_seed(B * 100 + H)
    qkv = bf(torch.randn(B * L, 3 * C, generator=g))
    qw, qb = bf(1 + 0.2 * torch.randn(D, generator=g)), bf(0.1 * torch.randn(D, generator=g))
    kw, kb = bf(1 + 0.2 * torch.randn(D, generator=g)), bf(0.1 * torch.randn(D, generator=g))
    ang = torch.rand(Lv, D // 2, generator=g) * 6.0
    cos, sin = ang.cos().repeat_interleave(2, -1).contiguous(), ang.sin().repeat_interleave(2, -1).contiguous()
    qd = qkv.to(dev())
    kp, vt = ops.alloc_kv_buffers64(B, H, L, dev())
    cd, sd = (cos.to(dev()), sin.to(dev())) if rope else (None, None)
    ops.attn_prep_kv64(qd[:, C:2 * C], qd[:, 2 * C:], kw.to(dev()) if norm else None, kb.to(dev()) if norm else None, cd, sd, Lt,
                       kp, vt, B, H, L)
    out = torch.empty(B * L, C, dtype=torch.bfloat16, device=dev())
    ops.flash_attn64(qd[:, :C], qw.to(dev()) if norm else None, qb.to(dev()) if norm else None, cd, sd, Lt, kp, vt, out, B, H, L, L)
    q, k, v = [t.float().view(B, L, H, D).transpose(1, 2) for t in qkv.split(C, dim=1)]
    if norm:
        q = bf(torch.nn.functional.layer_norm(q, (D,), qw.float(), qb.float(), 1e-6)).float()
        k = bf(torch.nn.functional.layer_norm(k, (D,), kw.float(), kb.float(), 1e-6)).float()
    if rope:
        q = torch.cat([q[:, :, :Lt], bf(rope_ref(q[:, :, Lt:], cos, sin)).float()], 2)
        k = torch.cat([k[:, :, :Lt], bf(rope_ref(k[:, :, Lt:], cos, sin)).float()], 2)
    ref = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(D), -1) @ v
    check(out, ref.transpose(1, 2).reshape(B * L, C), what=f"flash d64 B{B} H{H} L{Lt}+{Lv}")
    # the shipped kernel runs a three-stage K/V ring (counted vmcnt(4), raw barrier); flash variant 12 selects the two-stage ring:
    # the same arithmetic in the same order, so the same bits — for one tile, two tiles (no steady-state iteration) and many
    from videosys_amd import _lib

    for fv, what in ((12, "two-stage"),):
        out3 = torch.empty_like(out)
        try:
            assert _lib.load().vsys_tune_flash_variant(fv) == 0
            ops.flash_attn64(qd[:, :C], qw.to(dev()) if norm else None, qb.to(dev()) if norm else None, cd, sd, Lt, kp, vt, out3, B, H, L, L)
            torch.cuda.synchronize()
        finally:
            _lib.load().vsys_tune_flash_variant(0)
        assert torch.equal(out3, out), f"{what} ring (d64) differs from the shipped three-stage kernel"


@pytest.mark.parametrize("B,H,Lt,Lv,norm,rope", [(1, 3, 226, 2100, True, True), (2, 2, 20, 300, True, True), (1, 2, 7, 2553, False, True),
                                                  (1, 2, 0, 2304, True, False)])
def test_flash_attn_d64_w64_stream(B, H, Lt, Lv, norm, rope):
    """The 64-rows-per-wave instruction stream generated for head_dim 64 (attention64_w64.hip; default from 2048 keys, flash variant
    14 forces it from 256): against fp32 torch on the same bf16 inputs and against the 32-row kernel (variant 15) — same P
    rounding, the row sum on the matrix pipe instead of the VALU, so equal to within one bf16 step of the output.  Ragged last
    tiles (2326 = 36 x 64 + 22; 320 = 5 tiles), query rows past the last 256-block, with / without norm and RoPE."""
    from videosys_amd import _lib, ops

    D, L = 64, Lt + Lv
    C = H * D
    g = torch.Generator().manual_seed(B * 100 + H + L)
    qkv = bf(torch.randn(B * L, 3 * C, generator=g))
    qw, qb = bf(1 + 0.2 * torch.randn(D, generator=g)), bf(0.1 * torch.randn(D, generator=g))
    kw, kb = bf(1 + 0.2 * torch.randn(D, generator=g)), bf(0.1 * torch.randn(D, generator=g))
    ang = torch.rand(max(Lv, 1), D // 2, generator=g) * 6.0
    cos, sin = ang.cos().repeat_interleave(2, -1).contiguous(), ang.sin().repeat_interleave(2, -1).contiguous()
    qd = qkv.to(dev())
    kp, vt = ops.alloc_kv_buffers64(B, H, L, dev())
    cd, sd = (cos.to(dev()), sin.to(dev())) if rope else (None, None)
    nq, nb = (qw.to(dev()), qb.to(dev())) if norm else (None, None)
    ops.attn_prep_kv64(qd[:, C:2 * C], qd[:, 2 * C:], kw.to(dev()) if norm else None, kb.to(dev()) if norm else None, cd, sd, Lt,
                       kp, vt, B, H, L)
    outs = {}
    lib = _lib.load()
    kbound = ops.ln_key_bound(qw, qb, kw, kb) if norm else None   # (no norm: nothing can be promised about the keys)
    try:
        for fv in (14, 15) + ((17,) if kbound else ()):
            assert lib.vsys_tune_flash_variant(fv) == 0
            o = torch.full((B * L, C), float("nan"), dtype=torch.bfloat16, device=dev())
            ops.flash_attn64(qd[:, :C], nq, nb, cd, sd, Lt, kp, vt, o, B, H, L, L, k_norm_bound=kbound if fv == 17 else None)
            torch.cuda.synchronize()
            outs[fv] = o
    finally:
        lib.vsys_tune_flash_variant(0)
    q, k, v = [t.float().view(B, L, H, D).transpose(1, 2) for t in qkv.split(C, dim=1)]
    if norm:
        q = bf(torch.nn.functional.layer_norm(q, (D,), qw.float(), qb.float(), 1e-6)).float()
        k = bf(torch.nn.functional.layer_norm(k, (D,), kw.float(), kb.float(), 1e-6)).float()
    if rope:
        q = torch.cat([q[:, :, :Lt], bf(rope_ref(q[:, :, Lt:], cos, sin)).float()], 2)
        k = torch.cat([k[:, :, :Lt], bf(rope_ref(k[:, :, Lt:], cos, sin)).float()], 2)
    ref = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(D), -1) @ v).transpose(1, 2).reshape(B * L, C)
    check(outs[14], ref, what=f"flash d64 w64 B{B} H{H} L{Lt}+{Lv}")
    a, b_ = outs[14].float().cpu(), outs[15].float().cpu()
    assert torch.isfinite(a).all()
    err14, err15 = (a - ref).abs().max().item(), (b_ - ref).abs().max().item()
    assert err14 <= 1.25 * err15 + 1e-3, (err14, err15)
    assert (a - b_).abs().max().item() <= 2.0 ** -7 * max(1.0, ref.abs().max().item())
    if 17 in outs:   # the same stream without the running max (the caller's bound on the key norms): same tolerance
        c = outs[17].float().cpu()
        assert torch.isfinite(c).all()
        check(outs[17], ref, what=f"flash d64 w64 without running max B{B} H{H} L{Lt}+{Lv}")
        assert (c - ref).abs().max().item() <= 1.25 * err15 + 1e-3


def test_ln_modulate_two_segments_and_plain():
    from videosys_amd import ops

    B, Lt, Lv, C = 2, 5, 19, 3072
    L = Lt + Lv
    g = torch.Generator().manual_seed(0)
    x = bf(torch.randn(B * L, C, generator=g) * 2 + 0.3)
    w, b = bf(1 + 0.1 * torch.randn(C, generator=g)), bf(0.1 * torch.randn(C, generator=g))
    mod = bf(torch.randn(B, 6 * C, generator=g) * 0.3)  # shift, scale, gate, enc_shift, enc_scale, enc_gate
    md = mod.to(dev())
    y = ops.ln_modulate(x.to(dev()), w.to(dev()), b.to(dev()), md[0, 0:C], md[0, C:2 * C], L, mod_stride=6 * C, seg_split=Lt,
                        mod_alt=3 * C, eps=1e-5)
    ln = torch.nn.functional.layer_norm(x.float(), (C,), w.float(), b.float(), 1e-5).view(B, L, C)
    m = mod.float()
    ref = torch.cat([ln[:, :Lt] * (1 + m[:, None, 4 * C:5 * C]) + m[:, None, 3 * C:4 * C],
                     ln[:, Lt:] * (1 + m[:, None, C:2 * C]) + m[:, None, 0:C]], 1)
    check(y, ref.reshape(B * L, C), what="ln_modulate two segments")
    y2 = ops.ln_modulate(x.to(dev()), w.to(dev()), b.to(dev()), None, None, L, eps=1e-5)
    check(y2, ln.reshape(B * L, C), what="plain affine LayerNorm")
    xs = x[:, :1920].contiguous()
    y3 = ops.ln_modulate(xs.to(dev()), None, None, None, None, L, eps=1e-6)
    check(y3, torch.nn.functional.layer_norm(xs.float(), (1920,), None, None, 1e-6), what="LayerNorm no affine")


@pytest.mark.parametrize("variant", [0, 8, 20])
def test_gemm_gate2_and_gate_add_rows(variant):
    from videosys_amd import _lib

    assert _lib.load().vsys_tune_gemm_variant(variant) == 0
    try:
        _gemm_gate2_and_gate_add_rows()
    finally:
        _lib.load().vsys_tune_gemm_variant(0)


def _gemm_gate2_and_gate_add_rows():
    from videosys_amd import ops

    B, Lt, Lv, C, K = 2, 30, 500, 576, 192
    L = Lt + Lv
    g = torch.Generator().manual_seed(1)
    x = bf(torch.randn(B * L, K, generator=g))
    w = bf(torch.randn(C, K, generator=g) / math.sqrt(K))
    bias = bf(torch.randn(C, generator=g) * 0.1)
    res = bf(torch.randn(B * L, C, generator=g))
    mod = bf(torch.randn(B, 6 * C, generator=g))
    md = mod.to(dev())
    xr = res.to(dev()).clone()
    ops.gemm_gate2(x.to(dev()), w.to(dev()), bias.to(dev()), md[0, 2 * C:3 * C], 6 * C, L, Lt, 3 * C, res=xr, out=xr)
    u = (x.float() @ w.float().t() + bias.float()).view(B, L, C)
    m = mod.float()
    gated = torch.cat([u[:, :Lt] * m[:, None, 5 * C:6 * C], u[:, Lt:] * m[:, None, 2 * C:3 * C]], 1).reshape(B * L, C)
    check(xr, res.float() + gated, what="gemm two-segment gate + residual")
    y = bf(u.reshape(B * L, C))
    xr2 = res.to(dev()).clone()
    ops.gate_add_rows(xr2, y.to(dev()), md[0, 2 * C:3 * C], L, 6 * C, Lt, 3 * C)
    yv = y.float().view(B, L, C)
    g2 = torch.cat([yv[:, :Lt] * m[:, None, 5 * C:6 * C], yv[:, Lt:] * m[:, None, 2 * C:3 * C]], 1).reshape(B * L, C)
    check(xr2, res.float() + g2, what="gate_add_rows")


def test_im2col_patch_and_unpatchify_roundtrip():
    from videosys_amd import ops

    Bz, F, Cin, H, W, p = 1, 3, 16, 12, 20, 2
    g = torch.Generator().manual_seed(2)
    z = bf(torch.randn(Bz, F, Cin, H, W, generator=g)).float()
    cols = ops.im2col_patch(z.to(dev()), 2, p).cpu().float()  # B = 2 reads z twice (CFG duplicate)
    w = torch.randn(32, Cin, p, p, generator=g)
    ref = torch.nn.functional.conv2d(z.view(Bz * F, Cin, H, W), w, stride=p).flatten(2).transpose(1, 2).reshape(-1, 32)
    got = cols @ w.view(32, -1).t()
    torch.testing.assert_close(got[: ref.shape[0]], ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(got[ref.shape[0]:], ref, rtol=1e-4, atol=1e-4)
    # unpatchify inverts the patch gather: feed the columns back (Cout = Cin)
    x = torch.zeros(cols.shape[0], 192, dtype=torch.bfloat16)
    x[:, :64] = cols.to(torch.bfloat16)
    out = ops.unpatchify_cvx(x.to(dev()), 2, F, H // p, W // p, Cin, p).cpu()
    assert torch.equal(out[0], z[0]) and torch.equal(out[1], z[0])
