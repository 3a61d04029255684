"""TEST INFRASTRUCTURE: torch restatements, on CPU tensors, of the VAE kernel wrappers of videosys_amd.ops (same signatures, same
row / grid layouts, bf16 storage, fp32 arithmetic) so that the HOST composition of vae_open_sora.py — which grids, paddings,
strides and weights it hands to which launch — can be checked against the oracle without a GPU.  The product never imports
this (it has no CPU path, tests/test_host_cpu.py::test_no_cpu_fallback); the kernels themselves are checked on the GPU against
torch in tests/test_gpu_vae.py."""
import contextlib

import torch
import torch.nn.functional as F


def _v5(rows, g, C=None):
    """rows over grid g -> view [n, T + tf, Hp, Wp, C]"""
    C = rows.shape[1] if C is None else C
    return rows.view(g.n, g.sample_rows, -1)[:, :(g.T + g.tf) * g.plane].view(g.n, g.T + g.tf, g.Hp, g.Wp, -1)[..., :C]


def _interior(rows, g, C=None):
    return _v5(rows, g, C)[:, g.tf:, g.pad:g.pad + g.H, g.pad:g.pad + g.W]


def vae_first_im2col(z, kt, kcols, params):
    scale, shift = torch.tensor(params[0:4]), torch.tensor(params[4:8])
    pw, pb = torch.tensor(params[8:24]).view(4, 4), torch.tensor(params[24:28])
    _, Fr, H, W = z.shape
    x = z.float() * scale[:, None, None, None] + shift[:, None, None, None]
    x = torch.einsum("oc,cfhw->ofhw", pw, x) + pb[:, None, None, None]
    x = x.to(torch.bfloat16).float()
    xp = F.pad(x, (1, 1, 1, 1, kt - 1, 0))                                  # zeros AFTER the 1x1 (the conv's own padding)
    cols = []
    for a in range(kt):
        for dy in range(3):
            for dx in range(3):
                cols.append(xp[:, a:a + Fr, dy:dy + H, dx:dx + W])           # [4, F, H, W] per tap, k = tap * 4 + c
    m = torch.stack(cols, 0).permute(2, 3, 4, 0, 1).reshape(Fr * H * W, kt * 36)
    out = torch.zeros(Fr * H * W, kcols)
    out[:, :kt * 36] = m
    return out.to(torch.bfloat16)


def gemm128(a, w, bias=None, res=None, out=None, out_f32=None, out_scale=1.0, batch=1, batch_a=0, batch_w=0, batch_o=0, M=None):
    assert batch == 1 and out_f32 is None and out_scale == 1.0, "the encode host flow uses the plain form only"
    y = a.float() @ w.float().t()
    if bias is not None:
        y = y + bias.float()
    if res is not None:
        y = y + res.float()
    y = y.to(torch.bfloat16)
    if out is not None:
        out.copy_(y)
        return out
    return y


def conv(a, grid, w, bias, cin, kt, ks, out=None, res=None):
    assert grid.tf == kt - 1 and (ks == 1 or grid.pad == 1) and a.shape[0] == grid.rows
    og = grid.conv_out()
    x = _v5(a, grid, cin).permute(0, 4, 1, 2, 3).float()                      # [n, C, T + tf, Hp, Wp], borders are the padding
    wt = w.float().view(w.shape[0], kt, ks, ks, cin).permute(0, 4, 1, 2, 3)
    y = F.conv3d(x, wt, None if bias is None else bias.float())               # valid: [n, N, T, H, W]
    o = torch.zeros(og.rows, w.shape[0])
    _interior(o, og).copy_(y.permute(0, 2, 3, 4, 1))
    if res is not None:
        o = o + res.float()
    o = o.to(torch.bfloat16)
    if out is not None:
        out.copy_(o)
        return out
    return o


def group_norm(x, gs, y, gd, C, gamma, beta, eps, silu_act, groups=32):
    v = _interior(x, gs, C).permute(0, 4, 1, 2, 3).float()
    h = F.group_norm(v, groups, gamma.float(), beta.float(), eps)
    if silu_act:
        h = F.silu(h)
    _interior(y, gd, C).copy_(h.permute(0, 2, 3, 4, 1).to(torch.bfloat16))
    return y


def regrid(x, gs, y, gd, C, up=0, tmode=0):
    assert up == 0 and tmode == 0
    _interior(y, gd, C).copy_(_interior(x, gs, C))
    return y


def subsample(x, gs, y, gd, C, t_stride=1, s_stride=1, t_first=0, s_first=0):
    src = _interior(x, gs, C)[:, t_first::t_stride, s_first::s_stride, s_first::s_stride]
    _interior(y, gd, C).copy_(src[:, :gd.T, :gd.H, :gd.W])
    return y


def extract_planar(x, g, nc, tskip, out, f0):
    assert 1 <= nc <= 4 and x.stride(0) % 4 == 0 and out.is_contiguous()       # the kernel's contract
    v = _interior(x, g, nc)                                                    # [n, T, H, W, nc]
    fr = v.reshape(g.n * g.T, g.H, g.W, nc)[tskip:]
    out[:, f0:f0 + fr.shape[0]] = fr.permute(3, 0, 1, 2)
    return out


def attention_2d(x, g, A):
    """OpenSoraVAE._attention (the batched-GEMM / softmax composition is the decode path's, checked on the GPU): plain torch."""
    from videosys_amd.ops import VaeGrid

    C, L, n = 512, g.H * g.W, g.n
    t = _interior(x, g, C).reshape(n, L, C).float()
    hn = F.group_norm(t.transpose(1, 2), 32, A.a_norm.g.float(), A.a_norm.b.float(), A.a_norm.eps).transpose(1, 2)
    hn = hn.to(torch.bfloat16).float()
    q = (hn @ A.a_wq.float().t() + A.a_bq.float()).to(torch.bfloat16).float()
    k = (hn @ A.a_wk.float().t() + A.a_bk.float()).to(torch.bfloat16).float()
    v = (hn @ A.a_wv.float().t()).to(torch.bfloat16).float()
    p = torch.softmax(q @ k.transpose(1, 2) / (C ** 0.5), dim=-1).to(torch.bfloat16).float()
    o = (p @ v).to(torch.bfloat16).float()
    y = (o @ A.a_wo.float().t() + A.a_bo.float() + t).to(torch.bfloat16)
    gd = VaeGrid(n, 1, g.H, g.W, 0, 0)
    return y.reshape(n * L, C), gd


@contextlib.contextmanager
def emulated_vae_ops():
    from videosys_amd import ops
    from videosys_amd.vae_open_sora import OpenSoraVAE

    mine = dict(vae_first_im2col=vae_first_im2col, gemm128=gemm128, conv=conv, group_norm=group_norm, regrid=regrid,
                subsample=subsample, extract_planar=extract_planar)
    saved = {k: getattr(ops, k) for k in mine}
    saved_attn = OpenSoraVAE._attention
    for k, v in mine.items():
        setattr(ops, k, v)
    OpenSoraVAE._attention = lambda self, x, g, aw=None: attention_2d(x, g, aw if aw is not None else self)
    try:
        yield
    finally:
        for k, v in saved.items():
            setattr(ops, k, v)
        OpenSoraVAE._attention = saved_attn


def cpu_vae(state_dict, encoder=True):
    """An OpenSoraVAE object on CPU tensors for the emulated host-flow test (bypasses the constructor's device check)."""
    from videosys_amd.vae_open_sora import OpenSoraVAE

    v = OpenSoraVAE.__new__(OpenSoraVAE)
    v.device = torch.device("cpu")
    v.micro_frame_size, v.micro_batch_size, v.frames_per_launch = 17, 4, 16
    v.micro_z_frame_size = 5
    v._padded = {}
    v._init_temporal(state_dict, v.device)
    v._init_spatial(state_dict, v.device, "spatial_vae.module.")
    v.has_encoder = encoder
    if encoder:
        v._init_encoders(state_dict, v.device)
    return v
