"""Tensor-level wrappers over the C ABI (include/videosys_amd.h).

PyTorch is plumbing here: it owns the HBM allocations and the HIP stream; every wrapper passes raw device pointers
and sizes to libvideosys_amd.so.  All wrappers require CUDA(HIP) tensors and raise otherwise — no eager fallback.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib, program

EPI_BIAS, EPI_BIAS_GELU, EPI_GATE_RES = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_GELU_TANH = 0, 1, 2
HEAD_DIM = 72
VT_ROWS = 96


class _TPtr(int):
    """A device address that remembers its tensor: ctypes and the launch-program recorder see the integer, the custom-op route
    (torch.ops.vsys.launch) hands the TENSOR to the dispatcher."""


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if program.active() is not None:
        program.keep(t)          # a recorded launch holds this address: the program keeps the tensor alive
    a = _TPtr(t.data_ptr())
    a.t = t
    return a


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _call(name: str, *args):
    """One C-ABI launch on torch's current stream (``args`` = the entry point's arguments without the trailing stream); under a
    launch-program recorder (program.py) the launch is also logged for replay.  Route: ``torch.ops.vsys.launch`` (the TORCH_LIBRARY
    fragment of csrc/torch_binding.cpp: tensors travel as tensors) for every entry point that has a VSYS_OP code — the whole denoise
    step — and ctypes for the rest (VAE, T5) or when the fragment is not there."""
    import ctypes

    s = torch.cuda.current_stream()
    rec = program.active()
    sig = _lib.SIGNATURES[name][:-1]
    if rec is not None:
        rec.launch(name, args, sig, s)
    tv = _lib.torch_ops()
    op = program.OPCODES.get(name) if tv is not None else None
    if op is None:
        _lib.check(getattr(_lib.load(), name)(*args, s.cuda_stream), name)
        return
    tensors, ints, floats = [], [], []
    for v, t in zip(args, sig):
        if t is _lib._f32:
            floats.append(float(v))
            continue
        if isinstance(v, _TPtr):
            tensors.append(v.t)
            ints.append(0)
        else:
            tensors.append(None)
            ints.append(ctypes.addressof(v) if isinstance(v, ctypes.Array) else (0 if v is None else int(v)))
    try:
        tv.launch(op, tensors, ints, floats, s.cuda_stream)
    except RuntimeError as e:
        raise _lib.VsysError(f"{name}: {str(e).splitlines()[0]}") from None


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.VsysError("videosys_amd ops need HIP device tensors (no CPU fallback)")


def _bf16(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.bfloat16:
            raise _lib.VsysError(f"expected bf16 tensor, got {t.dtype}")


def gemm(x, w, bias=None, *, epilogue=EPI_BIAS, gate=None, gate_stride=0, rows_per_sample=0, res=None, aux=None, out=None):
    """out[M,N] = epilogue(x[M,K] @ w[N,K]^T + bias). x may be a row-strided 2-D view (last dim contiguous)."""
    _chk(x, w, bias, gate, res, aux, out)
    _bf16(x, w, bias, gate, res, aux, out)
    assert x.dim() == 2 and w.dim() == 2 and x.stride(1) == 1 and w.stride(1) == 1
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    assert out.stride(1) == 1
    lib = _lib.load()
    _call("vsys_gemm_bf16", _p(x), x.stride(0), _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), M, N, K, epilogue,
                                  _p(gate), gate_stride, rows_per_sample, _p(res), res.stride(0) if res is not None else 0,
                                  _p(aux), aux.stride(0) if aux is not None else 0)
    return out


LN_BLOCK = 96   # columns per LayerNorm partial of the AdaLN fold (include/videosys_amd.h, vsys_gemm_bf16_ln)


def _stats_ld(stats, rows, nblk):
    """Leading dimension (rows per 96-column block) of a statistics buffer or of a ROW SLICE of one (``buf[:, r0:r1]``: the C side
    addresses partial b of row r at base + (b * ld + r) * 8 bytes, so a slice is just another base pointer with the parent's ld)."""
    assert stats.dtype == torch.float32 and stats.dim() == 3 and stats.shape[0] == nblk and stats.shape[1] >= rows and stats.shape[2] == 2
    assert stats.stride(2) == 1 and stats.stride(1) == 2 and stats.stride(0) % 2 == 0 and stats.stride(0) // 2 >= stats.shape[1]
    return stats.stride(0) // 2


def ln_stats_buffer(rows, C, device):
    """fp32 [C / 96, rows, 2]: (mean, M2) of every 96-column block of every row (the statistics format of the AdaLN fold)."""
    assert C % LN_BLOCK == 0
    return torch.empty(C // LN_BLOCK, rows, 2, dtype=torch.float32, device=device)


def gemm_ln(x, wp, cs, cv, stats, *, gelu=False, eps=1e-6, out=None):
    """out = [gelu](Linear(t2i_modulate(LayerNorm(x), shift, scale))) with the modulation folded into wp / cs / cv
    (adaln_prescale) and the LayerNorm statistics of x's rows taken from ``stats`` (gemm_stats / ln_row_stats)."""
    _chk(x, wp, cs, cv, stats, out)
    _bf16(x, wp, out)
    assert x.dim() == 2 and x.stride(1) == 1 and wp.stride(1) == 1 and cs.dtype == torch.float32 and cv.dtype == torch.float32
    M, K = x.shape
    N = wp.shape[0]
    assert wp.shape[1] == K
    ld = _stats_ld(stats, M, K // LN_BLOCK)
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    _call("vsys_gemm_bf16_ln", _p(x), x.stride(0), _p(wp), wp.stride(0), _p(cs), _p(cv), _p(out), out.stride(0), M, N, K,
          EPI_BIAS_GELU if gelu else EPI_BIAS, _p(stats), ld, float(eps))
    return out


def gemm_stats(x, w, bias, stats, *, gate=None, gate_stride=0, rows_per_sample=0, res=None, out=None):
    """gemm(..., epilogue=EPI_GATE_RES) that also writes the LayerNorm partials of the rows it stores into ``stats``."""
    _chk(x, w, bias, gate, res, out, stats)
    _bf16(x, w, bias, gate, res, out)
    assert x.dim() == 2 and x.stride(1) == 1 and w.stride(1) == 1
    M, K = x.shape
    N = w.shape[0]
    ld = _stats_ld(stats, M, N // LN_BLOCK)
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    _call("vsys_gemm_bf16_stats", _p(x), x.stride(0), _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), M, N, K, _p(gate),
          gate_stride, rows_per_sample, _p(res), res.stride(0) if res is not None else 0, _p(stats), ld)
    return out


def gemm_gate_res_add(x, w, bias, *, res, gate=None, gate_stride=0, rows_per_sample=0, aux=None, adds=(), stats=None, out=None):
    """gemm(..., epilogue=EPI_GATE_RES) whose store phase also performs the ``out += a`` passes (one bf16 rounding each, in order) of
    up to two tensors ``adds`` shaped like ``res`` — the PAB broadcasts that follow the GEMM in program order — and, with ``stats``,
    writes the LayerNorm partials of what it stored (gemm_stats format)."""
    adds = tuple(adds)
    assert len(adds) <= 2 and res is not None
    _chk(x, w, bias, gate, res, aux, out, stats, *adds)
    _bf16(x, w, bias, gate, res, aux, out, *adds)
    assert x.dim() == 2 and x.stride(1) == 1 and w.stride(1) == 1
    M, K = x.shape
    N = w.shape[0]
    for t in adds + ((aux,) if aux is not None else ()):
        assert t.shape == res.shape and t.stride() == res.stride()
    if stats is not None:
        assert stats.dtype == torch.float32 and stats.is_contiguous() and stats.shape[0] == N // LN_BLOCK and stats.shape[1] >= M
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    _call("vsys_gemm_bf16_gate_res_add", _p(x), x.stride(0), _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), M, N, K, _p(gate),
          gate_stride, rows_per_sample, _p(res), res.stride(0), _p(aux), _p(adds[0]) if adds else None, _p(adds[1]) if len(adds) > 1 else None,
          _p(stats), stats.shape[1] if stats is not None else 0)
    return out


def adaln_prescale(sites, nblocks, mod):
    """sites: int64 device tensor [nsites, 10] (include/videosys_amd.h, vsys_adaln_prescale); mod: the step's modulation table."""
    _chk(sites, mod)
    _bf16(mod)
    assert sites.dtype == torch.int64 and sites.is_contiguous() and sites.shape[1] == 10 and mod.is_contiguous()
    _call("vsys_adaln_prescale", _p(sites), sites.shape[0], nblocks, _p(mod))


def ln_row_stats(x, stats):
    _chk(x, stats)
    _bf16(x)
    assert x.is_contiguous() and x.dim() == 2 and stats.dtype == torch.float32 and stats.is_contiguous()
    rows, C = x.shape
    assert stats.shape[0] == C // LN_BLOCK and stats.shape[1] >= rows
    _call("vsys_ln_row_stats", _p(x), rows, C, _p(stats), stats.shape[1])
    return stats


def linear_small(x, w, bias=None, act_in=ACT_NONE, act_out=ACT_NONE, out=None):
    _chk(x, w, bias, out)
    _bf16(x, w, bias, out)
    assert x.dim() == 2 and x.stride(1) == 1 and w.stride(1) == 1
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    lib = _lib.load()
    _call("vsys_linear_small", _p(x), x.stride(0), _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), M, N, K, act_in,
                                     act_out)
    return out


def adaln_modulate(x, shift, scale, rows_per_sample, mod_stride, eps=1e-6, out=None):
    """x [rows, C] contiguous; shift/scale point at sample-0 vectors, sample stride mod_stride elements."""
    _chk(x, shift, scale, out)
    _bf16(x, shift, scale, out)
    assert x.is_contiguous()
    rows, C = x.shape
    if out is None:
        out = torch.empty_like(x)
    lib = _lib.load()
    _call("vsys_adaln_modulate", _p(x), _p(shift), _p(scale), _p(out), rows, C, rows_per_sample, mod_stride, eps)
    return out


def mod_table(table, t_mlp, out=None):
    """table [nblk, 6*C], t_mlp [B, 6*C] -> [nblk, B, 6*C]"""
    _chk(table, t_mlp, out)
    _bf16(table, t_mlp, out)
    nblk, C6 = table.shape
    B = t_mlp.shape[0]
    if out is None:
        out = torch.empty(nblk, B, C6, dtype=torch.bfloat16, device=table.device)
    lib = _lib.load()
    _call("vsys_mod_table", _p(table), _p(t_mlp), _p(out), nblk, B, C6)
    return out


def timestep_embedding(t_f32, dim=256):
    _chk(t_f32)
    assert t_f32.dtype == torch.float32 and t_f32.is_contiguous()
    B = t_f32.numel()
    out = torch.empty(B, dim, dtype=torch.bfloat16, device=t_f32.device)
    lib = _lib.load()
    _call("vsys_timestep_embedding", _p(t_f32), _p(out), B, dim)
    return out


def patch_embed(z_f32, w, bias, pos, B, patch, C):
    """z fp32 [Bz, Cin, T, H, W] -> bf16 [B, T, S, C] (sample b reads z[b % Bz])."""
    _chk(z_f32, w, bias, pos)
    _bf16(w, bias, pos)
    assert z_f32.dtype == torch.float32 and z_f32.is_contiguous() and w.is_contiguous() and pos.is_contiguous()
    Bz, Cin, T, H, W = z_f32.shape
    assert patch[0] == 1
    ph, pw = patch[1], patch[2]
    Hp, Wp = -(-H // ph), -(-W // pw)
    out = torch.empty(B, T, Hp * Wp, C, dtype=torch.bfloat16, device=z_f32.device)
    lib = _lib.load()
    _call("vsys_patch_embed", _p(z_f32), Bz, _p(w), _p(bias), _p(pos), _p(out), B, Cin, T, H, W, ph, pw, C)
    return out


def patch_embed_shard(z_f32, w, bias, pos, B, patch, C, s0, Sl):
    """The tokens s0 .. s0+Sl-1 of every (b, t) only: bf16 [B, T, Sl, C]; tokens past Hp*Wp are zero rows (sequence padding)."""
    _chk(z_f32, w, bias, pos)
    _bf16(w, bias, pos)
    assert z_f32.dtype == torch.float32 and z_f32.is_contiguous() and w.is_contiguous() and pos.is_contiguous()
    Bz, Cin, T, H, W = z_f32.shape
    assert patch[0] == 1
    out = torch.empty(B, T, Sl, C, dtype=torch.bfloat16, device=z_f32.device)
    _call("vsys_patch_embed_shard", _p(z_f32), Bz, _p(w), _p(bias), _p(pos), _p(out), B, Cin, T, H, W, patch[1], patch[2], C, s0, Sl)
    return out


def final_layer_tokens(x, table, tvec, w, bias, B, T, Sl, eps=1e-6):
    """T2IFinalLayer on the local rows [B*T*Sl, C] -> fp32 [B, T, Sl, n_out] (n_out = rows of the linear = ph*pw*Cout)."""
    _chk(x, table, tvec, w, bias)
    _bf16(x, table, tvec, w, bias)
    assert x.is_contiguous() and w.is_contiguous() and table.is_contiguous() and tvec.is_contiguous()
    C, n_out = x.shape[-1], w.shape[0]
    out = torch.empty(B, T, Sl, n_out, dtype=torch.float32, device=x.device)
    _call("vsys_final_layer_tokens", _p(x), _p(table), _p(tvec), _p(w), _p(bias), _p(out), B, T, Sl, n_out, C, eps)
    return out


def unpatchify_tokens(tokens, P, B, T, Sl, Hp, Wp, H, W, patch, Cout):
    """tokens fp32 [P, B, T, Sl, ph*pw*Cout] (gathered S-shards) -> fp32 [B, Cout, T, H, W]."""
    _chk(tokens)
    assert tokens.dtype == torch.float32 and tokens.is_contiguous() and tokens.numel() == P * B * T * Sl * patch[1] * patch[2] * Cout
    out = torch.empty(B, Cout, T, H, W, dtype=torch.float32, device=tokens.device)
    _call("vsys_unpatchify_tokens", _p(tokens), _p(out), P, B, T, Sl, Hp, Wp, H, W, patch[1], patch[2], Cout)
    return out


def final_layer(x, table, tvec, w, bias, B, T, Hp, Wp, H, W, patch, Cout, eps=1e-6):
    _chk(x, table, tvec, w, bias)
    _bf16(x, table, tvec, w, bias)
    assert x.is_contiguous() and w.is_contiguous() and table.is_contiguous() and tvec.is_contiguous()
    C = x.shape[-1]
    out = torch.empty(B, Cout, T, H, W, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    _call("vsys_final_layer", _p(x), _p(table), _p(tvec), _p(w), _p(bias), _p(out), B, T, Hp, Wp, H, W, patch[1],
                                    patch[2], Cout, C, eps)
    return out


def cfg_euler_step(z_f32, model_out_f32, guidance, dt):
    _chk(z_f32, model_out_f32)
    assert z_f32.dtype == torch.float32 and model_out_f32.dtype == torch.float32
    assert z_f32.is_contiguous() and model_out_f32.is_contiguous()
    Bz, Cin = z_f32.shape[:2]
    Cout = model_out_f32.shape[1]
    assert model_out_f32.shape[0] == 2 * Bz
    thw = z_f32[0, 0].numel()
    lib = _lib.load()
    _call("vsys_cfg_euler_step", _p(z_f32), _p(model_out_f32), Bz, Cin, Cout, thw, float(guidance), float(dt))
    return z_f32


def cfg_linear_step(z_f32, model_out_f32, guidance, c_z, c_eps, cond_first=False):
    """z = c_z z + c_eps (uncond + g (cond - uncond)); model_out [2*Bz, Cout >= Cin, ...] fp32."""
    _chk(z_f32, model_out_f32)
    assert z_f32.dtype == torch.float32 and model_out_f32.dtype == torch.float32
    assert z_f32.is_contiguous() and model_out_f32.is_contiguous()
    Bz, Cin = z_f32.shape[:2]
    Cout = model_out_f32.shape[1]
    assert model_out_f32.shape[0] == 2 * Bz
    thw = z_f32[0, 0].numel()
    lib = _lib.load()
    _call("vsys_cfg_linear_step", _p(z_f32), _p(model_out_f32), Bz, Cin, Cout, thw, float(guidance), float(c_z),
                                        float(c_eps), 1 if cond_first else 0)
    return z_f32


def add_bcast_rows(x, e, group, period):
    """x [rows, C] += e[(row // group) % period]"""
    _chk(x, e)
    _bf16(x, e)
    assert x.is_contiguous() and e.is_contiguous() and x.shape[-1] == e.shape[-1] and e.shape[0] >= period
    rows = x.numel() // x.shape[-1]
    lib = _lib.load()
    _call("vsys_add_bcast_rows", _p(x), _p(e), rows, x.shape[-1], group, period)
    return x


def add_rows(x, y):
    _chk(x, y)
    _bf16(x, y)
    assert x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
    lib = _lib.load()
    _call("vsys_add_rows", _p(x), _p(y), x.numel())
    return x


def copy_4d(src, dst, n0, n1, n2, C, sstr, dstr, n1_valid=None, n2_valid=None):
    _chk(src, dst)
    _bf16(src, dst)
    lib = _lib.load()
    _call("vsys_copy_4d", _p(src), _p(dst), n0, n1, n2, C, sstr[0], sstr[1], sstr[2], dstr[0], dstr[1], dstr[2],
                                n1 if n1_valid is None else n1_valid, n2 if n2_valid is None else n2_valid)
    return dst


def copy_4d_batch(src, dst, descs):
    """descs: list of 14-tuples (src_off, dst_off, n0, n1, n2, run, ss0, ss1, ss2, ds0, ds1, ds2, n1_valid, n2_valid)."""
    import ctypes

    _chk(src, dst)
    _bf16(src, dst)
    lib = _lib.load()
    for i in range(0, len(descs), 16):
        part = descs[i:i + 16]
        flat = [int(v) for d in part for v in d]
        arr = (ctypes.c_int64 * len(flat))(*flat)
        _call("vsys_copy_4d_batch", _p(src), _p(dst), len(part), arr)


def kv_pad_len(kv_len: int) -> int:
    return (kv_len + 63) // 64 * 64


def alloc_kv_buffers(batch, heads, kv_len, device):
    kv_pad = kv_pad_len(kv_len)
    kp = torch.zeros(batch, heads, kv_pad, HEAD_DIM, dtype=torch.bfloat16, device=device)
    vt = torch.zeros(batch, heads, VT_ROWS, kv_pad, dtype=torch.bfloat16, device=device)  # rows 72..95 stay zero
    return kp, vt


def attn_prep_kv(k, v, k_norm_w, kp, vt, batch, heads, kv_len, eps=1e-6):
    """k, v: 2-D row-strided views [batch*kv_len, >= heads*72] (head h at column h*72)."""
    _chk(k, v, k_norm_w, kp, vt)
    _bf16(k, v, k_norm_w, kp, vt)
    assert k.stride(1) == 1 and v.stride(1) == 1 and kp.is_contiguous() and vt.is_contiguous()
    kv_pad = kp.shape[2]
    assert vt.shape[3] == kv_pad and vt.shape[2] == VT_ROWS
    lib = _lib.load()
    _call("vsys_attn_prep_kv", _p(k), k.stride(0), _p(v), v.stride(0), _p(k_norm_w), _p(kp), _p(vt), batch, heads,
                                     kv_len, kv_pad, eps)


def flash_attn(q, q_norm_w, kp, vt, out, batch, heads, q_len, kv_len, eps=1e-6, k_norm_bound=None, keys_exact=False):
    """q/out: 2-D row-strided views [batch*q_len, >= heads*72].  ``k_norm_bound``: the caller's promise about the norms of the Kp
    rows (include/videosys_amd.h, vsys_flash_attn_d72_kb; see rms_key_bound); None = no promise.  ``keys_exact``: the caller's promise
    that (kp, vt) were prepared by attn_prep_kv for exactly ``kv_len`` on zeroed buffers (vsys_flash_attn_d72_exact: no mask on the
    ragged last tile) — never with a kv_len shorter than the buffers were prepared for."""
    _chk(q, q_norm_w, kp, vt, out)
    _bf16(q, q_norm_w, kp, vt, out)
    assert q.stride(1) == 1 and out.stride(1) == 1
    kv_pad = kp.shape[2]
    if keys_exact and not k_norm_bound:
        _call("vsys_flash_attn_d72_exact", _p(q), q.stride(0), _p(q_norm_w), _p(kp), _p(vt), _p(out), out.stride(0), batch, heads,
              q_len, kv_len, kv_pad, eps)
    elif k_norm_bound:
        _call("vsys_flash_attn_d72_kb", _p(q), q.stride(0), _p(q_norm_w), _p(kp), _p(vt), _p(out), out.stride(0), batch, heads,
              q_len, kv_len, kv_pad, eps, float(k_norm_bound))
    else:
        _call("vsys_flash_attn_d72", _p(q), q.stride(0), _p(q_norm_w), _p(kp), _p(vt), _p(out), out.stride(0), batch, heads,
              q_len, kv_len, kv_pad, eps)
    return out


def rms_key_bound(q_norm_w, k_norm_w, head_dim=HEAD_DIM):
    """The k_norm_bound of flash_attn for RMS-normed q and k (LlamaRMSNorm, normalization.py:28-33: x / rms(x) has norm sqrt(d), then
    the weight elementwise), or None when the promise |q_i| k_norm_bound <= 60 cannot be given from the weights alone.  Host-side,
    once per block at load time (two 72-element reads)."""
    import math

    wq, wk = float(q_norm_w.float().abs().max()), float(k_norm_w.float().abs().max())
    kb = math.sqrt(head_dim) * wk * (math.log2(math.e) / math.sqrt(head_dim)) * (1.0 + 2.0 ** -6)   # Kp rows: normed, scaled, rounded
    qb = math.sqrt(head_dim) * wq * (1.0 + 2.0 ** -6)
    return kb if qb * kb * (1.0 + 2.0 ** -5) <= 60.0 else None


def attn_temporal(qkv, C, q_norm_w, k_norm_w, rope_cos, rope_sin, out, B, T, S, heads, eps=1e-6):
    """qkv [B*T*S, 3C] rows ordered (b,t,s); out [B*T*S, C]."""
    _chk(qkv, q_norm_w, k_norm_w, rope_cos, rope_sin, out)
    _bf16(qkv, q_norm_w, k_norm_w, out)
    assert qkv.stride(1) == 1 and out.stride(1) == 1
    if rope_cos is not None:
        assert rope_cos.dtype == torch.float32 and rope_cos.is_contiguous() and rope_cos.shape == (T, HEAD_DIM)
        assert rope_sin.dtype == torch.float32 and rope_sin.is_contiguous() and rope_sin.shape == (T, HEAD_DIM)
    lib = _lib.load()
    _call("vsys_attn_temporal_d72", _p(qkv), qkv.stride(0), C, _p(q_norm_w), _p(k_norm_w), _p(rope_cos), _p(rope_sin),
                                          _p(out), out.stride(0), B, T, S, heads, eps)
    return out


# ------------------------------------------------------------------------------------------------ CogVideoX (head_dim 64)
def gemm_gate2(x, w, bias, gate, gate_stride, rows_per_sample, seg_split, gate_alt, res, out, aux=None):
    """EPI_GATE_RES with two gate vectors per sample (text rows: gate + gate_alt)."""
    _chk(x, w, bias, gate, res, aux, out)
    _bf16(x, w, bias, gate, res, aux, out)
    assert x.dim() == 2 and x.stride(1) == 1 and w.stride(1) == 1 and out.stride(1) == 1
    M, K = x.shape
    N = w.shape[0]
    lib = _lib.load()
    _call("vsys_gemm_bf16_gate2", _p(x), x.stride(0), _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), M, N, K, _p(gate),
                                        gate_stride, rows_per_sample, seg_split, gate_alt, _p(res),
                                        res.stride(0) if res is not None else 0, _p(aux),
                                        aux.stride(0) if aux is not None else 0)
    return out


def ln_modulate(x, ln_w, ln_b, shift, scale, rows_per_sample, mod_stride=0, seg_split=0, mod_alt=0, eps=1e-5, out=None):
    _chk(x, ln_w, ln_b, shift, scale, out)
    _bf16(x, ln_w, ln_b, shift, scale, out)
    assert x.is_contiguous()
    rows, C = x.shape
    if out is None:
        out = torch.empty_like(x)
    lib = _lib.load()
    _call("vsys_ln_modulate", _p(x), _p(ln_w), _p(ln_b), _p(shift), _p(scale), _p(out), rows, C, rows_per_sample, mod_stride,
                                    seg_split, mod_alt, eps)
    return out


def gate_add_rows(x, y, gate, rows_per_sample, gate_stride, seg_split=0, gate_alt=0):
    _chk(x, y, gate)
    _bf16(x, y, gate)
    assert x.is_contiguous() and y.is_contiguous() and x.shape == y.shape
    rows, C = x.shape
    lib = _lib.load()
    _call("vsys_gate_add_rows", _p(x), _p(y), _p(gate), rows, C, rows_per_sample, gate_stride, seg_split, gate_alt)
    return x


def im2col_patch(z_f32, B, p):
    """z fp32 [Bz, F, Cin, H, W] -> bf16 [B*F*(H/p)*(W/p), Cin*p*p]"""
    _chk(z_f32)
    assert z_f32.dtype == torch.float32 and z_f32.is_contiguous()
    Bz, F, Cin, H, W = z_f32.shape
    out = torch.empty(B * F * (H // p) * (W // p), Cin * p * p, dtype=torch.bfloat16, device=z_f32.device)
    lib = _lib.load()
    _call("vsys_im2col_patch", _p(z_f32), Bz, _p(out), B, F, Cin, H, W, p)
    return out


def unpatchify_cvx(x, B, F, Hp, Wp, Cout, p):
    _chk(x)
    _bf16(x)
    assert x.stride(1) == 1
    out = torch.empty(B, F, Cout, Hp * p, Wp * p, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    _call("vsys_unpatchify_cvx", _p(x), x.stride(0), _p(out), B, F, Hp, Wp, Cout, p)
    return out


def alloc_kv_buffers64(batch, heads, kv_len, device):
    kv_pad = kv_pad_len(kv_len)
    kp = torch.zeros(batch, heads, kv_pad, 64, dtype=torch.bfloat16, device=device)
    vt = torch.zeros(batch, heads, 64, kv_pad, dtype=torch.bfloat16, device=device)
    return kp, vt


def attn_prep_kv64(k, v, ln_w, ln_b, rope_cos, rope_sin, rope_start, kp, vt, batch, heads, kv_len, eps=1e-6):
    _chk(k, v, ln_w, ln_b, rope_cos, rope_sin, kp, vt)
    _bf16(k, v, ln_w, ln_b, kp, vt)
    assert k.stride(1) == 1 and v.stride(1) == 1 and kp.is_contiguous() and vt.is_contiguous()
    rope_len = 0
    if rope_cos is not None:
        assert rope_cos.dtype == torch.float32 and rope_cos.is_contiguous() and rope_sin.is_contiguous() and rope_cos.shape[1] == 64
        rope_len = rope_cos.shape[0]
    lib = _lib.load()
    _call("vsys_attn_prep_kv64", _p(k), k.stride(0), _p(v), v.stride(0), _p(ln_w), _p(ln_b), _p(rope_cos), _p(rope_sin),
                                       rope_start, rope_len, _p(kp), _p(vt), batch, heads, kv_len, kp.shape[2], eps)


def flash_attn64(q, ln_w, ln_b, rope_cos, rope_sin, rope_start, kp, vt, out, batch, heads, q_len, kv_len, eps=1e-6, k_norm_bound=None):
    """``k_norm_bound``: the promise about the Kp row norms (vsys_flash_attn_d64_kb; ln_key_bound); None = none."""
    _chk(q, ln_w, ln_b, rope_cos, rope_sin, kp, vt, out)
    _bf16(q, ln_w, ln_b, kp, vt, out)
    assert q.stride(1) == 1 and out.stride(1) == 1
    rope_len = 0 if rope_cos is None else rope_cos.shape[0]
    if k_norm_bound:
        _call("vsys_flash_attn_d64_kb", _p(q), q.stride(0), _p(ln_w), _p(ln_b), _p(rope_cos), _p(rope_sin), rope_start, rope_len,
              _p(kp), _p(vt), _p(out), out.stride(0), batch, heads, q_len, kv_len, kp.shape[2], eps, float(k_norm_bound))
    else:
        _call("vsys_flash_attn_d64", _p(q), q.stride(0), _p(ln_w), _p(ln_b), _p(rope_cos), _p(rope_sin), rope_start, rope_len,
              _p(kp), _p(vt), _p(out), out.stride(0), batch, heads, q_len, kv_len, kp.shape[2], eps)
    return out


def ln_key_bound(q_w, q_b, k_w, k_b, head_dim=64):
    """The k_norm_bound of flash_attn64 for LayerNorm-ed q and k with affine weights (diffusers Attention qk_norm="layer_norm"; the
    rotary embedding rotates pairs: norms unchanged): |LN(x) w + b| <= sqrt(d) max|w| + |b|_2.  None when |q| |k| <= 60 cannot be
    promised from the weights, or when a norm is missing.  Host-side, once per block."""
    import math

    if q_w is None or k_w is None:
        return None

    def side(w, b):
        return math.sqrt(head_dim) * float(w.float().abs().max()) + (float(b.float().norm()) if b is not None else 0.0)

    kb = side(k_w, k_b) * (math.log2(math.e) / math.sqrt(head_dim)) * (1.0 + 2.0 ** -6)
    qb = side(q_w, q_b) * (1.0 + 2.0 ** -6)
    return kb if qb * kb * (1.0 + 2.0 ** -5) <= 60.0 else None


# ------------------------------------------------------------------------------------------------ VAE decode (row a14)
import ctypes as _ct


class VaeGrid:
    """Activation grid of the VAE kernels (include/videosys_amd.h): ``n`` samples of (T, H, W) with a spatial zero border of
    ``pad`` pixels and ``tf`` zero frames in front; rows are channels-last.  ``guard`` rows of slack precede and follow the
    grid in the allocation so tap-shifted conv reads stay inside it."""

    def __init__(self, n, T, H, W, pad=0, tf=0, sample_rows=None):
        self.n, self.T, self.H, self.W, self.pad, self.tf = n, T, H, W, pad, tf
        self.Hp, self.Wp = H + 2 * pad, W + 2 * pad
        self.plane = self.Hp * self.Wp
        self.sample_rows = (T + tf) * self.plane if sample_rows is None else sample_rows
        assert self.sample_rows >= (T + tf) * self.plane
        self.rows = n * self.sample_rows
        self.guard = (self.Wp + 1) if pad else 0
        self._c = (_ct.c_int64 * 6)(T, H, W, pad, tf, self.sample_rows)

    def alloc(self, C, device, zero=False):
        """[guard + rows + guard, C] bf16; returns (storage, view of the grid rows)."""
        total = self.rows + 2 * self.guard
        buf = (torch.zeros if zero else torch.empty)(total, C, dtype=torch.bfloat16, device=device)
        return buf, buf[self.guard:self.guard + self.rows]

    def conv_out(self):
        """Grid of a conv's output when this grid is its (padded) input: same rows minus the front frames."""
        return VaeGrid(self.n, self.T, self.H, self.W, self.pad, 0)


def conv(a, grid: VaeGrid, w, bias, cin, kt, ks, out=None, res=None):
    """Tap-shifted implicit-GEMM conv over the padded grid ``grid`` whose rows are the 2-D tensor ``a``.  ks = spatial kernel
    (1 or 3), kt = temporal taps (needs grid.tf == kt - 1).  Returns rows of grid.conv_out() [M, N].

    ``a`` MUST be the row view returned by ``grid.alloc`` (or a buffer laid out the same way): a 3 x 3 kernel reads up to
    grid.guard = W + 3 rows before the first and after the last row of the view, and the border pixels / front frames of the
    grid must be zero — that is where the convolution's zero padding comes from."""
    _chk(a, w, bias, res, out)
    _bf16(a, w, bias, res, out)
    assert grid.tf == kt - 1 and (ks == 1 or grid.pad == 1) and a.shape[0] == grid.rows and a.stride(1) == 1
    og = grid.conv_out()
    M, N = og.rows, w.shape[0]
    assert w.shape[1] == cin * kt * ks * ks and a.shape[1] >= cin
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    shift_rows = (grid.Wp + 1) if ks == 3 else 0
    a_ptr = a.data_ptr() - shift_rows * a.stride(0) * 2
    lib = _lib.load()
    _call("vsys_conv_bf16", a_ptr, a.stride(0), _p(w), w.stride(0), _p(bias), _p(res), res.stride(0) if res is not None else 0,
                                  _p(out), None, out.stride(0), M, N, cin, kt, ks, ks, grid.Wp, grid.plane, 1, 0, 0, 0, 1.0)
    return out


def gemm128(a, w, bias=None, res=None, out=None, out_f32=None, out_scale=1.0, batch=1, batch_a=0, batch_w=0, batch_o=0, M=None, K=None):
    """Plain C = A W^T (+ bias, + res) on the 128-column tile kernel; batch > 1 strides the operands (elements).  ``K``: contract
    over the first K columns only (with batch strides along K: a split-K run, one K slice per batch entry)."""
    _chk(a, w, bias, res, out, out_f32)
    _bf16(a, w, bias, res, out)
    M = a.shape[0] if M is None else M
    N = w.shape[-2]                   # (a batched w is [batch, N, K]; also when the batch happens to be 1)
    K = a.shape[-1] if K is None else K
    if out is None and out_f32 is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    o = out if out is not None else out_f32
    lib = _lib.load()
    _call("vsys_conv_bf16", _p(a), a.stride(-2), _p(w), w.stride(-2), _p(bias), _p(res), res.stride(-2) if res is not None else 0,
                                  _p(out), _p(out_f32), o.stride(-2), M, N, K, 1, 1, 1, 0, 0, batch, batch_a, batch_w, batch_o,
                                  float(out_scale))
    return o


_GN_NBLK = 128


def group_norm(x, gs: VaeGrid, y, gd: VaeGrid, C, gamma, beta, eps, silu_act, groups=32):
    """y[interior of gd] = act(GroupNorm(x[interior of gs])); statistics per sample over (T, H, W) and the group's channels."""
    _chk(x, y, gamma, beta)
    _bf16(x, y, gamma, beta)
    assert x.shape[0] == gs.rows and y.shape[0] == gd.rows and x.shape[1] == C and y.shape[1] == C
    assert x.is_contiguous() and y.is_contiguous()
    lib = _lib.load()
    partial = torch.empty(gs.n * _GN_NBLK * (C // 4) * 2, dtype=torch.float32, device=x.device)
    stats = torch.empty(gs.n * groups * 2, dtype=torch.float32, device=x.device)
    _call("vsys_gn_stats", _p(x), gs._c, gs.n, C, groups, float(eps), _p(partial), _GN_NBLK, _p(stats))
    _call("vsys_gn_apply", _p(x), gs._c, _p(y), gd._c, gs.n, C, groups, _p(stats), _p(gamma), _p(beta),
                                 ACT_SILU if silu_act else ACT_NONE)
    return y


def regrid(x, gs: VaeGrid, y, gd: VaeGrid, C, up=0, tmode=0):
    _chk(x, y)
    _bf16(x, y)
    assert x.shape[0] == gs.rows and y.shape[0] == gd.rows and x.is_contiguous() and y.is_contiguous()
    _call("vsys_regrid", _p(x), gs._c, _p(y), gd._c, gs.n, C, up, tmode)
    return y


def subsample(x, gs: VaeGrid, y, gd: VaeGrid, C, t_stride=1, s_stride=1, t_first=0, s_first=0):
    """y[(t, h, w) of gd] = x[(t * t_stride + t_first, h * s_stride + s_first, w * s_stride + s_first) of gs] (strided convs)."""
    _chk(x, y)
    _bf16(x, y)
    assert x.shape[0] == gs.rows and y.shape[0] == gd.rows and x.is_contiguous() and y.is_contiguous()
    _call("vsys_subsample", _p(x), gs._c, _p(y), gd._c, gs.n, C, t_stride, s_stride, t_first, s_first)
    return y


def spatial_norm_silu(x, gs: VaeGrid, y, gd: VaeGrid, C, gamma, beta, yb, zdims, eps=1e-6, groups=32):
    """y[interior of gd] = silu(GroupNorm(x) * Y + B) with [Y | B] = yb rows over the latent grid zdims = (zT, zH, zW)."""
    _chk(x, y, gamma, beta, yb)
    _bf16(x, y, gamma, beta, yb)
    zT, zH, zW = zdims
    assert x.shape == (gs.rows, C) and y.shape == (gd.rows, C) and x.is_contiguous() and y.is_contiguous()
    assert yb.shape == (gs.n * zT * zH * zW, 2 * C) and yb.is_contiguous()
    lib = _lib.load()
    partial = torch.empty(gs.n * _GN_NBLK * (C // 4) * 2, dtype=torch.float32, device=x.device)
    stats = torch.empty(gs.n * groups * 2, dtype=torch.float32, device=x.device)
    _call("vsys_gn_stats", _p(x), gs._c, gs.n, C, groups, float(eps), _p(partial), _GN_NBLK, _p(stats))
    _call("vsys_spatial_norm_apply", _p(x), gs._c, _p(y), gd._c, gs.n, C, groups, _p(stats), _p(gamma), _p(beta), _p(yb), zT, zH, zW)
    return y


def blend_edge(a, b, ext, axis):
    """a, b planar bf16 [..., H, W] (contiguous); in place on b (blend_v: axis 0, blend_h: axis 1)."""
    _chk(a, b)
    _bf16(a, b)
    assert a.is_contiguous() and b.is_contiguous() and a.shape[:-2] == b.shape[:-2]
    outer = a.numel() // (a.shape[-2] * a.shape[-1])
    _call("vsys_blend_edge", _p(a), _p(b), outer, a.shape[-2], a.shape[-1], b.shape[-2], b.shape[-1], ext, axis)
    return b


def d2s_time(x, gs: VaeGrid, y, gd: VaeGrid, Cout):
    _chk(x, y)
    _bf16(x, y)
    assert x.shape == (gs.rows, 2 * Cout) and y.shape == (gd.rows, Cout) and x.is_contiguous() and y.is_contiguous()
    _call("vsys_d2s_time", _p(x), gs._c, _p(y), gd._c, gs.n, Cout)
    return y


def vae_first_im2col(z, kt, kcols, params):
    """z bf16 planar [4, F, H, W] -> [F*H*W, kcols]; params = 28 host floats (scale, shift, pq_w row-major, pq_b)."""
    _chk(z)
    _bf16(z)
    assert z.dim() == 4 and z.shape[0] == 4 and z.is_contiguous() and len(params) == 28
    _, F, H, W = z.shape
    out = torch.empty(F * H * W, kcols, dtype=torch.bfloat16, device=z.device)
    arr = (_ct.c_float * 28)(*[float(v) for v in params])
    _call("vsys_vae_first_im2col", _p(z), F, H, W, kt, kcols, arr, _p(out))
    return out


def extract_planar(x, g: VaeGrid, nc, tskip, out, f0):
    """first nc channels of the rows of grid g -> out[c, f0 + frame - tskip, h, w] (planar bf16 [nc, Ftot, H, W])."""
    _chk(x, out)
    _bf16(x, out)
    assert x.shape[0] == g.rows and out.is_contiguous() and out.shape[0] == nc and tuple(out.shape[2:]) == (g.H, g.W)
    _call("vsys_extract_planar", _p(x), g._c, g.n, x.stride(0), nc, tskip, _p(out), out.shape[1], f0)
    return out


def softmax_rows(s_f32, n=None, out=None):
    """softmax over the first n columns of every row of s_f32 [..., ld]; columns n.. of the bf16 result are zero."""
    _chk(s_f32, out)
    assert s_f32.dtype == torch.float32 and s_f32.is_contiguous()
    ld = s_f32.shape[-1]
    n = ld if n is None else n
    rows = s_f32.numel() // ld
    if out is None:
        out = torch.empty(s_f32.shape, dtype=torch.bfloat16, device=s_f32.device)
    _call("vsys_softmax_rows", _p(s_f32), _p(out), rows, n, ld)
    return out


# ------------------------------------------------------------------------------------------------ T5 encoder
def t5_attention_mfma(qkv, bias_pad, center, lens, B, L, heads, out=None, ws=None):
    """T5 self-attention on the matrix pipe (the head_dim-64 flash kernel with a (head, key - query) bias), one launch pair per
    sample.  qkv bf16 [B*L, 3*inner]; bias_pad fp32 [heads, ld]: log2(e) * relative-position bias of (h, key - query) at column
    center + key - query (t5.padded_bias_table); lens: the B key lengths (host ints); ws: (kp, vt) workspaces."""
    _chk(qkv, bias_pad, out)
    _bf16(qkv, out)
    inner = heads * 64
    assert qkv.shape == (B * L, 3 * inner) and qkv.stride(1) == 1 and bias_pad.dtype == torch.float32 and bias_pad.is_contiguous()
    assert bias_pad.shape[0] == heads and len(lens) == B
    kv_pad = (L + 63) // 64 * 64
    if out is None:
        out = torch.empty(B * L, inner, dtype=torch.bfloat16, device=qkv.device)
    if ws is None:
        ws = (torch.empty(heads * kv_pad * 64, dtype=torch.bfloat16, device=qkv.device),
              torch.empty(heads * kv_pad * 64, dtype=torch.bfloat16, device=qkv.device))
    lib = _lib.load()
    for b in range(B):
        q_b, o_b = qkv[b * L:(b + 1) * L], out[b * L:(b + 1) * L]
        _call("vsys_t5_attention_mfma", _p(q_b), qkv.stride(0), inner, _p(bias_pad), bias_pad.shape[1], center, int(lens[b]),
                                              _p(ws[0]), _p(ws[1]), _p(o_b), out.stride(0), L, heads)
    return out


def skinny_split(N, Mp, K, wide=None):
    """K slices for linear_skinny.  Wide path (Mp a multiple of 384: the 256 x 384 tile, ONE workgroup per CU): the largest slice
    count that keeps (weight panels) x (activation tiles) x slices within one round of 256 workgroups, slices of >= 512 columns
    (a multiple of 32).  128-column path (tools/skinny_probe.py, profiles/r03_t5_skinny_probe.jsonl): one slice with >= 144
    workgroups, else the power of two that brings the count closest to 192, slices of >= 1024 columns."""
    wide = (Mp % 384 == 0) if wide is None else wide
    if wide:
        tiles = ((N + 255) // 256) * (Mp // 384)
        best = 1
        for s in range(2, 33):   # (slices need not divide K: k-tiles of 32 are dealt out evenly, the last slice takes what is left)
            ks = -(-(K // 32) // s) * 32
            if tiles * s <= 256 and ks >= 512 and ks * (s - 1) < K:
                best = s
        return best
    tiles = ((N + 255) // 256) * (Mp // 128)
    s = 1
    while tiles * s < 128 and s < 8 and K % (64 * s) == 0 and K // (2 * s) >= 1024:
        s *= 2
    return s


def linear_skinny(x, M, w, res=None, out=None, nsplit=None, part=None, wide=None):
    """out[:M] = x[:M] @ w^T (+ res[:M]) for FEW rows against a large weight (T5 at 300 tokens: 300 MACs per weight element, a
    weight stream).  x bf16 [Mp, K] (rows >= M are never looked at in the result), w bf16 [N, K].  The GEMM runs with the WEIGHT
    as the row operand and K cut in ``nsplit`` slices, leaving fp32 partials that a reduce kernel sums, rounds and adds the
    residual to (in place on the residual stream if asked).
      * Mp % 384 == 0 (``wide``): the 256 x 384 tile of gemm2_bf16.hip — every 256-row weight panel is read by ONE workgroup
        together with all activation rows (vsys_gemm_skinny_slices, partials [s][m][n], vsys_splitk_reduce);
      * Mp % 128 == 0: the 128-column kernel of conv_bf16.hip — a panel is shared by its Mp / 128 column-tile workgroups
        (partials [s][n][m], vsys_splitk_reduce_t transposes back).
    ``part``: fp32 workspace of >= nsplit * N * Mp elements (allocated when None)."""
    _chk(x, w, res, out, part)
    _bf16(x, w, res, out)
    Mp, K = x.shape
    N = w.shape[0]
    wide = (Mp % 384 == 0) if wide is None else wide
    assert Mp % (384 if wide else 128) == 0 and M <= Mp and w.shape[1] == K and x.is_contiguous() and w.is_contiguous()
    nsplit = skinny_split(N, Mp, K, wide) if nsplit is None else nsplit
    assert K % 32 == 0 and (wide or (K % nsplit == 0 and (K // nsplit) % 32 == 0))
    Ks = K // nsplit
    if part is None:
        part = torch.empty(nsplit * N * Mp, dtype=torch.float32, device=x.device)
    assert part.dtype == torch.float32 and part.numel() >= nsplit * N * Mp and part.is_contiguous()
    if out is None:
        out = torch.empty(Mp, N, dtype=torch.bfloat16, device=x.device)
    assert out.shape[1] == N and out.stride(1) == 1 and (res is None or (res.shape[1] == N and res.stride(1) == 1))
    lib = _lib.load()
    ldr = res.stride(0) if res is not None else 0
    if wide:
        _call("vsys_gemm_skinny_slices", _p(w), w.stride(0), _p(x), x.stride(0), _p(part), M, Mp, N, K, nsplit)
        _call("vsys_splitk_reduce", _p(part), nsplit, Mp * N, N, _p(res), ldr, _p(out), out.stride(0), M, N)
        return out
    pv = part[:nsplit * N * Mp].view(nsplit, N, Mp)
    gemm128(w, x, out_f32=pv, batch=nsplit, batch_a=Ks, batch_w=Ks, batch_o=N * Mp, M=N, K=Ks)
    _call("vsys_splitk_reduce_t", _p(pv), nsplit, N * Mp, Mp, _p(res), ldr, _p(out), out.stride(0), M, N)
    return out


def gather_rows(table, ids):
    _chk(table, ids)
    _bf16(table)
    assert ids.dtype == torch.int64 and ids.is_contiguous() and table.is_contiguous()
    n, C = ids.numel(), table.shape[1]
    out = torch.empty(n, C, dtype=torch.bfloat16, device=table.device)
    _call("vsys_gather_rows", _p(table), _p(ids), _p(out), n, C, table.shape[0])
    return out


def rms_norm_rows(x, w, eps=1e-6, out=None):
    _chk(x, w, out)
    _bf16(x, w, out)
    assert x.is_contiguous() and x.dim() == 2
    if out is None:
        out = torch.empty_like(x)
    _call("vsys_rms_norm_rows", _p(x), _p(w), _p(out), x.shape[0], x.shape[1], float(eps))
    return out


def geglu(h, out=None):
    _chk(h, out)
    _bf16(h, out)
    assert h.is_contiguous() and h.shape[1] % 2 == 0
    F = h.shape[1] // 2
    if out is None:
        out = torch.empty(h.shape[0], F, dtype=torch.bfloat16, device=h.device)
    _call("vsys_geglu", _p(h), _p(out), h.shape[0], F)
    return out


def t5_attention(qkv, relbias, klen, B, L, heads, out=None):
    _chk(qkv, relbias, klen, out)
    _bf16(qkv, out)
    inner = heads * 64
    assert qkv.shape == (B * L, 3 * inner) and qkv.stride(1) == 1 and relbias.dtype == torch.float32 and klen.dtype == torch.int32
    assert relbias.shape == (heads, 2 * L - 1) and relbias.is_contiguous() and klen.numel() == B
    if out is None:
        out = torch.empty(B * L, inner, dtype=torch.bfloat16, device=qkv.device)
    _call("vsys_t5_attention", _p(qkv), qkv.stride(0), inner, _p(relbias), _p(klen), _p(out), out.stride(0), B, L, heads)
    return out
