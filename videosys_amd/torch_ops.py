"""PyTorch custom-op registration of the denoise-step kernels: ``torch.ops.videosys_amd.*``.

north_star asks for the host code to reach the HIP kernels "through PyTorch-ROCm custom ops"; SURVEY.md §8(b) lists the operator
set (adaln_modulate, gemm + epilogues, attn_spatial / attn_temporal / attn_cross, cfg_euler_step).  This module registers that set
with ``torch.library.custom_op`` — out-variant ops that mutate their ``out`` / in-place arguments, run on the current HIP stream,
never allocate, never sync and raise on CPU tensors — each a thin shim over the same C-ABI entry point ``videosys_amd.ops`` binds.
A reference maintainer can therefore call, inside the reference's own modules::

    import videosys_amd.torch_ops                      # registers the namespace
    torch.ops.videosys_amd.adaln_modulate(x, shift, scale, rows_per_sample, mod_stride, 1e-6, out)

The in-tree model (stdit3.py, latte.py, cogvideox.py) calls ``ops.*`` directly: the two routes are bit-identical
(tests/test_gpu_torch_ops.py) and the direct one costs ~10 us less host time per launch, which matters at 8-way DSP where a rank
has ~19 ms of device time per step for ~450 launches.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops

_NS = "videosys_amd"


@torch.library.custom_op(f"{_NS}::gemm", mutates_args=("out", "aux"))
def gemm(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], epilogue: int, gate: Optional[torch.Tensor],
         gate_stride: int, rows_per_sample: int, res: Optional[torch.Tensor], aux: Optional[torch.Tensor], out: torch.Tensor) -> None:
    """out = epilogue(x @ w^T + bias); epilogue 0 bias, 1 bias + GELU(tanh), 2 res + gate * (.)  (vsys_gemm_bf16).  ``res`` may be
    ``out`` itself (the residual stream updated in place, as the model does)."""
    ops.gemm(x, w, bias, epilogue=epilogue, gate=gate, gate_stride=gate_stride, rows_per_sample=rows_per_sample, res=res, aux=aux, out=out)


@torch.library.custom_op(f"{_NS}::adaln_modulate", mutates_args=("out",))
def adaln_modulate(x: torch.Tensor, shift: torch.Tensor, scale: torch.Tensor, rows_per_sample: int, mod_stride: int, eps: float,
                   out: torch.Tensor) -> None:
    """out = LayerNorm(x) * (1 + scale) + shift  (t2i_modulate(norm(x), shift, scale), open_sora_transformer_3d.py:47-48,196-197)."""
    ops.adaln_modulate(x, shift, scale, rows_per_sample, mod_stride, eps=eps, out=out)


@torch.library.custom_op(f"{_NS}::attn_prep_kv", mutates_args=("kp", "vt"))
def attn_prep_kv(k: torch.Tensor, v: torch.Tensor, k_norm_w: Optional[torch.Tensor], kp: torch.Tensor, vt: torch.Tensor, batch: int,
                 heads: int, kv_len: int, eps: float) -> None:
    """K (RMS-normed, scale folded in) head-major and V transposed: the K/V side of attn_spatial / attn_cross."""
    ops.attn_prep_kv(k, v, k_norm_w, kp, vt, batch, heads, kv_len, eps=eps)


@torch.library.custom_op(f"{_NS}::flash_attn", mutates_args=("out",))
def flash_attn(q: torch.Tensor, q_norm_w: Optional[torch.Tensor], kp: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, batch: int,
               heads: int, q_len: int, kv_len: int, eps: float) -> None:
    """attn_spatial (q_norm_w given) / attn_cross (None, kv_len = text length) of SURVEY.md §8(b): softmax(q k^T / sqrt(72)) v."""
    ops.flash_attn(q, q_norm_w, kp, vt, out, batch, heads, q_len, kv_len, eps=eps)


@torch.library.custom_op(f"{_NS}::attn_temporal", mutates_args=("out",))
def attn_temporal(qkv: torch.Tensor, C: int, q_norm_w: Optional[torch.Tensor], k_norm_w: Optional[torch.Tensor],
                  rope_cos: Optional[torch.Tensor], rope_sin: Optional[torch.Tensor], out: torch.Tensor, B: int, T: int, S: int,
                  heads: int, eps: float) -> None:
    """temporal self-attention over the T frames of every pixel token, reading the (b, t, s)-ordered rows strided (no transpose)."""
    ops.attn_temporal(qkv, C, q_norm_w, k_norm_w, rope_cos, rope_sin, out, B, T, S, heads, eps=eps)


@torch.library.custom_op(f"{_NS}::cfg_euler_step", mutates_args=("z",))
def cfg_euler_step(z: torch.Tensor, model_out: torch.Tensor, guidance: float, dt: float) -> None:
    """z += (uncond + g (cond - uncond)) * dt on the velocity half of the model output (scheduling_rflow_open_sora.py:243-252)."""
    ops.cfg_euler_step(z, model_out, guidance, dt)


@torch.library.custom_op(f"{_NS}::add_rows", mutates_args=("x",))
def add_rows(x: torch.Tensor, y: torch.Tensor) -> None:
    """x += y (the PAB broadcast residual)."""
    ops.add_rows(x, y)


OPS = ("gemm", "adaln_modulate", "attn_prep_kv", "flash_attn", "attn_temporal", "cfg_euler_step", "add_rows")
