"""videosys/core/distributed/comm.py mirror — the function-level DSP operators, for code written against the reference's names:
``split_sequence`` / ``gather_sequence`` (comm.py:148-190, 256-262), ``all_to_all_comm`` (:104-143), ``all_to_all_with_pad``
(:282-304), ``split_from_second_dim`` / ``gather_from_second_dim`` (:307-318), ``set_pad`` / ``get_pad`` / ``PAD_DICT`` (:268-279).

Same results as the reference functions on any tensor rank, dimension and group (RCCL on a GPU node, gloo in the CPU tests);
forward only — this is an inference build, so ``grad_scale`` is accepted and has nothing to scale, and a tensor that carries a
graph is refused.  The transport differs: one packed send buffer, ONE ``all_to_all_single`` / ``all_gather_into_tensor`` and one
unpack, instead of the reference's P ``.contiguous()`` slices + list collective + ``cat`` + ``.contiguous()``: xGMI is
point-to-point, one large message per peer is its efficient shape.  The models of this build do not go through these generic
entry points: their layout switches run on resident staging buffers with the padding folded into the HIP pack / unpack copies
(dsp.SequenceParallel, dsp.UlyssesParallel)."""
from __future__ import annotations

import torch
import torch.distributed as dist

from .dsp import PAD_DICT, all_gather_into_tensor, all_to_all_single, get_pad, group_rank, group_size, set_pad  # noqa: F401


def _group(process_group):
    return dist.group.WORLD if process_group is None else process_group


def _no_graph(x):
    if torch.is_grad_enabled() and x.requires_grad:
        raise NotImplementedError("videosys_amd.comm is forward-only (inference build): call under torch.no_grad()")


def _all_to_all_func(input_, world_size, group, scatter_dim, gather_dim):
    """comm.py:104-108: equal slices of ``scatter_dim`` go one to each rank, the received slices are joined along ``gather_dim``."""
    if input_.shape[scatter_dim] % world_size:
        raise ValueError(f"dimension to scatter ({input_.shape[scatter_dim]}) is not divisible by the group size ({world_size})")
    send = torch.stack(input_.tensor_split(world_size, scatter_dim))       # the pack: [P, *slice]
    recv = torch.empty_like(send)
    all_to_all_single(recv, send, group)
    return torch.cat(recv.unbind(0), dim=gather_dim)                        # the unpack


def all_to_all_comm(input_, process_group=None, scatter_dim=2, gather_dim=1):
    _no_graph(input_)
    g = _group(process_group)
    return _all_to_all_func(input_, group_size(g), g, scatter_dim, gather_dim)


def _split_sequence_func(input_, pg, dim: int, pad: int, pad_val: int = 0):
    """comm.py:148-167: pad ``dim`` by ``pad`` entries of ``pad_val``, keep this rank's equal slice."""
    P, r = group_size(pg), group_rank(pg)
    if P == 1:
        return input_
    n = input_.size(dim) + max(pad, 0)
    assert n % P == 0, f"dim_size ({n}) is not divisible by world_size ({P})"
    per = n // P
    lo, hi = r * per, min((r + 1) * per, input_.size(dim))
    if hi - lo == per:                                   # entirely inside the tensor: no padded copy of the whole input
        return input_.narrow(dim, lo, per).contiguous()
    shape = list(input_.shape)
    shape[dim] = per
    out = input_.new_full(shape, pad_val)
    if hi > lo:
        out.narrow(dim, 0, hi - lo).copy_(input_.narrow(dim, lo, hi - lo))
    return out


def _gather_sequence_func(input_, pg, dim: int, pad: int):
    """comm.py:170-190: all-gather, join along ``dim``, drop the last ``pad`` entries."""
    input_ = input_.contiguous()
    P = group_size(pg)
    if P == 1:
        return input_
    buf = input_.new_empty((P * input_.shape[0],) + tuple(input_.shape[1:]))     # rank-major: already the result for dim 0
    all_gather_into_tensor(buf, input_, pg)
    out = torch.cat(buf.view((P,) + tuple(input_.shape)).unbind(0), dim=dim) if dim % input_.dim() else buf
    return out.narrow(dim, 0, out.size(dim) - pad) if pad > 0 else out


def split_sequence(input_, process_group, dim, grad_scale=1.0, pad=0, pad_val=0):
    _no_graph(input_)
    return _split_sequence_func(input_, _group(process_group), dim, pad, pad_val)


def gather_sequence(input_, process_group, dim, grad_scale=1.0, pad=0):
    _no_graph(input_)
    return _gather_sequence_func(input_, _group(process_group), dim, pad)


def all_to_all_with_pad(input_, process_group, scatter_dim: int = 2, gather_dim: int = 1, scatter_pad: int = 0, gather_pad: int = 0):
    """comm.py:282-304: zero-pad ``scatter_dim`` by ``scatter_pad``, all-to-all, drop the last ``gather_pad`` of ``gather_dim``."""
    _no_graph(input_)
    g = _group(process_group)
    if scatter_pad > 0:
        shape = list(input_.shape)
        shape[scatter_dim] = scatter_pad
        input_ = torch.cat([input_, input_.new_zeros(shape)], dim=scatter_dim)
    P = group_size(g)
    assert input_.shape[scatter_dim] % P == 0, \
        f"Dimension to scatter ({input_.shape[scatter_dim]}) is not divisible by world size ({P})"
    out = _all_to_all_func(input_, P, g, scatter_dim, gather_dim)
    return out.narrow(gather_dim, 0, out.size(gather_dim) - gather_pad) if gather_pad > 0 else out


def split_from_second_dim(x, batch_size, parallel_group):
    """comm.py:307-311: [(b t), ...] -> this rank's frames [(b t / P), ...] (pad = the registered "temporal" pad)."""
    x = x.view(batch_size, -1, *x.shape[1:])
    x = split_sequence(x, parallel_group, dim=1, grad_scale="down", pad=get_pad("temporal"))
    return x.reshape(-1, *x.shape[2:])


def gather_from_second_dim(x, batch_size, parallel_group):
    """comm.py:314-318: the inverse of split_from_second_dim."""
    x = x.view(batch_size, -1, *x.shape[1:])
    x = gather_sequence(x, parallel_group, dim=1, grad_scale="up", pad=get_pad("temporal"))
    return x.reshape(-1, *x.shape[2:])
