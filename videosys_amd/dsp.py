"""Dynamic Sequence Parallelism over RCCL/xGMI — host mirror of videosys/core/distributed/{parallel_mgr,comm}.py.

Reference behaviour kept (file:line relative to /root/reference/videosys/core/distributed):
  * ``initialize``            parallel_mgr.py:103-117  (one process per GPU, backend "nccl" == RCCL on ROCm)
  * ``ParallelManager``       parallel_mgr.py:14-39    (dp x cp x sp mesh; built on dist.new_group, no colossalai)
  * ``set_pad/get_pad``       comm.py:268-279
  * ``split_sequence``        comm.py:148-167,256-258  (zero pad + local slice)
  * ``gather_sequence``       comm.py:170-190,260-262  (all-gather + un-pad)
  * ``all_to_all_with_pad``   comm.py:104-108,282-304  (the DSP layout switch)

MI355X design: the reference's tensor_split / P x .contiguous() / list all_to_all / cat / .contiguous() (two extra
full copies around every NCCL call) becomes ONE packed send buffer -> ``all_to_all_single`` -> ONE unpack, with the
zero-padding / narrowing folded into the pack/unpack copies (``vsys_copy_4d``).  xGMI is point-to-point, so the
all-to-all uses all 7 links of a GPU at once; a single large message per peer is the efficient shape.  The pack/unpack
*plans* (pure Python: which strided region goes where) are separated from the copy *executor* so the plans are
testable with gloo on CPU; on a GPU the executor is the HIP kernel and nothing else.
"""
from __future__ import annotations

import logging
from dataclasses import dataclass
from typing import Callable, List, Optional

import torch
import torch.distributed as dist

PAD_DICT = {}


# ---------------------------------------------------------------------------------------------------------
# Group protocol.  ``group`` is a torch.distributed ProcessGroup (RCCL on a GPU node, gloo in the CPU tests) OR any object that
# implements the collectives itself: ``size``, ``rank``, ``all_to_all_single(recv, send)``, ``all_gather_into_tensor(out, x)``
# (stream-ordered on torch's current stream, like a ProcessGroupNCCL collective).  tools/local_group.py uses the second form to
# run all P ranks of a DSP group inside one process on one GPU (tests) and to stub the wire for per-rank timing; it is also
# where a peer-to-peer (IPC) exchange would plug in.
# ---------------------------------------------------------------------------------------------------------
def group_size(group) -> int:
    return group.size if hasattr(group, "all_to_all_single") else dist.get_world_size(group)


def group_rank(group) -> int:
    return group.rank if hasattr(group, "all_to_all_single") else dist.get_rank(group)


def all_to_all_single(recv, send, group):
    """Stream-ordered all-to-all on torch's current stream.  A host-side action: under a launch-program recorder it is logged
    and re-issued on every replay (program.host_call); the buffers are the SequenceParallel object's resident staging buffers."""
    from . import program

    def issue():
        with COMM_TIMER.comm():
            if hasattr(group, "all_to_all_single"):
                group.all_to_all_single(recv, send)
            else:
                dist.all_to_all_single(recv, send, group=group)

    program.keep(recv), program.keep(send)
    program.host_call(issue)


def all_gather_into_tensor(out, x, group):
    from . import program

    def issue():
        with COMM_TIMER.comm():
            if hasattr(group, "all_gather_into_tensor"):
                group.all_gather_into_tensor(out, x)
            else:
                dist.all_gather_into_tensor(out, x, group=group)

    program.keep(out), program.keep(x)
    program.host_call(issue)


def initialize(rank=0, world_size=1, init_method=None, backend: Optional[str] = None):
    """parallel_mgr.py:103-117.  backend defaults to "nccl" (RCCL) when a GPU is present, "gloo" otherwise."""
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, init_method=init_method, world_size=world_size, rank=rank)
        if torch.cuda.is_available():
            torch.cuda.set_device(rank % torch.cuda.device_count())
        if not logging.getLogger().handlers:   # (:115 init_logger()) INFO lines on rank 0, silence on the others — unless the
            from .utils import init_logger     # application configured logging itself, which is then left alone

            init_logger()


class ParallelManager:
    """dp x cp x sp process-group mesh (row-major rank = (dp*cp_size + cp)*sp_size + sp), parallel_mgr.py:14-39."""

    def __init__(self, dp_size, cp_size, sp_size):
        self.dp_size, self.cp_size, self.sp_size = dp_size, cp_size, sp_size
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        assert dp_size * cp_size * sp_size == world, f"mesh {dp_size}x{cp_size}x{sp_size} != world {world}"
        self.dp_group = self.cp_group = self.sp_group = None
        self.dp_rank, self.cp_rank, self.sp_rank = 0, None, None
        if world > 1:
            coords = lambda r: (r // (cp_size * sp_size), (r // sp_size) % cp_size, r % sp_size)
            me = coords(rank)
            for axis, size in ((0, dp_size), (1, cp_size), (2, sp_size)):
                # every rank must create every group (collective), keep the one it belongs to
                seen = {}
                for r in range(world):
                    c = coords(r)
                    key = tuple(v for i, v in enumerate(c) if i != axis)
                    seen.setdefault(key, []).append(r)
                for key, ranks in sorted(seen.items()):
                    g = dist.new_group(ranks) if size > 1 else None
                    if key == tuple(v for i, v in enumerate(me) if i != axis) and g is not None:
                        if axis == 0:
                            self.dp_group, self.dp_rank = g, dist.get_rank(g)
                        elif axis == 1:
                            self.cp_group, self.cp_rank = g, dist.get_rank(g)
                        else:
                            self.sp_group, self.sp_rank = g, dist.get_rank(g)
        logging.info(f"Init parallel manager with dp_size: {dp_size}, cp_size: {cp_size}, sp_size: {sp_size}")


def set_pad(name: str, dim_size: int, parallel_group) -> None:
    sp_size = dist.get_world_size(parallel_group)
    PAD_DICT[name] = (sp_size - (dim_size % sp_size)) % sp_size


def get_pad(name) -> int:
    return PAD_DICT[name]


# ---------------------------------------------------------------------------------------------------------
# copy plans: a plan is a list of CopyOp(src_offset, dst_offset, n0, n1, n2, run, sstr, dstr, n1_valid, n2_valid)
# over flat bf16 buffers; "run" contiguous elements are moved per (i0, i1, i2).
# ---------------------------------------------------------------------------------------------------------
@dataclass
class CopyOp:
    src_off: int
    dst_off: int
    n0: int
    n1: int
    n2: int
    run: int
    sstr: tuple
    dstr: tuple
    n1_valid: int
    n2_valid: int


def plan_split(B, T, S, C, P, rank):
    """split_sequence(x[B,T,S,C], dim=2): local [B,T,Sl,C], Sl = (S+pad)/P, zero rows past S."""
    Sl = (S + (P - S % P) % P) // P
    valid = max(0, min(Sl, S - rank * Sl))
    return [CopyOp(rank * Sl * C, 0, B, T, Sl, C, (T * S * C, S * C, C), (T * Sl * C, Sl * C, C), T, valid)], (B, T, Sl, C)


def _chunk(Tp, chunk):
    c0, c1 = (0, Tp) if chunk is None else chunk
    assert 0 <= c0 < c1 <= Tp, (chunk, Tp)
    return c0, c1 - c0


def plan_switch_to_temporal_shard(B, T, Sl, S, C, P, chunk=None):
    """[B,T,Sl,C] (S-shard) -> send [P][B,Tc,Sl,C]; after all_to_all_single recv [P][B,Tc,Sl,C] -> [B,Tc,S,C].
    (all_to_all_with_pad with scatter_dim=1 (T, padded), gather_dim=2 (S, un-padded); open_sora_transformer_3d.py:299-303)
    Rank r owns frames [r*Tp, (r+1)*Tp), Tp = ceil(T/P).  ``chunk = (c0, c1)`` moves only frames c0..c1-1 of every rank's block
    (Tc = c1 - c0; default the whole block): two chunks on two streams overlap one's exchange with the other's attention."""
    Tp = (T + (P - T % P) % P) // P
    c0, Tc = _chunk(Tp, chunk)
    run = Sl * C
    pack = [CopyOp((r * Tp + c0) * run, r * B * Tc * run, B, Tc, 1, run, (T * run, run, run), (Tc * run, run, run),
                   max(0, min(Tc, T - r * Tp - c0)), 1) for r in range(P)]
    unpack = []
    for src in range(P):
        valid = max(0, min(Sl, S - src * Sl))
        # narrowing side: copy only the valid columns (never zero-fill past the row end)
        unpack.append(CopyOp(src * B * Tc * run, src * Sl * C, B, Tc, valid, C, (Tc * run, run, C), (Tc * S * C, S * C, C),
                             Tc, valid))
    unpack = [u for u in unpack if u.n2 > 0]
    return pack, unpack, (P, B, Tc, Sl, C), (B, Tc, S, C)


def plan_switch_to_spatial_shard(B, Tp, T, S, Sl, C, P, chunk=None):
    """[B,Tc,S,C] (T-shard; frames c0..c1-1 of this rank's block of Tp) -> send [P][B,Tc,Sl,C]; recv -> the frames
    src*Tp + c0 .. of [B,T,Sl,C] (scatter_dim=2 padded, gather_dim=1 narrowed)."""
    c0, Tc = _chunk(Tp, chunk)
    run = Sl * C
    pack = []
    for r in range(P):
        valid = max(0, min(Sl, S - r * Sl))
        pack.append(CopyOp(r * Sl * C, r * B * Tc * run, B, Tc, Sl, C, (Tc * S * C, S * C, C), (Tc * run, run, C), Tc, valid))
    unpack = []
    for src in range(P):
        valid = max(0, min(Tc, T - src * Tp - c0))
        if valid > 0:
            unpack.append(CopyOp(src * B * Tc * run, (src * Tp + c0) * run, B, valid, 1, run, (Tc * run, run, run),
                                 (T * run, run, run), valid, 1))
    return pack, unpack, (P, B, Tc, Sl, C), (B, T, Sl, C)


def plan_p2p_to_temporal_shard(B, T, Sl, S, C, P, rank, chunk=None):
    """The to-temporal-shard switch as ONE copy per peer, straight from this rank's [B,T,Sl,C] into peer r's [B,Tc,S,C]: pack op r of
    plan_switch_to_temporal_shard composed with the unpack op for source ``rank`` that peer r would run (zero fill for the padded
    frames from the pack side, the narrowing of the padded columns from the unpack side).  Entry r is None when nothing travels."""
    Tp = (T + (P - T % P) % P) // P
    c0, Tc = _chunk(Tp, chunk)
    run = Sl * C
    valid_s = max(0, min(Sl, S - rank * Sl))             # columns of MY shard that exist (the last rank's may be padding)
    ops = []
    for r in range(P):
        if valid_s == 0:
            ops.append(None)
            continue
        ops.append(CopyOp((r * Tp + c0) * run, rank * Sl * C, B, Tc, valid_s, C, (T * run, run, C), (Tc * S * C, S * C, C),
                          max(0, min(Tc, T - r * Tp - c0)), valid_s))
    return ops, (B, Tc, S, C)


def plan_p2p_to_spatial_shard(B, Tp, T, S, Sl, C, P, rank, chunk=None):
    """The to-spatial-shard switch as ONE copy per peer: this rank's [B,Tc,S,C] (frames c0.. of its block of Tp) into the frames
    rank*Tp + c0 .. of peer r's [B,T,Sl,C] (columns past S zero-filled on the padded shard, frames past T never written)."""
    c0, Tc = _chunk(Tp, chunk)
    run = Sl * C
    valid_t = max(0, min(Tc, T - rank * Tp - c0))        # frames of MY block that exist
    ops = []
    for r in range(P):
        if valid_t == 0:
            ops.append(None)
            continue
        valid_s = max(0, min(Sl, S - r * Sl))
        ops.append(CopyOp(r * Sl * C, (rank * Tp + c0) * run, B, valid_t, Sl, C, (Tc * S * C, S * C, C), (T * run, run, C),
                          valid_t, valid_s))
    return ops, (B, T, Sl, C)


def plan_gather(B, T, Sl, S, C, P):
    """gather_sequence(dim=2): recv [P][B,T,Sl,C] -> [B,T,S,C] (un-padded)."""
    ops = []
    for src in range(P):
        valid = max(0, min(Sl, S - src * Sl))
        if valid > 0:
            ops.append(CopyOp(src * B * T * Sl * C, src * Sl * C, B, T, valid, C, (T * Sl * C, Sl * C, C), (T * S * C, S * C, C),
                              T, valid))
    return ops, (B, T, S, C)


def frames_per_rank(B: int, T: int, P: int, scatter: str = "flat") -> int:
    """Attention problems (frames) the busiest rank holds in the T-shard phase.  "sample" = the reference's layout: T is padded
    and scattered per sample (comm.py:282-304 on the [B,T,S,C] view): B * ceil(T/P).  "flat" = the (sample, frame) axis is
    scattered as ONE axis of B*T frames — spatial attention and the row-wise qkv GEMM do not care which sample a frame
    belongs to: ceil(B*T/P) (config 2, P = 8: 5 instead of 6 frames against 4.75 ideal)."""
    return -(-B * T // P) if scatter == "flat" else B * -(-T // P)


def choose_spatial_switch(B: int, T: int, S: int, C: int, P: int, gemm_tflops: float = 750.0, a2a_gbytes_s: float = 300.0,
                          overlapped: bool = True, scatter: str = "sample") -> dict:
    """Cost model for the layout switch around the spatial attention of one block on one rank (bf16, critical-path rank).

    ``"activations"`` (the reference's order, open_sora_transformer_3d.py:208-216): the C-wide modulated activations travel,
    the qkv GEMM runs on the T-shard — whose frame count is PADDED to ceil(T/P) per rank, so the busiest rank multiplies
    B*ceil(T/P)*S rows instead of B*T*S/P.  ``"qkv"``: the qkv GEMM runs at rest on the un-padded S-shard and the 3C-wide q|k|v
    travels (3x the message of the first exchange; the way back is the same).  Returns the estimated extra time of each order
    over the ideal and the cheaper one; constants are the measured per-rank GEMM rate at M = 4864 and a conservative share of
    7 x 153 GB/s of xGMI; with the two CFG samples overlapped only the un-hidden half of the extra traffic is charged."""
    Tp = -(-T // P)
    Sl = -(-S // P)
    nfr = frames_per_rank(B, T, P, scatter)
    rows_padded, rows_rest = nfr * S, B * T * Sl
    gemm_extra_s = 2.0 * max(rows_padded - rows_rest, 0) * C * (3 * C) / (gemm_tflops * 1e12)
    msg_bytes = (P - 1) / P * nfr * P * Sl * C * 2             # what leaves a rank in the first exchange (C wide)
    comm_extra_s = 2.0 * msg_bytes / (a2a_gbytes_s * 1e9) * (0.5 if overlapped else 1.0)
    best = "qkv" if comm_extra_s < gemm_extra_s else "activations"
    return dict(order=best, padded_frames_per_rank=nfr, gemm_extra_us=gemm_extra_s * 1e6, comm_extra_us=comm_extra_s * 1e6,
                first_message_mb=msg_bytes / 1e6)


class CommTimer:
    """HIP-event brackets around the collectives of one rank (bench.py --gpus N reports them per rank): ``with timer.comm():``
    around a collective records its device time; ``report()`` sums them after a synchronize."""

    def __init__(self):
        self.events = []
        self.enabled = False

    def comm(self):
        return _CommSpan(self) if self.enabled and torch.cuda.is_available() else _NullSpan()

    def reset(self):
        self.events = []

    def report(self) -> dict:
        ms = [s.elapsed_time(e) for s, e in self.events]
        return dict(collectives=len(ms), comm_ms=sum(ms))


class _NullSpan:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _CommSpan:
    def __init__(self, timer):
        self.timer = timer

    def __enter__(self):
        self.s = torch.cuda.Event(enable_timing=True)
        self.e = torch.cuda.Event(enable_timing=True)
        self.s.record()   # on the stream the collective is enqueued on (torch's current stream)
        return self

    def __exit__(self, *a):
        self.e.record()
        self.timer.events.append((self.s, self.e))
        return False


COMM_TIMER = CommTimer()


def hip_copy_executor(src: torch.Tensor, dst: torch.Tensor, ops: List[CopyOp]):
    """The product executor: ONE vsys_copy_4d_batch launch for the whole plan (device tensors only)."""
    from . import ops as vops

    if not ops:
        return
    vops.copy_4d_batch(src.view(-1), dst.view(-1), [(o.src_off, o.dst_off, o.n0, o.n1, o.n2, o.run, *o.sstr, *o.dstr, o.n1_valid,
                                                      o.n2_valid) for o in ops])


# ---------------------------------------------------------------------------------------------------------
# One-kernel peer-to-peer exchange (vsys_p2p_exchange, csrc/p2p.hip): every rank stores its rows straight into the peers' DESTINATION
# tensors over xGMI and waits, in the same launch, for the peers' rows to land in its own.  A "peers" object provides the two
# collective set-up calls (every rank, same order): ``all_gather_object(obj) -> [obj of rank 0 .. P-1]``; in-process groups
# (tools/local_group.py) hand tensors and addresses over as they are, IpcPeers below maps them through HIP IPC for one process per GPU.
#
# Why a destination tensor may be overwritten without an acknowledgement.  A spatial block has two sites: X (to the temporal shard,
# destination: the attention input) and Y (back, destination: the projection input).  Peer r writes my X destination for block n+1 only
# after ITS Y kernel of block n returned, which waited for MY flag of Y(n); I raise that flag inside my Y(n) kernel, which my stream runs
# after the consumers of X(n) (qkv GEMM, K/V prep, attention).  The same argument with X and Y swapped covers the Y destination, and
# chunked switches (two side streams) use one site pair per chunk.  PAB decisions depend on the timestep only, so every rank skips the
# same sites in the same steps and the per-site sequence numbers stay in step.
# ---------------------------------------------------------------------------------------------------------
class IpcPeers:
    """Set-up side of the peer-to-peer exchange for ONE PROCESS PER GPU: tensors travel as torch's CUDA-IPC descriptors
    (torch.multiprocessing.reductions), flag arrays as 64-byte HIP IPC handles (vsys_p2p_ipc_export / _open), both through
    ``dist.all_gather_object`` on ``group``.  Needs HSA_ENABLE_IPC_MODE_LEGACY=0 on this driver (dmabuf IPC)."""

    def __init__(self, group):
        self.group = group
        self.size, self.rank = dist.get_world_size(group), dist.get_rank(group)

    def all_gather_object(self, obj):
        out = [None] * self.size
        dist.all_gather_object(out, obj, group=self.group)
        return out

    def share_tensor(self, t: torch.Tensor) -> List[torch.Tensor]:
        """Collective.  A rank whose export fails still takes part in the all-gather (with None), so nobody is left waiting in it; the
        error is raised on EVERY rank afterwards."""
        from torch.multiprocessing.reductions import reduce_tensor

        try:
            payload, err = reduce_tensor(t), None
        except Exception as e:     # noqa: BLE001 (whatever the IPC export raises on this driver)
            payload, err = None, e
        every = self.all_gather_object(payload)
        if any(v is None for v in every):
            bad = [q for q, v in enumerate(every) if v is None]
            raise RuntimeError(f"HIP IPC export of a tensor failed on rank(s) {bad}" + (f": {err}" if err is not None else ""))
        return [t if q == self.rank else f(*a) for q, (f, a) in enumerate(every)]

    def share_flags(self, n: int):
        """Collective, same failure discipline as share_tensor."""
        import ctypes

        from . import _lib

        lib = _lib.load()
        mine = ctypes.c_void_p()
        payload, err = None, None
        try:
            _lib.check(lib.vsys_p2p_alloc(4 * n, 1, ctypes.byref(mine)), "vsys_p2p_alloc")
            h = (ctypes.c_char * 64)()
            _lib.check(lib.vsys_p2p_ipc_export(mine, h), "vsys_p2p_ipc_export")
            payload = bytes(h)
        except Exception as e:     # noqa: BLE001
            err = e
        every = self.all_gather_object(payload)
        if any(v is None for v in every):
            if mine.value:
                lib.vsys_p2p_free(mine)
            bad = [q for q, v in enumerate(every) if v is None]
            raise RuntimeError(f"HIP IPC export of a flag array failed on rank(s) {bad}" + (f": {err}" if err is not None else ""))
        ptrs = []
        for q, hb in enumerate(every):
            if q == self.rank:
                ptrs.append(mine.value)
                continue
            peer = ctypes.c_void_p()
            _lib.check(lib.vsys_p2p_ipc_open(ctypes.create_string_buffer(hb, 64), ctypes.byref(peer)), "vsys_p2p_ipc_open")
            ptrs.append(peer.value)
        return mine.value, ptrs


def _share_tensor(peers, t):
    if hasattr(peers, "share_tensor"):
        return peers.share_tensor(t)
    return peers.all_gather_object(t)            # one address space: the tensors themselves


def _share_flags(peers, n):
    if hasattr(peers, "share_flags"):
        return peers.share_flags(n)
    import ctypes

    from . import _lib

    mine = ctypes.c_void_p()
    _lib.check(_lib.load().vsys_p2p_alloc(4 * n, 1, ctypes.byref(mine)), "vsys_p2p_alloc")
    return mine.value, peers.all_gather_object(mine.value)


class PeerExchange:
    """The exchange sites of one rank.  ``exchange(key, src, out, ops)``: ops[r] (CopyOp or None) moves rows of ``src`` into peer r's
    ``out`` of the same site; returns when the launch is enqueued — the launch itself ends when this rank's ``out`` is complete."""

    def __init__(self, peers, P: int, rank: int):
        import os

        self.peers, self.P, self.rank = peers, P, rank
        self.sites = {}
        # (a bound against hanging for ever on a dead peer, not a pace: ranks may reach their first exchange tens of seconds apart)
        self.timeout_ticks = int(float(os.environ.get("VSYS_P2P_TIMEOUT_S", "120")) * 1e8)    # 100 MHz wall clock
        self.launches = 0

    def _site(self, key, out):
        # A resident workspace that GROWS (stdit3._buf: a later call asks for more frames under the same name) is a new allocation on
        # every rank at the same call — the requests are a function of the shapes, which all ranks share — so the allocation's size is
        # part of the site's identity: growth opens a fresh site collectively, on all ranks together.
        key = (key, out.untyped_storage().nbytes(), out.storage_offset())
        st = self.sites.get(key)
        if st is not None and st["out"].data_ptr() != out.data_ptr():
            # Creating a site is COLLECTIVE (tensor descriptors and flag handles are all-gathered): a rank that re-created one on its own
            # because ITS destination buffer moved would sit in that all-gather alone.  The callers hand over resident workspaces keyed
            # by shape and allocation size (above), so what is left is an allocation that moved on THIS rank only: say so instead of
            # hanging (VSYS_P2P_RESHARE=1 when every rank is known to re-create together).
            import os

            if os.environ.get("VSYS_P2P_RESHARE") != "1" and not bool(getattr(self.peers, "same_process", False)):
                raise RuntimeError(f"peer-to-peer exchange site {key}: the destination tensor of rank {self.rank} moved "
                                   f"({st['out'].data_ptr():#x} -> {out.data_ptr():#x}); a site is shared collectively and cannot be "
                                   "re-created by one rank (keep the destination resident, or VSYS_DSP_P2P=0)")
            self._release(st)                                        # (no leak: the old flag array and mappings go back first)
            st = None
        if st is None:
            outs = _share_tensor(self.peers, out)                    # collective: every rank creates its sites in the same order
            my_flags, flag_ptrs = _share_flags(self.peers, self.P)
            st = dict(out=out, outs=outs, my_flags=my_flags, flag_ptrs=flag_ptrs,
                      state=torch.zeros(19 * 32, dtype=torch.int32, device=out.device))   # one 128-byte line per word (p2p.hip)
            self.sites[key] = st
        return st

    def exchange(self, key, src, out, ops):
        """ops[r]: None, a CopyOp, or a list of CopyOps — the rows of ``src`` that go into peer r's ``out`` (r == rank: this rank's own)."""
        import ctypes

        from . import ops as vops
        from . import program

        st = self._site(key, out)
        flat = []
        same_process = bool(getattr(self.peers, "same_process", False))     # ranks as threads / a stub: plain (cached) stores
        for r in range(self.P):
            mine = ops[r]
            mine = [] if mine is None else (list(mine) if isinstance(mine, (list, tuple)) else [mine])
            flag = 0 if r == self.rank else st["flag_ptrs"][r] + 4 * self.rank
            remote = int(r != self.rank and not same_process)
            for o in mine:
                flat += [o.src_off, o.dst_off, o.n0, o.n1, o.n2, o.run, *o.sstr, *o.dstr, o.n1_valid, o.n2_valid,
                         st["outs"][r].data_ptr(), flag, remote]
            if not mine and r != self.rank:     # nothing of mine travels to r (a fully padded shard): an EMPTY problem still raises my flag there
                flat += [0, 0, 0, 0, 0, 8, 0, 0, 0, 0, 0, 0, 0, 0, st["outs"][r].data_ptr(), flag, remote]
        n = len(flat) // 17
        assert n <= 16, "one launch carries at most 16 problems (VSYS_COPY_BATCH_MAX)"
        arr = (ctypes.c_int64 * len(flat))(*[int(v) for v in flat])
        program.keep(st)
        # ranks that are threads of ONE process (tools/local_group) order the launches on the host instead of polling flags on the
        # device: their streams share a handful of hardware queues, where a polling kernel can sit in front of the launch it waits for
        host_sync = getattr(self.peers, "p2p_sync", None)
        with COMM_TIMER.comm():   # (bench.py --gpus N: the launch ends when the slowest peer's rows are here, like a collective)
            vops._call("vsys_p2p_exchange", vops._p(src), n, arr, st["my_flags"], self.P, self.rank, vops._p(st["state"]),
                       -1 if host_sync is not None else self.timeout_ticks)
        if host_sync is not None:
            program.host_call(host_sync)
        self.launches += 1

    def close(self):
        """Give the flag arrays back (this rank's allocation; the mappings of the peers' arrays when they came through HIP IPC).  The
        ranks must have finished their last exchange (a collective barrier, or the end of the run) before anybody closes."""
        for st in self.sites.values():
            self._release(st)
        self.sites.clear()

    def _release(self, st):
        import ctypes

        from . import _lib

        lib = _lib.load()
        if isinstance(self.peers, IpcPeers):
            for q, ptr in enumerate(st["flag_ptrs"]):
                if q != self.rank and ptr:
                    lib.vsys_p2p_ipc_close(ctypes.c_void_p(ptr))
        if st.get("my_flags"):
            lib.vsys_p2p_free(ctypes.c_void_p(st["my_flags"]))
        st["my_flags"] = 0
        st["flag_ptrs"] = []

    def check(self):
        """Raise if any exchange timed out waiting for a peer (state[31] of a site; synchronises the device)."""
        for key, st in self.sites.items():
            err = int(st["state"][18 * 32].item())
            if err:
                raise RuntimeError(f"peer-to-peer exchange {key}: rank {self.rank} never received the rows of rank {err - 1} "
                                   f"(VSYS_P2P_TIMEOUT_S); set VSYS_DSP_P2P=0 to use the RCCL all_to_all_single path")


def _agree(group, ok: bool, why: str = ""):
    """Collective: every rank learns every rank's (ok, why); returns (all ok, first reason)."""
    P = dist.get_world_size(group)
    every = [None] * P
    dist.all_gather_object(every, (bool(ok), str(why)[:300]), group=group)
    bad = [(q, w) for q, (o, w) in enumerate(every) if not o]
    return (not bad), ("" if not bad else f"rank {bad[0][0]}: {bad[0][1]}")


def p2p_selftest(group, device=None, rows: int = 768, C: int = 1152, rounds: int = 4, timing_iters: int = 20) -> dict:
    """One-time guarded trial of the one-kernel peer-to-peer exchange over a REAL process group, before a model may use it (VERDICT r5
    item 3: the first multi-GPU run should exercise vsys_p2p_exchange without anybody setting an environment variable, and must not be
    able to corrupt a video or hang if the path does not work on that machine).  Collective: every rank of ``group`` calls it at the
    same point (SequenceParallel / UlyssesParallel construction).

      1. set-up: scratch source / destination tensors [P, rows, C] bf16 and flag arrays shared through HIP IPC (IpcPeers) — any rank's
         failure is agreed on collectively;
      2. ``rounds`` exchanges of a payload that is a function of (sender, receiver, round, row): the receiver compares every peer's block
         with what that peer must have sent (a later launch on the stream, as the model's consumers are) — a stale or torn row, or a
         flag that never arrives inside VSYS_P2P_SELFTEST_TIMEOUT_S (default 5 s; the kernel's own wall-clock bound), fails the test;
      3. the ranks agree (all-gather of the verdicts): one failure anywhere sends EVERY rank to the RCCL path;
      4. on a pass, ``timing_iters`` exchanges of the same 1.77 MB-per-peer message (config 2's size at 8 ranks) each way — the
         one-kernel exchange against pack + all_to_all_single + unpack on the same buffers, HIP-event timed, slowest rank counts — and
         the faster one is chosen.
    Returns {"selftest": "pass" | "fail: ...", "exchange_path": "p2p" | "rccl", "p2p_ms": .., "rccl_ms": .., "message_mb_per_peer": ..}.
    VSYS_P2P_SELFTEST_FAULT = mismatch | timeout (tests): rank 0 expects a wrong payload / rank P-1 skips its first exchange."""
    import os
    import time

    P, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    fault = os.environ.get("VSYS_P2P_SELFTEST_FAULT", "")
    info = {"selftest": "pass", "exchange_path": "rccl", "p2p_ms": None, "rccl_ms": None,
            "message_mb_per_peer": round(rows * C * 2 / 1e6, 3), "ranks": P}
    saved = os.environ.get("VSYS_P2P_TIMEOUT_S")
    os.environ["VSYS_P2P_TIMEOUT_S"] = os.environ.get("VSYS_P2P_SELFTEST_TIMEOUT_S", "5")
    px, ok, why = None, True, ""
    try:
        px = PeerExchange(IpcPeers(group), P, rank)
    finally:
        if saved is None:
            os.environ.pop("VSYS_P2P_TIMEOUT_S", None)
        else:
            os.environ["VSYS_P2P_TIMEOUT_S"] = saved
    src = torch.empty(P, rows, C, dtype=torch.bfloat16, device=dev)
    dst = torch.zeros(P, rows, C, dtype=torch.bfloat16, device=dev)
    n = rows * C
    plan = [CopyOp(q * n, rank * n, 1, rows, 1, C, (n, C, C), (n, C, C), rows, 1) for q in range(P)]
    row_id = torch.arange(rows, device=dev, dtype=torch.float32)[:, None]

    def payload(sender, receiver, rnd):     # small integers: exact in bf16, different for every (sender, receiver, round, row)
        return ((row_id + (sender * 131 + receiver * 17 + rnd * 7)) % 251.0).to(torch.bfloat16).expand(rows, C)

    try:                                     # (1) collective set-up of the site; a local failure is agreed on below
        px._site("selftest", dst)
    except Exception as e:                   # noqa: BLE001
        ok, why = False, f"set-up: {type(e).__name__}: {e}"
    ok, why = _agree(group, ok, why)
    for rnd in range(rounds if ok else 0):   # (2) patterned rounds; (3) the verdict of EVERY round is agreed on by all ranks (which is
        try:                                 #     also the barrier that keeps a destination from being overwritten while a peer compares)
            for q in range(P):
                src[q].copy_(payload(rank, q, rnd))
            if not (fault == "timeout" and rnd == 0 and rank == P - 1):
                px.exchange("selftest", src, dst, plan)
            torch.cuda.synchronize(dev)
            px.check()
            for q in range(P):
                want = payload(q, rank, rnd + (1 if fault == "mismatch" and rank == 0 and q != rank else 0))
                if not torch.equal(dst[q], want):
                    bad = int((dst[q] != want).any(dim=1).sum())
                    raise RuntimeError(f"round {rnd}: {bad} of {rows} rows from rank {q} differ from what it sent")
        except Exception as e:               # noqa: BLE001
            ok, why = False, f"{type(e).__name__}: {e}"
        ok, why = _agree(group, ok, why)
        if not ok:
            break
    if not ok:
        info["selftest"] = "fail: " + why
    else:                                    # (4) both paths on the same message, slowest rank counts
        send = torch.empty(P, rows, C, dtype=torch.bfloat16, device=dev)
        recv = torch.empty_like(send)
        ident = [CopyOp(q * n, q * n, 1, rows, 1, C, (n, C, C), (n, C, C), rows, 1) for q in range(P)]

        def rccl_once():
            hip_copy_executor(src, send, ident)
            dist.all_to_all_single(recv, send, group=group)
            hip_copy_executor(recv, dst, ident)

        def timed(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize(dev)
            dist.barrier(group)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(timing_iters):
                fn()
            e1.record()
            torch.cuda.synchronize(dev)
            every = [None] * P
            dist.all_gather_object(every, e0.elapsed_time(e1) / timing_iters, group=group)
            return max(every)

        try:
            t_p2p = timed(lambda: px.exchange("selftest", src, dst, plan))
            px.check()
            t_rccl = timed(rccl_once)
            info.update(p2p_ms=round(t_p2p, 4), rccl_ms=round(t_rccl, 4), exchange_path="p2p" if t_p2p <= t_rccl else "rccl")
        except Exception as e:               # noqa: BLE001  (a failure here is local: the ranks agree once more before anybody acts on it)
            ok, why = False, f"timing: {type(e).__name__}: {e}"
        ok2, why2 = _agree(group, ok, why)
        if not ok2:
            info.update(selftest="fail: " + why2, exchange_path="rccl")
    torch.cuda.synchronize(dev)
    dist.barrier(group)                      # every rank has left its last exchange: the scratch sites may go
    if px is not None:
        px.close()
    if rank == 0:
        logging.info(f"peer-to-peer exchange self-test over {P} ranks: {info}")
    return info


def _make_peer_exchange(group, P, rank, copy_executor, info: Optional[dict] = None):
    """Which exchange a SequenceParallel / UlyssesParallel object uses.  In-process groups (tools/local_group: no wire of their own):
    peer-to-peer unless VSYS_DSP_P2P=0.  A torch ProcessGroup: VSYS_DSP_P2P = 0 -> pack + all_to_all_single + unpack; 1 -> peer-to-peer
    (the self-test still runs first and a failure raises); auto (default) -> ``p2p_selftest`` decides: the one-kernel exchange when it
    passes AND is the faster of the two on this machine, RCCL otherwise.  VSYS_P2P_SELFTEST=0 skips the trial (auto then means RCCL).
    Every rank takes the same path: the setting is compared and the verdict agreed collectively.  ``info`` receives what happened."""
    import os

    info = {} if info is None else info
    mode = os.environ.get("VSYS_DSP_P2P", "auto")
    if copy_executor is not hip_copy_executor or P < 2:
        info.update(exchange_path="rccl", selftest="not applicable (no HIP copy executor / single rank)")
        return None
    if hasattr(group, "all_gather_object"):
        info.update(exchange_path="rccl" if mode == "0" else "p2p", selftest="not applicable (in-process group)")
        return None if mode == "0" else PeerExchange(group, P, rank)
    # a torch ProcessGroup: every rank must take the same path (a rank that waits for flags nobody raises would sit out its timeout,
    # one that waits in all_to_all_single for peers that never call it would hang) — compare the setting once, collectively
    every = [None] * P
    dist.all_gather_object(every, (mode, os.environ.get("VSYS_P2P_SELFTEST", "1")), group=group)
    if any(v != every[0] for v in every):
        raise RuntimeError(f"VSYS_DSP_P2P / VSYS_P2P_SELFTEST differ between the ranks of the sequence-parallel group: {every} (rank order); set them identically")
    if mode == "0" or not torch.cuda.is_available():
        info.update(exchange_path="rccl", selftest="skipped (VSYS_DSP_P2P=0)" if mode == "0" else "skipped (no device)")
        return None
    if os.environ.get("VSYS_P2P_SELFTEST", "1") == "0":
        info.update(exchange_path="p2p" if mode == "1" else "rccl", selftest="skipped (VSYS_P2P_SELFTEST=0)")
        return PeerExchange(IpcPeers(group), P, rank) if mode == "1" else None
    res = p2p_selftest(group)
    info.update(res)
    if mode == "1":
        if res["selftest"] != "pass":
            raise RuntimeError(f"VSYS_DSP_P2P=1 but the peer-to-peer self-test failed: {res['selftest']}")
        info["exchange_path"] = "p2p"
        return PeerExchange(IpcPeers(group), P, rank)
    if res["selftest"] != "pass":
        logging.warning(f"peer-to-peer exchange self-test failed ({res['selftest']}): using pack + all_to_all_single + unpack")
        return None
    return PeerExchange(IpcPeers(group), P, rank) if res["exchange_path"] == "p2p" else None


def check_exchange(model) -> None:
    """Raise if a peer-to-peer exchange of ``model``'s sequence-parallel object timed out (PeerExchange.check).  A timed-out site keeps
    going with an incomplete destination tensor and only sets a sticky error word, so the PRODUCT path asks once per generate(), after
    the last denoise step and before the decode (one device synchronisation per site; nothing when the RCCL path is in use)."""
    sp = getattr(model, "_sp", None)
    p2p = getattr(sp, "p2p", None) if sp is not None else None
    if p2p is not None:
        p2p.check()


class SequenceParallel:
    """The DSP data path of one rank: owns the packed send/recv buffers and the side stream."""

    def __init__(self, group, copy_executor: Callable = hip_copy_executor):
        self.group = group
        self.P = group_size(group)
        self.rank = group_rank(group)
        self.exec = copy_executor
        self._bufs = {}
        self.exchange_info = {}   # what the set-up decided (self-test verdict, both timings, the path): bench.py reports it
        self.p2p = _make_peer_exchange(group, self.P, self.rank, copy_executor, self.exchange_info)   # None: pack + all_to_all_single + unpack

    def _buf(self, name, shape, like):
        key = (name, tuple(shape))
        b = self._bufs.get(key)
        if b is None or b.device != like.device or b.dtype != like.dtype:
            b = torch.empty(shape, dtype=like.dtype, device=like.device)
            self._bufs[key] = b
        return b

    def split(self, x):
        B, T, S, C = x.shape
        ops, shape = plan_split(B, T, S, C, self.P, self.rank)
        out = torch.empty(shape, dtype=x.dtype, device=x.device)
        self.exec(x, out, ops)
        return out

    def gather(self, x, S):
        B, T, Sl, C = x.shape
        recv = self._buf("gather_recv", (self.P, B, T, Sl, C), x)
        all_gather_into_tensor(recv.view(self.P * B, T, Sl, C), x.contiguous(), self.group)
        ops, shape = plan_gather(B, T, Sl, S, C, self.P)
        out = torch.empty(shape, dtype=x.dtype, device=x.device)
        self.exec(recv, out, ops)
        return out

    def to_temporal_shard(self, x, S, out=None, tag="", chunk=None):
        """[B,T,Sl,C] -> [B,Tc,S,C]  (before spatial attention).  ``tag`` selects a private pair of staging buffers (two
        switches in flight on different streams must not share them); ``chunk`` as in plan_switch_to_temporal_shard."""
        B, T, Sl, C = x.shape
        if self.p2p is not None and x.is_cuda and out is not None:      # ONE launch: my rows straight into the peers' ``out``
            ops_, oshape = plan_p2p_to_temporal_shard(B, T, Sl, S, C, self.P, self.rank, chunk)
            assert tuple(out.shape) == oshape and out.is_contiguous() and x.is_contiguous()
            self.p2p.exchange(("T", tag, tuple(x.shape), S, chunk), x, out, ops_)
            return out
        pack, unpack, sshape, oshape = plan_switch_to_temporal_shard(B, T, Sl, S, C, self.P, chunk)
        send = self._buf(f"a2a_send{tag}", sshape, x)
        recv = self._buf(f"a2a_recv{tag}", sshape, x)
        self.exec(x, send, pack)
        all_to_all_single(recv, send, self.group)
        if out is None:
            out = torch.empty(oshape, dtype=x.dtype, device=x.device)
        self.exec(recv, out, unpack)
        return out

    def to_spatial_shard(self, x, T, Sl, out=None, tag="", chunk=None, Tp=None):
        """[B,Tc,S,C] -> frames of [B,T,Sl,C]  (after spatial attention).  With a ``chunk`` pass the rank's whole block size
        ``Tp`` and a caller-owned ``out`` (the chunks of a block write disjoint frames of the same buffer)."""
        B, Tc, S, C = x.shape
        Tp = Tc if Tp is None else Tp
        if self.p2p is not None and x.is_cuda and out is not None:
            ops_, oshape = plan_p2p_to_spatial_shard(B, Tp, T, S, Sl, C, self.P, self.rank, chunk)
            assert tuple(out.shape) == oshape and out.is_contiguous() and x.is_contiguous()
            self.p2p.exchange(("S", tag, tuple(x.shape), T, Sl, chunk, Tp), x, out, ops_)
            return out
        pack, unpack, sshape, oshape = plan_switch_to_spatial_shard(B, Tp, T, S, Sl, C, self.P, chunk)
        send = self._buf(f"a2a_send{tag}", sshape, x)
        recv = self._buf(f"a2a_recv{tag}", sshape, x)
        self.exec(x, send, pack)
        all_to_all_single(recv, send, self.group)
        if out is None:
            assert chunk is None, "a chunked switch writes part of the frames: the caller owns the output buffer"
            out = torch.empty(oshape, dtype=x.dtype, device=x.device)
        self.exec(recv, out, unpack)
        return out


# ---------------------------------------------------------------------------------------------------------
# Ulysses head/sequence exchange for CogVideoX's joint [text | video] attention
# (cogvideox_transformer_3d.py:45-86 _remove/_add_extra_encoder, :112-123 and :160-165 all_to_all_comm).
# At rest a rank holds rows [text Lt | its video shard Lvl] of every sample; attention needs the FULL sequence for H/P heads.
# The reference sends the (replicated) text rows through the all-to-all and drops P-1 copies afterwards; here the text rows of
# q, k, v never travel (every rank already has them for all heads) and travel once on the way back (a rank computed the
# attention of the text rows for its own heads only).
# ---------------------------------------------------------------------------------------------------------
def plan_heads_scatter(B, Lt, Lvl, Lv, C, P, rank):
    """qkv local [B, Lt+Lvl, 3C] -> send [P][B][Lvl][3][C/P]; recv (same shape, indexed by source) + local text ->
    qkv_h [B, Lt+Lv, 3*C/P] (q | k | v column blocks of this rank's heads, full sequence, padding rows dropped)."""
    hw = C // P
    Ll, L = Lt + Lvl, Lt + Lv
    pack = [CopyOp(Lt * 3 * C + r * hw, r * B * Lvl * 3 * hw, B, Lvl, 3, hw, (Ll * 3 * C, 3 * C, C), (Lvl * 3 * hw, 3 * hw, hw), Lvl, 3)
            for r in range(P)]
    unpack = [CopyOp(rank * hw, 0, B, Lt, 3, hw, (Ll * 3 * C, 3 * C, C), (L * 3 * hw, 3 * hw, hw), Lt, 3)]  # from the LOCAL qkv
    unpack_recv = []
    for s in range(P):
        valid = max(0, min(Lvl, Lv - s * Lvl))
        if valid > 0:
            unpack_recv.append(CopyOp(s * B * Lvl * 3 * hw, (Lt + s * Lvl) * 3 * hw, B, valid, 1, 3 * hw, (Lvl * 3 * hw, 3 * hw, 3 * hw),
                                      (L * 3 * hw, 3 * hw, 3 * hw), valid, 1))
    return pack, unpack, unpack_recv, (P, B, Lvl, 3, hw), (B, L, 3 * hw)


def plan_heads_gather(B, Lt, Lvl, Lv, C, P):
    """attention output of this rank's heads [B, Lt+Lv, C/P] -> send [P][B][Lt+Lvl][C/P] (text rows to everyone, video rows
    to their owner, zero rows past Lv); recv -> local [B, Lt+Lvl, C] (column block s from source s)."""
    hw = C // P
    Ll, L = Lt + Lvl, Lt + Lv
    pack = []
    for r in range(P):
        base = r * B * Ll * hw
        pack.append(CopyOp(0, base, B, Lt, 1, hw, (L * hw, hw, hw), (Ll * hw, hw, hw), Lt, 1))
        valid = max(0, min(Lvl, Lv - r * Lvl))
        pack.append(CopyOp((Lt + r * Lvl) * hw if valid > 0 else 0, base + Lt * hw, B, Lvl, 1, hw, (L * hw, hw, hw), (Ll * hw, hw, hw), valid, 1))
    unpack = [CopyOp(s * B * Ll * hw, s * hw, B, Ll, 1, hw, (Ll * hw, hw, hw), (Ll * C, C, C), Ll, 1) for s in range(P)]
    return pack, unpack, (P, B, Ll, hw), (B, Ll, C)


def plan_p2p_heads_scatter(B, Lt, Lvl, Lv, C, P, rank):
    """scatter_heads as copies straight into the peers' [B, Lt+Lv, 3C/P]: entry r = the head slice r of this rank's video rows into
    peer r's rows Lt + rank*Lvl ..; entry ``rank`` additionally carries this rank's own text rows (which never travel)."""
    hw = C // P
    Ll, L = Lt + Lvl, Lt + Lv
    valid = max(0, min(Lvl, Lv - rank * Lvl))
    ops = []
    for r in range(P):
        mine = []
        if valid > 0:
            mine.append(CopyOp(Lt * 3 * C + r * hw, (Lt + rank * Lvl) * 3 * hw, B, valid, 3, hw, (Ll * 3 * C, 3 * C, C), (L * 3 * hw, 3 * hw, hw),
                               valid, 3))
        if r == rank:
            mine.append(CopyOp(rank * hw, 0, B, Lt, 3, hw, (Ll * 3 * C, 3 * C, C), (L * 3 * hw, 3 * hw, hw), Lt, 3))
        ops.append(mine or None)
    return ops, (B, L, 3 * hw)


def plan_p2p_heads_gather(B, Lt, Lvl, Lv, C, P, rank):
    """gather_heads as copies straight into the peers' [B, Lt+Lvl, C] (column block ``rank``): to every peer the text rows of this
    rank's heads, and to peer r the video rows r owns (zero rows past Lv on the padded shard)."""
    hw = C // P
    Ll, L = Lt + Lvl, Lt + Lv
    ops = []
    for r in range(P):
        valid = max(0, min(Lvl, Lv - r * Lvl))
        ops.append([CopyOp(0, rank * hw, B, Lt, 1, hw, (L * hw, hw, hw), (Ll * C, C, C), Lt, 1),
                    CopyOp((Lt + r * Lvl) * hw if valid > 0 else 0, Lt * C + rank * hw, B, Lvl, 1, hw, (L * hw, hw, hw), (Ll * C, C, C), valid, 1)])
    return ops, (B, Ll, C)


class UlyssesParallel:
    """Head <-> sequence exchange of one rank (packed buffers + all_to_all_single, like SequenceParallel)."""

    def __init__(self, group, copy_executor: Callable = hip_copy_executor):
        self.group = group
        self.P = group_size(group)
        self.rank = group_rank(group)
        self.exec = copy_executor
        self._bufs = {}
        self.exchange_info = {}
        self.p2p = _make_peer_exchange(group, self.P, self.rank, copy_executor, self.exchange_info)   # None: pack + all_to_all_single + unpack
        if self.p2p is not None and 2 * self.P > 16:
            self.p2p = None        # (the gather carries two problems per peer: at most 8 ranks in one launch)
            self.exchange_info["exchange_path"] = "rccl"

    _buf = SequenceParallel._buf

    def shard_len(self, Lv):
        return -(-Lv // self.P)

    def scatter_heads(self, qkv, B, Lt, Lv, C, out=None):
        """qkv [B*(Lt+Lvl), 3C] (local rows) -> [B*(Lt+Lv), 3C/P] (this rank's heads, whole sequence)."""
        Lvl = self.shard_len(Lv)
        if self.p2p is not None and qkv.is_cuda and out is not None:
            ops_, oshape = plan_p2p_heads_scatter(B, Lt, Lvl, Lv, C, self.P, self.rank)
            assert tuple(out.shape) == oshape and out.is_contiguous() and qkv.is_contiguous()
            self.p2p.exchange(("HS", B, Lt, Lv, C), qkv, out, ops_)
            return out.view(oshape[0] * oshape[1], oshape[2])
        pack, unpack_local, unpack_recv, sshape, oshape = plan_heads_scatter(B, Lt, Lvl, Lv, C, self.P, self.rank)
        send = self._buf("u_send", sshape, qkv)
        recv = self._buf("u_recv", sshape, qkv)
        self.exec(qkv, send, pack)
        all_to_all_single(recv, send, self.group)
        if out is None:
            out = torch.empty(oshape, dtype=qkv.dtype, device=qkv.device)
        self.exec(qkv, out, unpack_local)
        self.exec(recv, out, unpack_recv)
        return out.view(oshape[0] * oshape[1], oshape[2])

    def gather_heads(self, ao, B, Lt, Lv, C, out=None):
        """ao [B*(Lt+Lv), C/P] -> [B*(Lt+Lvl), C] (all heads, local rows)."""
        Lvl = self.shard_len(Lv)
        if self.p2p is not None and ao.is_cuda and out is not None:
            ops_, oshape = plan_p2p_heads_gather(B, Lt, Lvl, Lv, C, self.P, self.rank)
            assert tuple(out.shape) == oshape and out.is_contiguous() and ao.is_contiguous()
            self.p2p.exchange(("HG", B, Lt, Lv, C), ao, out, ops_)
            return out.view(oshape[0] * oshape[1], oshape[2])
        pack, unpack, sshape, oshape = plan_heads_gather(B, Lt, Lvl, Lv, C, self.P)
        send = self._buf("g_send", sshape, ao)
        recv = self._buf("g_recv", sshape, ao)
        self.exec(ao, send, pack)
        all_to_all_single(recv, send, self.group)
        if out is None:
            out = torch.empty(oshape, dtype=ao.dtype, device=ao.device)
        self.exec(recv, out, unpack)
        return out.view(oshape[0] * oshape[1], oshape[2])
