"""All P ranks of a sequence-parallel group inside ONE process (one thread per rank) — measurement / test infrastructure.

videosys_amd.dsp accepts, besides a torch.distributed ProcessGroup, any object that implements the collectives itself
(``size``, ``rank``, ``all_to_all_single``, ``all_gather_into_tensor``; dsp.py "Group protocol").  Two such groups live here:

  * ``LocalWorld(P).group(r)``: the ranks are threads of this process; a collective is a rendezvous (threading.Barrier) plus device
    copies between the ranks' buffers, ordered by HIP events on each rank's current stream exactly like a ProcessGroupNCCL
    collective is ordered on the stream it is issued on.  With it the REAL per-rank code of an 8-way DSP run (padded shard
    shapes, pack / unpack launches, both side streams, the collectives' issue order) executes on the one GPU of a test box and its
    result can be compared with the single-process run and with the oracle.  Works on CPU tensors too (the CPU tests of the plans).
  * ``StubGroup(P, r)``: ONE rank of a P-way group with the wire stubbed (recv <- send, a device copy of the same bytes): the
    per-rank kernel sequence for timing (tools/issue_time.py --dsp-rank).  Results are meaningless by construction.

Neither is on the product path; the product's group is RCCL.
"""
from __future__ import annotations

import threading
from typing import Callable, List

import torch


class LocalWorld:
    def __init__(self, P: int, timeout: float = 300.0):
        self.P = P
        self.barrier = threading.Barrier(P, timeout=timeout)
        self.send = [None] * P     # (tensor, event or None) deposited by each rank for the collective in flight
        self.done = [None] * P     # event: this rank's reads of the others' buffers have been enqueued ... and executed once it fires

    def group(self, rank: int) -> "LocalGroup":
        return LocalGroup(self, rank)

    def run(self, fn: Callable[[int, "LocalGroup"], object]) -> List[object]:
        """fn(rank, group) on P threads; returns the per-rank results, re-raises the first failure."""
        out, err = [None] * self.P, [None] * self.P

        def body(r):
            try:
                out[r] = fn(r, self.group(r))
            except BaseException as e:   # noqa
                err[r] = e
                self.barrier.abort()     # wake the peers instead of letting them time out

        ths = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(self.P)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        real = [e for e in err if e is not None and not isinstance(e, threading.BrokenBarrierError)]
        if real or any(err):
            raise (real or [e for e in err if e is not None])[0]
        return out


class LocalGroup:
    same_process = True     # (dsp.PeerExchange: every destination is read by launches ordered on this device)

    def __init__(self, world: LocalWorld, rank: int):
        self.world, self.rank, self.size = world, rank, world.P

    def _exchange(self, mine, read):
        """Deposit ``mine``; once every rank has, ``read(all deposits)`` enqueues this rank's copies; nobody returns before every
        rank's copies out of its deposit are ordered in front of whatever it enqueues next."""
        w, cuda = self.world, mine.is_cuda
        ev = None
        if cuda:
            ev = torch.cuda.Event()
            ev.record()                                  # on this thread's current stream: the pack kernel is in front of it
        w.send[self.rank] = (mine, ev)
        w.barrier.wait()
        cur = torch.cuda.current_stream() if cuda else None
        for src in range(w.P):
            if cuda and src != self.rank:
                cur.wait_event(w.send[src][1])
        read([t for t, _ in w.send])
        if cuda:
            d = torch.cuda.Event()
            d.record()
            w.done[self.rank] = d
        w.barrier.wait()
        if cuda:
            for r in range(w.P):
                if r != self.rank:
                    cur.wait_event(w.done[r])            # my send buffer may be rewritten only after every reader is through
        w.barrier.wait()                                 # (and the slots may be reused by the next collective)

    def all_gather_object(self, obj):
        """Set-up rendezvous of the peer-to-peer exchange (dsp.PeerExchange): every rank's object, in rank order.  One address space,
        so tensors and device addresses are handed over as they are."""
        w = self.world
        w.send[self.rank] = (obj, None)
        w.barrier.wait()
        out = [o for o, _ in w.send]
        w.barrier.wait()
        return out

    def p2p_sync(self):
        """After this rank enqueued its one-kernel exchange launch (dsp.PeerExchange): order EVERY rank's launch of the same exchange
        in front of whatever this rank enqueues next — what the kernel's own flag wait does between processes."""
        w = self.world
        ev = torch.cuda.Event()
        ev.record()
        w.send[self.rank] = (None, ev)
        w.barrier.wait()
        cur = torch.cuda.current_stream()
        for src in range(w.P):
            if src != self.rank:
                cur.wait_event(w.send[src][1])
        w.barrier.wait()

    def all_to_all_single(self, recv, send):
        P, r = self.size, self.rank
        rv = recv.view(P, -1)
        self._exchange(send, lambda sends: [rv[src].copy_(sends[src].view(P, -1)[r]) for src in range(P)])

    def all_gather_into_tensor(self, out, x):
        P = self.size
        ov = out.view(P, -1)
        self._exchange(x.contiguous(), lambda xs: [ov[src].copy_(xs[src].reshape(-1)) for src in range(P)])


class StubGroup:
    """One rank of a P-way group, wire stubbed: every collective is a device copy of its own send buffer."""

    same_process = True

    def __init__(self, P: int, rank: int = 0):
        self.size, self.rank = P, rank

    def all_gather_object(self, obj):
        return [obj] * self.size      # every "peer" is this rank: the one-kernel exchange writes the same bytes into its own tensors

    def share_flags(self, n):
        """Flag array of the one-kernel exchange with every peer folded onto this rank: the flag "of this rank in peer r's array" is
        slot r of this rank's own array, so the launch's wait sees P - 1 arrivals exactly as on a real group."""
        import ctypes

        from videosys_amd import _lib

        mine = ctypes.c_void_p()
        _lib.check(_lib.load().vsys_p2p_alloc(4 * n, 1, ctypes.byref(mine)), "vsys_p2p_alloc")
        return mine.value, [mine.value + 4 * (r - self.rank) for r in range(self.size)]

    def all_to_all_single(self, recv, send):
        recv.copy_(send)

    def all_gather_into_tensor(self, out, x):
        out.view(self.size, -1).copy_(x.reshape(1, -1).expand(self.size, -1))
