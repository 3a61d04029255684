"""Launch programs: record the launches of a denoise step once, replay them with one C call per segment.

A denoise step of STDiT3 is ~450 calls into libvideosys_amd.so whose arguments do not change from step to step: workspaces,
weights, K/V layouts and PAB slabs are resident, per-step values (timestep, latent) live in device buffers.  Issued from Python
(wrapper checks + ctypes marshalling + hipLaunchKernel) they cost ~16 us each, 7 ms per step — nothing at 100 ms of device time,
but P-way sequence parallelism divides the device time by P and leaves the launch count alone (the reference issues every op
from Python every step, open_sora_transformer_3d.py:608-613).  ``Recorder`` runs a step eagerly ONCE while logging every C-ABI
launch (``ops._call``) as a ``vsys_cmd`` (include/videosys_amd.h "Launch programs") and every host-side action between launches
(a collective, an event record / wait between streams, a small torch copy) as a Python closure; ``Program.run`` replays the
log: consecutive launches go to ``vsys_program_run`` as ONE call (a C loop over hipLaunchKernel), closures run in between.

Contract for code that runs under a recorder:
  * every tensor whose address is handed to a launch is kept alive by the program (``keep``), so addresses stay valid;
  * anything that must happen again on every replay and is not a C-ABI launch goes through ``host_call(fn)``;
  * results of host-side decisions (PAB flags, shapes) are part of the KEY the caller files the program under — a program is
    replayed only for the exact decision pattern it was recorded with.
A launch on the stream that was current when recording started is replayed on whatever stream is current at replay time;
launches on other (side) streams are replayed on those same stream objects.
"""
from __future__ import annotations

import ctypes
import threading
from typing import Callable, List, Optional

import torch

from . import _lib

from ._opcodes import OPCODES   # entry point -> VSYS_OP_* (generated from include/videosys_amd.h by csrc/gen/program_gen.py)

_i64, _f32 = ctypes.c_int64, ctypes.c_float
N_INT, N_FLOAT = 24, 4   # VSYS_CMD_MAX_INT / VSYS_CMD_MAX_FLOAT of include/videosys_amd.h


class VsysCmd(ctypes.Structure):
    _fields_ = [("op", ctypes.c_int32), ("stream", ctypes.c_int32), ("a", _i64 * N_INT), ("f", _f32 * N_FLOAT)]


_tls = threading.local()


def active() -> Optional["Recorder"]:
    return getattr(_tls, "rec", None)


def keep(obj):
    """Keep ``obj`` (a tensor whose address was handed to a launch, a ctypes array) alive as long as the program lives."""
    rec = active()
    if rec is not None:
        rec.keep.append(obj)
    return obj


def host_call(fn: Callable[[], object]):
    """Run ``fn`` now; under a recorder also on every replay, at this position of the launch sequence and under the stream that
    is current now (the program's main stream maps to the stream current at replay time)."""
    rec = active()
    if rec is None:
        return fn()
    s = torch.cuda.current_stream() if torch.cuda.is_available() else None
    if s is None or s.cuda_stream == rec.main_handle:
        rec.items.append(fn)
    else:
        def on_side(fn=fn, s=s):
            with torch.cuda.stream(s):
                fn()
        rec.items.append(on_side)
    return fn()


class Recorder:
    """``with Recorder() as rec: <one eager step>`` then ``rec.finish()`` -> Program (or None when something unrecordable ran)."""

    def __init__(self):
        self.items: List[object] = []      # ("cmd", VsysCmd fields) tuples and host closures, in issue order
        self.keep: List[object] = []
        self.streams: List[int] = []       # slot -> raw stream handle; slot 0 = the main stream (resolved at replay)
        self.side_objs = {}                # raw handle -> torch stream object (kept alive)
        self.invalid: Optional[str] = None
        self.main_handle = torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0
        self.streams.append(self.main_handle)

    def __enter__(self):
        if active() is not None:
            raise RuntimeError("launch-program recorders do not nest")
        _tls.rec = self
        return self

    def __exit__(self, *exc):
        _tls.rec = None
        return False

    def launch(self, name: str, args, argtypes, stream_obj):
        op = OPCODES.get(name)
        if op is None:
            self.invalid = self.invalid or f"{name} has no VSYS_OP code"
            return
        h = stream_obj.cuda_stream
        if h not in self.streams:
            self.streams.append(h)
            self.side_objs[h] = stream_obj
        ints, floats = [], []
        for v, t in zip(args, argtypes):
            if t is _f32:
                floats.append(float(v))
            elif isinstance(v, ctypes.Array):      # a host array the launch reads (copy descriptors): address + keep-alive
                self.keep.append(v)
                ints.append(ctypes.addressof(v))
            else:
                ints.append(0 if v is None else int(v))
        if len(ints) > N_INT or len(floats) > N_FLOAT:
            self.invalid = self.invalid or f"{name}: too many arguments for a vsys_cmd"
            return
        self.items.append(("cmd", op, self.streams.index(h), ints, floats))

    def finish(self) -> Optional["Program"]:
        return None if self.invalid else Program(self)


class Program:
    def __init__(self, rec: Recorder):
        self.keep = rec.keep
        self.side_objs = rec.side_objs
        self.stream_handles = list(rec.streams)
        self.n_launches = sum(1 for it in rec.items if isinstance(it, tuple))
        self.segments: List[object] = []   # (ctypes array, n) | callable
        run: List[tuple] = []

        def flush():
            if run:
                arr = (VsysCmd * len(run))()
                for c, (_, op, slot, ints, floats) in zip(arr, run):
                    c.op, c.stream = op, slot
                    for k, v in enumerate(ints):
                        c.a[k] = v
                    for k, v in enumerate(floats):
                        c.f[k] = v
                self.segments.append((arr, len(run)))
                run.clear()

        for it in rec.items:
            if isinstance(it, tuple):
                run.append(it)
            else:
                flush()
                self.segments.append(it)
        flush()
        self._streams = (ctypes.c_void_p * len(self.stream_handles))(*self.stream_handles)
        self._failed = ctypes.c_int64(-1)
        self._cmd_tensors = None

    def run(self):
        lib = _lib.load()
        tv = _lib.torch_ops()
        main = torch.cuda.current_stream().cuda_stream if self.n_launches else 0   # the main stream = whatever is current NOW
        if self.n_launches:
            self._streams[0] = main
        ns = len(self.stream_handles)
        if tv is not None and self._cmd_tensors is None:
            # the dispatcher route (torch.ops.vsys.program_run, csrc/torch_binding.cpp): a CPU uint8 view of every segment's records
            self._cmd_tensors = [torch.frombuffer(seg[0], dtype=torch.uint8) if isinstance(seg, tuple) else None for seg in self.segments]
        for k, seg in enumerate(self.segments):
            if isinstance(seg, tuple):
                if tv is not None:
                    try:
                        tv.program_run(self._cmd_tensors[k], seg[1], [main] + self.stream_handles[1:])
                    except RuntimeError as e:
                        raise _lib.VsysError(f"vsys_program_run ({seg[1]}-launch segment): {str(e).splitlines()[0]}") from None
                    continue
                rc = lib.vsys_program_run(seg[0], seg[1], self._streams, ns, ctypes.byref(self._failed))
                if rc != 0:
                    _lib.check(rc, f"vsys_program_run (command {self._failed.value} of a {seg[1]}-launch segment)")
            else:
                seg()
