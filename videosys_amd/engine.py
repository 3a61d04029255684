"""VideoSysEngine — host mirror of videosys/core/engine/{engine,mp_utils}.py.

Same surface (``VideoSysEngine(config)``, ``generate``, ``save_video``, ``shutdown``) and the same process model:
rank 0 runs in the caller's process, ranks 1..N-1 are *spawned* worker processes (one per GPU) that build the same
pipeline and execute the same method with the same pickled arguments (engine.py:23-95, mp_utils.py:181-254).
Rendezvous is tcp://127.0.0.1:<free port> (engine.py:38) and the collective backend is RCCL ("nccl" on ROCm).

Failure handling follows mp_utils.py:82-151.  Every call is one *future* per worker, keyed (task id, rank); a collector thread
(``ResultHandler``) resolves them from the shared result queue, and a monitor thread (``WorkerMonitor``) sleeps in
``multiprocessing.connection.wait`` on the workers' process SENTINELS — it wakes the instant any worker exits, kills the
rest and fails every pending future with ``ChildProcessError("worker died")``, so a caller never hangs in a collective
whose peer is gone.  ``_run_workers`` always collects the futures of ITS task id (also when the driver's own call raised or a
worker reported an error), so nothing of one call can be mistaken for the result of the next.
"""
from __future__ import annotations

import multiprocessing as mp
import os
import socket
import threading
import traceback
from concurrent.futures import Future
from multiprocessing.connection import wait as wait_sentinels
from typing import Any, Dict, List, Tuple

import torch
import torch.distributed as dist

_TERMINATE = "TERMINATE"   # queue sentinel (mp_utils.py:17)
JOIN_TIMEOUT_S = 2
# After the DRIVER's own call raised, its peers may sit in a collective rank 0 will never enter: they get this long to report
# before the engine takes them down and re-raises the driver's error (VSYS_DRIVER_FAIL_GRACE_S overrides).
DRIVER_FAIL_GRACE_S = float(os.environ.get("VSYS_DRIVER_FAIL_GRACE_S", "10"))


def get_open_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _PrefixedStream:
    """A worker's stdout / stderr with every line led by ``(VideoSysWorkerProcess-<rank> pid=<pid>)`` (mp_utils.py:154-178,190-194):
    the output of N ranks lands in one terminal and stays attributable.  Everything else is the wrapped stream's."""

    CYAN, RESET = "\033[1;36m", "\033[0;0m"

    def __init__(self, stream, worker_name: str, pid: int):
        self._stream, self._at_line_start = stream, True
        self._prefix = f"{self.CYAN}({worker_name} pid={pid}){self.RESET} "

    def write(self, s: str):
        if not s:
            return 0
        out = []
        for piece in s.splitlines(keepends=True):
            if self._at_line_start:
                out.append(self._prefix)
            out.append(piece)
            self._at_line_start = piece.endswith("\n")
        self._stream.write("".join(out))
        return len(s)

    def __getattr__(self, name):
        return getattr(self._stream, name)


def _worker_main(rank, world_size, init_method, config, task_q, result_q, backend):
    """Worker process event loop (mp_utils.py:181-216): build the pipeline, then serve (task id, method, args, kwargs) tuples."""
    import sys

    name = mp.current_process().name
    sys.stdout, sys.stderr = _PrefixedStream(sys.stdout, name, os.getpid()), _PrefixedStream(sys.stderr, name, os.getpid())
    try:
        try:
            from . import dsp

            dsp.initialize(rank=rank, world_size=world_size, init_method=init_method, backend=backend)
            pipeline = config.pipeline_cls(config)
        except BaseException as e:   # construction failed: say why before exiting (the monitor reports the exit itself)
            result_q.put((rank, 0, ("err", f"{type(e).__name__}: {e}\n{traceback.format_exc()}")))
            raise
        result_q.put((rank, 0, ("ok", None)))
        for item in iter(task_q.get, _TERMINATE):
            tid, method, args, kwargs = item
            try:
                out = getattr(pipeline, method)(*args, **kwargs)
                result_q.put((rank, tid, ("ok", out if rank == 0 else None)))
            except BaseException as e:  # reported to the caller, which re-raises (mp_utils.py:206-213)
                result_q.put((rank, tid, ("err", f"{type(e).__name__}: {e}\n{traceback.format_exc()}")))
    except KeyboardInterrupt:
        pass
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


class ResultHandler(threading.Thread):
    """Resolves the per-(task, rank) futures from the result queue (mp_utils.py:82-108)."""

    def __init__(self, result_q):
        super().__init__(daemon=True)
        self.result_q = result_q
        self.tasks: Dict[Tuple[int, int], Future] = {}
        self._lock = threading.Lock()
        self._dead = False

    def expect(self, tid: int, rank: int) -> Future:
        fut: Future = Future()
        with self._lock:
            if self._dead:
                fut.set_exception(ChildProcessError("worker died"))
            else:
                self.tasks[(tid, rank)] = fut
        return fut

    def run(self):
        for rank, tid, (status, payload) in iter(self.result_q.get, _TERMINATE):
            with self._lock:
                fut = self.tasks.pop((tid, rank), None)
            if fut is None:
                continue   # nobody waits for it (a call that was abandoned): drop, never hand it to a later call
            if status == "ok":
                fut.set_result(payload)
            else:
                fut.set_exception(RuntimeError(f"worker rank {rank} failed: {payload}"))
        with self._lock:   # make sure every waiter gets an exception
            self._dead = True
            pending, self.tasks = self.tasks, {}
        for fut in pending.values():
            fut.set_exception(ChildProcessError("worker died"))

    def close(self):
        self.result_q.put(_TERMINATE)


class WorkerMonitor(threading.Thread):
    """Blocks on the workers' process sentinels; the first exit takes everything down (mp_utils.py:111-151)."""

    def __init__(self, workers: List[mp.Process], handler: ResultHandler):
        super().__init__(daemon=True)
        self.workers, self.handler = workers, handler
        self._close = False
        self.dead_exit_codes: Dict[str, int] = {}

    def run(self):
        dead = wait_sentinels([p.sentinel for p in self.workers])   # returns as soon as ANY worker process has exited
        if not self._close:
            self._close = True
            for p in self.workers:
                if p.sentinel in dead:
                    p.join(JOIN_TIMEOUT_S)
                if p.exitcode is not None and p.exitcode != 0:
                    self.dead_exit_codes[p.name] = p.exitcode
            for p in self.workers:
                if p.is_alive():
                    p.kill()
            self.handler.close()   # after the workers are gone: fails whatever is still pending
        for p in self.workers:
            p.join(JOIN_TIMEOUT_S)

    def close(self):
        if self._close:
            return
        self._close = True
        for p in self.workers:
            if p.is_alive():
                p.terminate()
        self.handler.close()


class VideoSysEngine:
    def __init__(self, config, backend=None):
        self.config = config
        self._tid = 0
        self._closed = False
        self._failed = None   # set when a driver-side failure forced the workers down: later calls fail fast
        self.parallel_worker_tasks = None
        self._init_worker(config.pipeline_cls, backend)

    def _init_worker(self, pipeline_cls, backend):
        world_size = self.config.num_gpus
        os.environ.setdefault("OMP_NUM_THREADS", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if torch.cuda.is_available():
            assert world_size <= torch.cuda.device_count(), "num_gpus exceeds visible devices (engine.py:35)"
        init_method = f"tcp://127.0.0.1:{get_open_port()}"
        self.workers, self._task_qs = [], []
        self._handler = self._monitor = None
        ready: List[Future] = []
        if world_size > 1:
            ctx = mp.get_context("spawn")
            result_q = ctx.Queue()
            self._handler = ResultHandler(result_q)
            for rank in range(1, world_size):
                q = ctx.Queue()
                p = ctx.Process(target=_worker_main, daemon=True, name=f"VideoSysWorkerProcess-{rank}",
                                args=(rank, world_size, init_method, self.config, q, result_q, backend))
                ready.append(self._handler.expect(0, rank))
                p.start()
                self.workers.append(p)
                self._task_qs.append(q)
            self._handler.start()
            self._monitor = WorkerMonitor(self.workers, self._handler)
            self._monitor.start()
        from . import dsp

        try:
            if world_size > 1 or not dist.is_initialized():
                dsp.initialize(rank=0, world_size=world_size, init_method=init_method, backend=backend)
            self.driver_worker = pipeline_cls(self.config)
            for fut in ready:
                fut.result()   # raises the worker's construction error, or "worker died"
        except BaseException:
            self.shutdown()
            raise

    def _run_workers(self, method: str, *args, **kwargs) -> Any:
        if self._failed is not None:
            raise RuntimeError(f"engine is dead: an earlier call failed on the driver ({self._failed})")
        if self._handler is not None and self._handler._dead:
            raise ChildProcessError("worker died")   # fail fast: the driver must not enter a collective whose peers are gone
        self._tid += 1
        tid = self._tid
        futures = [self._handler.expect(tid, rank) for rank in range(1, len(self.workers) + 1)] if self._handler else []
        for q in self._task_qs:
            q.put((tid, method, args, kwargs))
        failure = None
        driver_out = None
        try:
            driver_out = getattr(self.driver_worker, method)(*args, **kwargs)
        except BaseException as e:
            failure = e
        if failure is not None and futures:
            # The driver left the call early (OOM, bad input, ...): a worker that already entered a collective waits for a peer
            # that will never arrive, never exits, and so never wakes the monitor.  Give the workers a bounded grace period to
            # report (they may have failed the same way), then take them down, mark the engine dead and re-raise the DRIVER's
            # error — the reference propagates it immediately (engine.py:85-95).
            import time
            from concurrent.futures import TimeoutError as FutTimeout

            deadline = time.monotonic() + DRIVER_FAIL_GRACE_S
            hung = False
            for fut in futures:
                try:
                    fut.result(timeout=max(0.0, deadline - time.monotonic()))
                except FutTimeout:
                    hung = True
                    break
                except BaseException:
                    pass
            if hung:
                self._failed = f"{type(failure).__name__}: {failure}"
                self._teardown_workers()
            raise failure
        for fut in futures:   # always collect what belongs to THIS call before returning or raising
            try:
                fut.result()
            except BaseException as e:
                failure = failure or e
        if failure is not None:
            raise failure
        return [driver_out]

    def _teardown_workers(self):
        """Kill every worker now (no TERMINATE handshake: they may be blocked inside a collective)."""
        if self._monitor is not None:
            self._monitor.close()
        for p in self.workers:
            if p.is_alive():
                p.kill()
        for p in self.workers:
            p.join(JOIN_TIMEOUT_S)

    def generate(self, *args, **kwargs):
        return self._run_workers("generate", *args, **kwargs)[0]

    def stop_remote_worker_execution_loop(self) -> None:
        """engine.py:103-111 of the reference: waits for worker tasks started with ``async_run_remote_workers_only``.  This engine
        never leaves such tasks behind (every _run_workers call collects its futures, with the bounded wait after a driver-side
        failure), so there is nothing to wait for; kept so that reference call sites run unchanged."""
        self.parallel_worker_tasks = None

    def _driver_execute_model(self, *args, **kwargs):
        """engine.py:97-98: the call on the driver's own pipeline only (no worker is told)."""
        return self.driver_worker.generate(*args, **kwargs)

    def _wait_for_tasks_completion(self, parallel_worker_tasks: Any) -> None:
        """engine.py:113-117: block on futures of worker tasks (``Future.result`` here, ``ResultFuture.get`` there)."""
        for result in parallel_worker_tasks or ():
            result.get() if hasattr(result, "get") else result.result()

    def save_video(self, video, output_path):
        return self.driver_worker.save_video(video, output_path)

    def shutdown(self):
        if self._closed:
            return
        self._closed = True
        for q in self._task_qs:
            try:
                q.put(_TERMINATE)
            except Exception:
                pass
        for p in self.workers:
            p.join(timeout=10)
        if self._monitor is not None:
            self._monitor.close()
        for p in self.workers:
            if p.is_alive():
                p.kill()
        if dist.is_initialized():
            dist.destroy_process_group()

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass
