"""VideoSysEngine — host mirror of videosys/core/engine/{engine,mp_utils}.py.

Same surface (``VideoSysEngine(config)``, ``generate``, ``save_video``, ``shutdown``) and the same process model:
rank 0 runs in the caller's process, ranks 1..N-1 are *spawned* worker processes (one per GPU) that build the same
pipeline and execute the same method with the same pickled arguments (engine.py:23-95, mp_utils.py:181-254); a
worker that dies fails all pending calls (WorkerMonitor, mp_utils.py:111-151).  Rendezvous is tcp://127.0.0.1:<free
port> (engine.py:38) and the collective backend is RCCL ("nccl" on ROCm).
"""
from __future__ import annotations

import multiprocessing as mp
import os
import socket
import threading
import traceback
from typing import Any

import torch
import torch.distributed as dist


def get_open_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker_main(rank, world_size, init_method, config, task_q, result_q, backend):
    try:
        from . import dsp

        dsp.initialize(rank=rank, world_size=world_size, init_method=init_method, backend=backend)
        pipeline = config.pipeline_cls(config)
        result_q.put((rank, "ready", None))
        while True:
            item = task_q.get()
            if item is None:
                break
            tid, method, args, kwargs = item
            try:
                out = getattr(pipeline, method)(*args, **kwargs)
                result_q.put((rank, tid, ("ok", None if rank != 0 else out)))
            except BaseException as e:  # pickled back and re-raised in the caller (mp_utils.py:206-213)
                result_q.put((rank, tid, ("err", f"{type(e).__name__}: {e}\n{traceback.format_exc()}")))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


class VideoSysEngine:
    def __init__(self, config, backend=None):
        self.config = config
        self._tid = 0
        self._closed = False
        self._init_worker(config.pipeline_cls, backend)

    def _init_worker(self, pipeline_cls, backend):
        world_size = self.config.num_gpus
        os.environ.setdefault("OMP_NUM_THREADS", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if torch.cuda.is_available():
            assert world_size <= torch.cuda.device_count(), "num_gpus exceeds visible devices (engine.py:35)"
        init_method = f"tcp://127.0.0.1:{get_open_port()}"
        self.workers, self._task_qs = [], []
        self._result_q = None
        if world_size > 1:
            ctx = mp.get_context("spawn")
            self._result_q = ctx.Queue()
            for rank in range(1, world_size):
                q = ctx.Queue()
                p = ctx.Process(target=_worker_main, daemon=True,
                                args=(rank, world_size, init_method, self.config, q, self._result_q, backend))
                p.start()
                self.workers.append(p)
                self._task_qs.append(q)
        from . import dsp

        if world_size > 1 or not dist.is_initialized():
            dsp.initialize(rank=0, world_size=world_size, init_method=init_method, backend=backend)
        self.driver_worker = pipeline_cls(self.config)
        for _ in self.workers:
            rank, tag, _ = self._get_result()
            assert tag == "ready"

    def _get_result(self):
        while True:
            try:
                return self._result_q.get(timeout=1.0)
            except Exception:
                dead = [p for p in self.workers if not p.is_alive()]
                if dead:
                    self._kill_all()
                    raise ChildProcessError("worker died")

    def _kill_all(self):
        for p in self.workers:
            if p.is_alive():
                p.kill()

    def _run_workers(self, method: str, *args, **kwargs) -> Any:
        self._tid += 1
        for q in self._task_qs:
            q.put((self._tid, method, args, kwargs))
        driver_out = getattr(self.driver_worker, method)(*args, **kwargs)
        for _ in self.workers:
            rank, tid, (status, payload) = self._get_result()
            if status == "err":
                raise RuntimeError(f"worker rank {rank} failed: {payload}")
        return [driver_out]

    def generate(self, *args, **kwargs):
        return self._run_workers("generate", *args, **kwargs)[0]

    def save_video(self, video, output_path):
        return self.driver_worker.save_video(video, output_path)

    def shutdown(self):
        if self._closed:
            return
        self._closed = True
        for q in self._task_qs:
            q.put(None)
        for p in self.workers:
            p.join(timeout=10)
        self._kill_all()
        if dist.is_initialized():
            dist.destroy_process_group()

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass
