"""CPU (-m "not gpu"): VideoSysEngine process model over a 2-rank gloo group with a stand-in pipeline — results, error propagation
from the driver and from a worker without leaving stale results behind, sentinel-based detection of a dead worker, constructor
failures (reference behaviour: core/engine/engine.py:23-128, mp_utils.py:82-254)."""
import time

import pytest

from engine_fakes import FakeConfig


def _engine(**kw):
    from videosys_amd.engine import VideoSysEngine

    return VideoSysEngine(FakeConfig(**kw), backend="gloo")


def test_engine_two_ranks_results_and_error_isolation(capfd):
    eng = _engine(num_gpus=2)
    try:
        assert eng.generate(10) == 21.0              # (10 + 0) + (10 + 1) over the group: both ranks ran the call
        with pytest.raises(RuntimeError, match="worker rank 1 failed: KeyError"):
            eng.generate(1, mode="raise", who=1)
        assert eng.generate(2) == 5.0                # nothing of the failed call is taken for this one's result
        with pytest.raises(KeyError):
            eng.generate(1, mode="raise", who=0)     # driver-side failure: the worker's result of that call is still collected
        assert eng.generate(3) == 7.0
        assert eng.save_video(None, "x.mp4") == "x.mp4"
        # what a worker prints is led by its process name and pid on every line (mp_utils.py:154-178); the driver's is not
        assert eng.generate(4, mode="talk") == 9.0
        import time as _t

        _t.sleep(0.3)
        out = capfd.readouterr().out
        lines = [ln for ln in out.splitlines() if "says" in ln or "two lines" in ln]
        assert "rank 0 says" in lines and "two lines" in lines
        tagged = [ln for ln in lines if "VideoSysWorkerProcess-1 pid=" in ln]
        assert len(tagged) == 2 and tagged[0].endswith("rank 1 says") and tagged[1].endswith("two lines")
    finally:
        eng.shutdown()


def test_engine_dead_worker_fails_the_call_promptly():
    eng = _engine(num_gpus=2)
    try:
        t0 = time.time()
        with pytest.raises(ChildProcessError, match="worker died"):
            eng.generate(1, mode="die", who=1)
        assert time.time() - t0 < 20, "a dead worker must be noticed through its process sentinel, not by a timeout"
        with pytest.raises(ChildProcessError):       # and the engine stays failed instead of hanging
            eng.generate(1)
    finally:
        eng.shutdown()


def test_engine_reports_worker_constructor_failure():
    with pytest.raises((RuntimeError, ChildProcessError)) as ei:
        _engine(num_gpus=2, fail_init_rank=1)
    assert "constructor refused on rank 1" in str(ei.value) or "worker died" in str(ei.value)


def test_engine_driver_failure_with_peers_blocked_in_a_collective(monkeypatch):
    """rank 0 raises before a collective the workers already entered.  The workers can never report, their
    processes never exit, so neither the futures nor the monitor would ever fire: the engine must bound the wait, take the
    workers down, re-raise the DRIVER's error and refuse further calls."""
    import videosys_amd.engine as E

    monkeypatch.setattr(E, "DRIVER_FAIL_GRACE_S", 3.0)
    eng = _engine(num_gpus=2)
    try:
        t0 = time.time()
        with pytest.raises(KeyError, match="bad prompt on rank 0"):
            eng.generate(1, mode="raise_peers_in_collective", who=0)
        assert time.time() - t0 < 90          # grace (3 s) + teardown; generous, the suite may share the box with other work
        # terminated workers are reaped asynchronously — by the monitor thread or by this one, whichever gets to waitpid first; the
        # loser of that race reads "no exit code yet" for an instant (multiprocessing's Popen.poll), so liveness is polled
        deadline = time.time() + 30
        while any(p.is_alive() for p in eng.workers) and time.time() < deadline:
            time.sleep(0.05)
        assert all(not p.is_alive() for p in eng.workers)
        with pytest.raises(RuntimeError, match="engine is dead"):
            eng.generate(1)
    finally:
        eng.shutdown()
