"""Pyramid Attention Broadcast policy — host mirror of videosys/core/pab/pab_mgr.py (reference :6-232).

Same public names, arguments and decisions as the reference module so configs drop in unchanged (pinned call for call against
the reference module in tests/test_pab_cpu.py).  The caches themselves are pre-allocated HBM slabs owned by the transformer
blocks; this module only decides.  The timestep handed to the decision functions is a Python int the sampler already has on the
host (``all_timesteps``), so no device->host sync happens per block (the reference does ``int(timestep[0])`` per block,
open_sora_transformer_3d.py:188,190,232,244).

Structure (differs from the reference on purpose): the three attention policies are one table-driven rule; the MLP policy is
expressed as *windows* — a window opens at a configured timestep ``k`` (the block computes and stores its MLP output) and covers
the next ``skip_count`` timesteps of the schedule (the block replays the stored output; the entry is dropped at the window's end).
"""
from __future__ import annotations

import logging
from typing import Dict, Optional, Sequence, Tuple

PAB_MANAGER = None

_ATTN_KINDS = ("cross", "spatial", "temporal")
_FIELDS = tuple(f"{k}_{s}" for k in _ATTN_KINDS for s in ("broadcast", "threshold", "range")) + (
    "mlp_broadcast", "mlp_spatial_broadcast_config", "mlp_temporal_broadcast_config")


class PABConfig:
    """pab_mgr.py:6-41 — identical kwargs/defaults; ``steps`` is filled in by ``update_steps``."""

    def __init__(self, cross_broadcast: bool = False, cross_threshold: list = None, cross_range: int = None,
                 spatial_broadcast: bool = False, spatial_threshold: list = None, spatial_range: int = None,
                 temporal_broadcast: bool = False, temporal_threshold: list = None, temporal_range: int = None,
                 mlp_broadcast: bool = False, mlp_spatial_broadcast_config: dict = None, mlp_temporal_broadcast_config: dict = None):
        given = locals()
        for name in _FIELDS:
            setattr(self, name, given[name])
        self.steps = None
        self.mlp_spatial_outputs: Dict[Tuple[int, int], object] = {}
        self.mlp_temporal_outputs: Dict[Tuple[int, int], object] = {}

    def attn_rule(self, kind: str):
        return getattr(self, kind + "_broadcast"), getattr(self, kind + "_range"), getattr(self, kind + "_threshold")

    def mlp_rule(self, is_temporal: bool):
        return self.mlp_temporal_broadcast_config if is_temporal else self.mlp_spatial_broadcast_config

    def mlp_store(self, is_temporal: bool):
        return self.mlp_temporal_outputs if is_temporal else self.mlp_spatial_outputs


def _mlp_window(schedule: Sequence[int], t, rule: dict) -> Optional[Tuple[int, int]]:
    """The (first, last) timesteps of the first configured window of ``rule`` that contains ``t`` (pab_mgr.py:93-106: a window is
    the configured timestep plus the ``skip_count`` schedule entries after it; keys missing from the schedule are ignored; a
    window reaching past the schedule's end still matches its existing entries but cannot be named, as in the reference, which
    raises IndexError there)."""
    for first in rule:
        if first not in schedule:
            continue
        at = schedule.index(first)
        n = int(rule[first]["skip_count"])
        if t in schedule[at:at + 1 + n]:
            return schedule[at], schedule[at + n]
    return None


class PABManager:
    """pab_mgr.py:43-181."""

    def __init__(self, config: PABConfig):
        self.config: PABConfig = config
        logging.info("Init Pyramid Attention Broadcast. " + " ".join(
            "%s: %s/%s/%s" % ((k,) + config.attn_rule(k)) for k in ("spatial", "temporal", "cross")) + f" mlp: {config.mlp_broadcast}")

    # ---- attention policies: inside the threshold window, recompute every ``range``-th call and broadcast in between
    def _attn(self, kind: str, timestep, count: int):
        on, every, (lo, hi) = self._rule(kind)
        reuse = bool(on) and timestep is not None and count % every != 0 and lo < timestep < hi
        return reuse, (count + 1) % self.config.steps

    def _rule(self, kind):
        on, every, thr = self.config.attn_rule(kind)
        return on, every, (thr if thr is not None else (0, 0))

    def if_broadcast_cross(self, timestep: int, count: int):
        return self._attn("cross", timestep, count)

    def if_broadcast_temporal(self, timestep: int, count: int):
        return self._attn("temporal", timestep, count)

    def if_broadcast_spatial(self, timestep: int, count: int):
        return self._attn("spatial", timestep, count)

    # ---- MLP policy (pab_mgr.py:108-141).  Unlike the reference's STDiT3.forward (which never hands ``all_timesteps`` to its
    # blocks and therefore raises TypeError whenever mlp_broadcast=True) the callers here pass it through.
    def if_skip_mlp(self, timestep: int, count: int, block_idx: int, all_timesteps, is_temporal=False):
        if not self.config.mlp_broadcast:
            return False, None, False, None
        rule = self.config.mlp_rule(is_temporal)
        window = _mlp_window(all_timesteps, timestep, rule)
        skip_range = list(window) if window is not None else self._last_seen_range(all_timesteps, rule)
        if timestep is None:
            return False, count, False, skip_range
        opens_here = timestep in rule and block_idx in rule[timestep]["block"]
        if opens_here:                       # compute now, keep the output for the window
            return False, count + 1, True, skip_range
        if window is not None and block_idx in rule[window[0]]["block"]:
            return True, 0, False, skip_range   # replay the stored output
        return False, count, False, skip_range

    @staticmethod
    def _last_seen_range(schedule, rule):
        """What the reference's loop leaves in ``skip_range`` when no window matches: the schedule slice of the LAST configured
        timestep that is on the schedule (or None) — returned verbatim so callers see the same 4-tuple."""
        last = None
        for first in rule:
            if first in schedule:
                at = schedule.index(first)
                last = schedule[at:at + 1 + int(rule[first]["skip_count"])]
        return last

    def save_skip_output(self, timestep, block_idx, ff_output, is_temporal=False):
        self.config.mlp_store(is_temporal)[(timestep, block_idx)] = ff_output

    def get_mlp_output(self, skip_range, timestep, block_idx, is_temporal=False):
        store, key = self.config.mlp_store(is_temporal), (skip_range[0], block_idx)
        if store.get(key) is None:
            raise ValueError(f"No stored MLP output found | t {timestep} |[{skip_range[0]}, {skip_range[-1]}] | block {block_idx}")
        return store.pop(key) if timestep == skip_range[-1] else store[key]


def set_pab_manager(config: PABConfig):
    global PAB_MANAGER
    PAB_MANAGER = PABManager(config) if config is not None else None


def enable_pab() -> bool:
    c = PAB_MANAGER.config if PAB_MANAGER is not None else None
    return c is not None and any(bool(c.attn_rule(k)[0]) for k in _ATTN_KINDS)


def update_steps(steps: int):
    if PAB_MANAGER is not None:
        PAB_MANAGER.config.steps = steps


def _attn_entry(kind):
    def decide(timestep: int, count: int):
        return getattr(PAB_MANAGER, "if_broadcast_" + kind)(timestep, count) if enable_pab() else (False, count)

    decide.__name__ = "if_broadcast_" + kind
    decide.__doc__ = f"pab_mgr.py module-level if_broadcast_{kind}: (broadcast?, next counter)."
    return decide


if_broadcast_cross = _attn_entry("cross")
if_broadcast_temporal = _attn_entry("temporal")
if_broadcast_spatial = _attn_entry("spatial")


def if_broadcast_mlp(timestep: int, count: int, block_idx: int, all_timesteps, is_temporal=False):
    if not enable_pab():
        return False, count, False, None
    return PAB_MANAGER.if_skip_mlp(timestep, count, block_idx, all_timesteps, is_temporal)


def save_mlp_output(timestep: int, block_idx: int, ff_output, is_temporal=False):
    return PAB_MANAGER.save_skip_output(timestep, block_idx, ff_output, is_temporal)


def get_mlp_output(skip_range, timestep, block_idx: int, is_temporal=False):
    return PAB_MANAGER.get_mlp_output(skip_range, timestep, block_idx, is_temporal)
