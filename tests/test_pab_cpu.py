"""CPU: videosys_amd.pab against the reference's own videosys/core/pab/pab_mgr.py (dependency-free, imported from /root/reference
when present) on randomised schedules: every attention / cross / MLP decision, counter, skip range and the MLP output cache
(store, fetch, delete-at-window-end, missing-entry error) must agree call for call."""
import importlib.util
import os
import random

import pytest

REF = "/root/reference/videosys/core/pab/pab_mgr.py"


def _ref_module():
    if not os.path.exists(REF):
        pytest.skip("reference tree not present on this box")
    spec = importlib.util.spec_from_file_location("ref_pab_mgr", REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _cfg_kwargs(rng):
    ts = sorted(rng.sample(range(0, 1000, 7), 24), reverse=True)           # a denoising schedule (descending ints)
    keys = rng.sample(ts[:-4], 4)
    mk = lambda: {k: {"block": rng.sample(range(6), rng.randint(1, 5)), "skip_count": rng.randint(1, 3)} for k in keys}
    kw = dict(cross_broadcast=rng.random() < 0.8, cross_threshold=[rng.randint(0, 300), rng.randint(500, 1000)], cross_range=rng.randint(2, 7),
              spatial_broadcast=rng.random() < 0.8, spatial_threshold=[rng.randint(0, 300), rng.randint(500, 1000)], spatial_range=rng.randint(2, 5),
              temporal_broadcast=True, temporal_threshold=[rng.randint(0, 300), rng.randint(500, 1000)], temporal_range=rng.randint(2, 5),
              mlp_broadcast=rng.random() < 0.8, mlp_spatial_broadcast_config=mk(), mlp_temporal_broadcast_config=mk())
    return ts, kw


@pytest.mark.parametrize("seed", range(8))
def test_pab_decisions_match_reference_module(seed):
    ref = _ref_module()
    from videosys_amd import pab as mine

    rng = random.Random(seed)
    ts, kw = _cfg_kwargs(rng)
    ref.set_pab_manager(ref.PABConfig(**kw))
    mine.set_pab_manager(mine.PABConfig(**kw))
    try:
        ref.update_steps(len(ts))
        mine.update_steps(len(ts))
        assert ref.enable_pab() == mine.enable_pab()
        nblk = 6
        cnt_r = {(k, b): 0 for k in ("s", "t", "c", "ms", "mt") for b in range(nblk)}
        cnt_m = dict(cnt_r)
        for t in ts:
            for b in range(nblk):
                for kind, name in (("s", "if_broadcast_spatial"), ("t", "if_broadcast_temporal"), ("c", "if_broadcast_cross")):
                    fr, cnt_r[(kind, b)] = getattr(ref, name)(t, cnt_r[(kind, b)])
                    fm, cnt_m[(kind, b)] = getattr(mine, name)(t, cnt_m[(kind, b)])
                    assert (fr, cnt_r[(kind, b)]) == (fm, cnt_m[(kind, b)]), (name, t, b)
                for kind, temporal in (("ms", False), ("mt", True)):
                    r = ref.if_broadcast_mlp(t, cnt_r[(kind, b)], b, ts, is_temporal=temporal)
                    m = mine.if_broadcast_mlp(t, cnt_m[(kind, b)], b, ts, is_temporal=temporal)
                    assert tuple(r) == tuple(m), (t, b, temporal, r, m)
                    flag, c, nxt, skip = r
                    if c is not None:
                        cnt_r[(kind, b)] = cnt_m[(kind, b)] = c
                    if flag:
                        try:
                            orf = ref.get_mlp_output(skip, timestep=t, block_idx=b, is_temporal=temporal)
                        except ValueError:
                            with pytest.raises(ValueError):
                                mine.get_mlp_output(skip, timestep=t, block_idx=b, is_temporal=temporal)
                        else:
                            assert mine.get_mlp_output(skip, timestep=t, block_idx=b, is_temporal=temporal) == orf
                    elif nxt:
                        ref.save_mlp_output(timestep=t, block_idx=b, ff_output=("ff", t, b, temporal), is_temporal=temporal)
                        mine.save_mlp_output(timestep=t, block_idx=b, ff_output=("ff", t, b, temporal), is_temporal=temporal)
            rc, mc = ref.PAB_MANAGER.config, mine.PAB_MANAGER.config
            assert set(rc.mlp_spatial_outputs) == set(mc.mlp_spatial_outputs) and set(rc.mlp_temporal_outputs) == set(mc.mlp_temporal_outputs)
    finally:
        mine.set_pab_manager(None)
        ref.PAB_MANAGER = None


def test_pab_disabled_paths():
    from videosys_amd import pab as mine

    mine.set_pab_manager(None)
    assert not mine.enable_pab()
    assert mine.if_broadcast_cross(500, 3) == (False, 3)
    assert mine.if_broadcast_mlp(500, 3, 0, [500, 400]) == (False, 3, False, None)
