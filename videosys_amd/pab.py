"""Pyramid Attention Broadcast policy — host mirror of videosys/core/pab/pab_mgr.py (reference :6-232).

Same names, arguments and decisions as the reference module so configs drop in unchanged.  The cache itself lives in
pre-allocated HBM slabs owned by the transformer blocks (``videosys_amd/stdit3.py``); this module only decides.
The timestep handed to the decision functions is a Python int that the sampler already has on the host
(``all_timesteps``), so no device->host sync happens per block (the reference does ``int(timestep[0])`` per block,
open_sora_transformer_3d.py:188,190,232,244).
"""
from __future__ import annotations

import logging

PAB_MANAGER = None


class PABConfig:
    """pab_mgr.py:6-41 — identical kwargs/defaults."""

    def __init__(
        self,
        cross_broadcast: bool = False,
        cross_threshold: list = None,
        cross_range: int = None,
        spatial_broadcast: bool = False,
        spatial_threshold: list = None,
        spatial_range: int = None,
        temporal_broadcast: bool = False,
        temporal_threshold: list = None,
        temporal_range: int = None,
        mlp_broadcast: bool = False,
        mlp_spatial_broadcast_config: dict = None,
        mlp_temporal_broadcast_config: dict = None,
    ):
        self.steps = None
        self.cross_broadcast = cross_broadcast
        self.cross_threshold = cross_threshold
        self.cross_range = cross_range
        self.spatial_broadcast = spatial_broadcast
        self.spatial_threshold = spatial_threshold
        self.spatial_range = spatial_range
        self.temporal_broadcast = temporal_broadcast
        self.temporal_threshold = temporal_threshold
        self.temporal_range = temporal_range
        self.mlp_broadcast = mlp_broadcast
        self.mlp_spatial_broadcast_config = mlp_spatial_broadcast_config
        self.mlp_temporal_broadcast_config = mlp_temporal_broadcast_config
        self.mlp_temporal_outputs = {}
        self.mlp_spatial_outputs = {}


class PABManager:
    """pab_mgr.py:43-181 (attention/cross decisions; the MLP-broadcast branch is kept for API parity, see
    if_skip_mlp)."""

    def __init__(self, config: PABConfig):
        self.config: PABConfig = config
        logging.info(
            "Init Pyramid Attention Broadcast. spatial: %s/%s/%s temporal: %s/%s/%s cross: %s/%s/%s mlp: %s",
            config.spatial_broadcast, config.spatial_range, config.spatial_threshold,
            config.temporal_broadcast, config.temporal_range, config.temporal_threshold,
            config.cross_broadcast, config.cross_range, config.cross_threshold, config.mlp_broadcast,
        )

    def _decide(self, enabled, rng, thr, timestep, count):
        flag = bool(enabled and (timestep is not None) and (count % rng != 0) and (thr[0] < timestep < thr[1]))
        return flag, (count + 1) % self.config.steps

    def if_broadcast_cross(self, timestep: int, count: int):
        c = self.config
        return self._decide(c.cross_broadcast, c.cross_range, c.cross_threshold, timestep, count)

    def if_broadcast_temporal(self, timestep: int, count: int):
        c = self.config
        return self._decide(c.temporal_broadcast, c.temporal_range, c.temporal_threshold, timestep, count)

    def if_broadcast_spatial(self, timestep: int, count: int):
        c = self.config
        return self._decide(c.spatial_broadcast, c.spatial_range, c.spatial_threshold, timestep, count)

    @staticmethod
    def _is_t_in_skip_config(all_timesteps, timestep, config):
        """pab_mgr.py:93-106."""
        is_t_in_skip_config = False
        skip_range = None
        for key in config:
            if key not in all_timesteps:
                continue
            index = all_timesteps.index(key)
            skip_range = all_timesteps[index : index + 1 + int(config[key]["skip_count"])]
            if timestep in skip_range:
                is_t_in_skip_config = True
                skip_range = [all_timesteps[index], all_timesteps[index + int(config[key]["skip_count"])]]
                break
        return is_t_in_skip_config, skip_range

    def if_skip_mlp(self, timestep: int, count: int, block_idx: int, all_timesteps, is_temporal=False):
        """pab_mgr.py:108-141.  Unlike the reference's STDiT3.forward (which never forwards ``all_timesteps`` to
        the blocks and therefore raises TypeError whenever mlp_broadcast=True — SURVEY.md §0.9), this build passes
        ``all_timesteps`` through, so the documented behaviour is reachable."""
        if not self.config.mlp_broadcast:
            return False, None, False, None
        cur_config = self.config.mlp_temporal_broadcast_config if is_temporal else self.config.mlp_spatial_broadcast_config
        is_t_in_skip_config, skip_range = self._is_t_in_skip_config(all_timesteps, timestep, cur_config)
        next_flag = False
        if (timestep is not None) and (timestep in cur_config) and (block_idx in cur_config[timestep]["block"]):
            flag = False
            next_flag = True
            count = count + 1
        elif (timestep is not None) and is_t_in_skip_config and (block_idx in cur_config[skip_range[0]]["block"]):
            flag = True
            count = 0
        else:
            flag = False
        return flag, count, next_flag, skip_range

    def save_skip_output(self, timestep, block_idx, ff_output, is_temporal=False):
        d = self.config.mlp_temporal_outputs if is_temporal else self.config.mlp_spatial_outputs
        d[(timestep, block_idx)] = ff_output

    def get_mlp_output(self, skip_range, timestep, block_idx, is_temporal=False):
        d = self.config.mlp_temporal_outputs if is_temporal else self.config.mlp_spatial_outputs
        skip_start_t = skip_range[0]
        skip_output = d.get((skip_start_t, block_idx), None)
        if skip_output is None:
            raise ValueError(
                f"No stored MLP output found | t {timestep} |[{skip_range[0]}, {skip_range[-1]}] | block {block_idx}"
            )
        if timestep == skip_range[-1]:
            del d[(skip_start_t, block_idx)]
        return skip_output


def set_pab_manager(config: PABConfig):
    global PAB_MANAGER
    PAB_MANAGER = PABManager(config) if config is not None else None


def enable_pab():
    if PAB_MANAGER is None:
        return False
    c = PAB_MANAGER.config
    return bool(c.cross_broadcast or c.spatial_broadcast or c.temporal_broadcast)


def update_steps(steps: int):
    if PAB_MANAGER is not None:
        PAB_MANAGER.config.steps = steps


def if_broadcast_cross(timestep: int, count: int):
    if not enable_pab():
        return False, count
    return PAB_MANAGER.if_broadcast_cross(timestep, count)


def if_broadcast_temporal(timestep: int, count: int):
    if not enable_pab():
        return False, count
    return PAB_MANAGER.if_broadcast_temporal(timestep, count)


def if_broadcast_spatial(timestep: int, count: int):
    if not enable_pab():
        return False, count
    return PAB_MANAGER.if_broadcast_spatial(timestep, count)


def if_broadcast_mlp(timestep: int, count: int, block_idx: int, all_timesteps, is_temporal=False):
    if not enable_pab():
        return False, count, False, None
    return PAB_MANAGER.if_skip_mlp(timestep, count, block_idx, all_timesteps, is_temporal)


def save_mlp_output(timestep: int, block_idx: int, ff_output, is_temporal=False):
    return PAB_MANAGER.save_skip_output(timestep, block_idx, ff_output, is_temporal)


def get_mlp_output(skip_range, timestep, block_idx: int, is_temporal=False):
    return PAB_MANAGER.get_mlp_output(skip_range, timestep, block_idx, is_temporal)
