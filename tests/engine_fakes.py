"""CPU-only stand-in pipeline for tests/test_engine_cpu.py: what VideoSysEngine needs of a pipeline plug-in is ``__init__(config)``,
``generate(...)`` and ``save_video``; this one sums a tensor over the gloo group so a missing or stale rank shows up in the value."""
import os

import torch
import torch.distributed as dist


class FakePipeline:
    def __init__(self, config):
        self.config = config
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        if config.fail_init_rank == self.rank:
            raise ValueError(f"constructor refused on rank {self.rank}")

    def generate(self, x, mode="ok", who=1):
        if mode == "raise" and self.rank == who:
            raise KeyError(f"bad prompt on rank {self.rank}")
        if mode == "die" and self.rank == who:
            os._exit(3)
        if mode == "raise_peers_in_collective":
            # rank `who` fails BEFORE the collective its peers enter: they block in all_reduce waiting for it
            if self.rank == who:
                raise KeyError(f"bad prompt on rank {self.rank}")
            t = torch.tensor([float(x)])
            dist.all_reduce(t)
            return float(t)
        if mode in ("raise", "die"):
            return None   # the other ranks do not enter a collective their peer will never reach
        if mode == "talk":
            print(f"rank {self.rank} says\ntwo lines", flush=True)
        t = torch.tensor([float(x) + self.rank])
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(t)
        return float(t)

    def save_video(self, video, path):
        return path


class FakeConfig:
    def __init__(self, num_gpus=2, fail_init_rank=None):
        self.num_gpus = num_gpus
        self.pipeline_cls = FakePipeline
        self.fail_init_rank = fail_init_rank
