"""core/pipeline/pipeline.py mirror: the base class of the pipelines and their output record.

The reference derives ``VideoSysPipeline`` from diffusers' ``DiffusionPipeline`` (third-party; hub download, component registry,
``.to()`` of nn.Modules).  None of that is on the MI355X path — weights are read from local safetensors files into flat HBM
tables — so the base here holds only what the reference class itself adds (pipeline.py:10-30): ``generate`` as the entry point,
``__call__`` forwarding to it, ``set_eval_and_device``; plus the staged cpu_offload and seeding shared by the three pipelines."""
from __future__ import annotations

from dataclasses import dataclass, fields

import torch

from .utils import StagedOffloadMixin


class VideoSysPipeline(StagedOffloadMixin):
    @staticmethod
    def set_eval_and_device(device, *modules):
        """pipeline.py:14-19: ``eval()`` and ``to(device)`` on every module that has them.  The model objects of this build are
        built on their device and have no train mode; torch modules a caller attaches (a text encoder, a VAE) are moved."""
        for m in modules:
            if hasattr(m, "eval"):
                m.eval()
            if hasattr(m, "to"):
                m.to(device)

    def generate(self, *args, **kwargs):   # pipeline.py:21-23 (abstract there)
        raise NotImplementedError(f"{type(self).__name__} does not define generate()")

    def __call__(self, *args, **kwargs):
        """pipeline.py:25-31: the diffusers calling convention, forwarded to ``generate``."""
        return self.generate(*args, **kwargs)


@dataclass
class VideoSysPipelineOutput:
    """core/pipeline/pipeline.py:50-52.  The reference's record is a diffusers ``BaseOutput`` (third-party): besides ``.video`` it
    answers ``out["video"]``, ``out[0]`` and ``to_tuple()``; so does this one."""

    video: torch.Tensor

    def to_tuple(self):
        return tuple(getattr(self, f.name) for f in fields(self))

    def __getitem__(self, k):
        return getattr(self, k) if isinstance(k, str) else self.to_tuple()[k]

    def keys(self):
        return [f.name for f in fields(self)]
