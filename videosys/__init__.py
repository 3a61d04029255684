"""``videosys`` — drop-in alias of the MI355X-native package ``videosys_amd``.

Programs written against the reference import ``from videosys import VideoSysEngine, OpenSoraConfig, ...``
(videosys/__init__.py:1-22 there); this module re-exports the same names for every pipeline family this build covers, so
e.g. the reference's examples/inference/open_sora/sample.py runs unchanged.  Families outside the MI355X hot path
(Open-Sora-Plan, Vchitect: DESIGN.md §7) raise an ImportError that says so instead of an AttributeError.
"""
from videosys_amd import (CogVideoXConfig, CogVideoXPABConfig, CogVideoXPipeline, LatteConfig, LattePABConfig,  # noqa: F401
                          LattePipeline, OpenSoraConfig, OpenSoraPABConfig, OpenSoraPipeline, VideoSysEngine, initialize)

__all__ = ["initialize", "VideoSysEngine", "LattePipeline", "LatteConfig", "LattePABConfig", "OpenSoraPipeline", "OpenSoraConfig",
           "OpenSoraPABConfig", "CogVideoXPipeline", "CogVideoXConfig", "CogVideoXPABConfig"]

_OUT_OF_SCOPE = {"OpenSoraPlanPipeline", "OpenSoraPlanConfig", "OpenSoraPlanV110PABConfig", "OpenSoraPlanV120PABConfig",
                 "VchitectXLPipeline", "VchitectConfig", "VchitectPABConfig"}


def __getattr__(name):
    if name in _OUT_OF_SCOPE:
        raise ImportError(f"videosys.{name}: the Open-Sora-Plan and Vchitect pipelines are outside the MI355X hot path of this "
                          "build (DESIGN.md §7); Open-Sora, Latte and CogVideoX are available")
    raise AttributeError(f"module 'videosys' has no attribute {name!r}")
