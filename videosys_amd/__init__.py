"""videosys_amd — MI355X (gfx950) native denoise-step engine behind the VideoSys Open-Sora plugin surface.

Public names mirror ``videosys/__init__.py`` of the reference for the paths this build covers: ``initialize``,
``VideoSysEngine``, ``OpenSoraConfig`` / ``OpenSoraPABConfig`` / ``OpenSoraPipeline`` and ``LatteConfig`` /
``LattePABConfig`` / ``LattePipeline``, ``CogVideoXConfig`` / ``CogVideoXPABConfig`` / ``CogVideoXPipeline``.
Importing the package does not need a GPU; running any op does (no CPU fallback).
"""
from .dsp import initialize  # noqa: F401
from .engine import VideoSysEngine  # noqa: F401
from .pipeline_cogvideox import CogVideoXConfig, CogVideoXPABConfig, CogVideoXPipeline  # noqa: F401
from .pipeline_latte import LatteConfig, LattePABConfig, LattePipeline  # noqa: F401
from .pipeline_open_sora import OpenSoraConfig, OpenSoraPABConfig, OpenSoraPipeline  # noqa: F401

__all__ = ["initialize", "VideoSysEngine", "OpenSoraPipeline", "OpenSoraConfig", "OpenSoraPABConfig", "LattePipeline",
           "LatteConfig", "LattePABConfig", "CogVideoXPipeline", "CogVideoXConfig", "CogVideoXPABConfig"]
