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

# ---------------------------------------------------------------------------------------------------------------------------
# The reference's module paths.  Its examples, eval scripts and tests import from submodules as well
# (``from videosys.core.pab.pab_mgr import PABConfig, set_pab_manager``, ``from videosys.utils.utils import set_seed``,
# ``from videosys.schedulers.scheduling_rflow_open_sora import RFLOW`` ...): every inference-side path of the reference tree
# resolves to the videosys_amd module that holds the same names.  An alias IS the target module object (one set of globals — the
# PAB manager installed through one path is the one read through the other).  Training-side modules (videosys.training,
# core.dcp, utils.training) are outside the MI355X hot path and are not aliased.
# ---------------------------------------------------------------------------------------------------------------------------
_MODULES = {
    "utils.utils": "utils",                                   # set_seed, str_to_dtype, batch_func, save_video, all_exists, requires_grad
    "utils.test": "utils",                                    # empty_cache
    "utils.logging": "utils",                                 # init_logger
    "core.pab.pab_mgr": "pab",                                # PABConfig, PABManager, set_pab_manager, enable_pab, update_steps, if_broadcast_*
    "core.distributed.parallel_mgr": "dsp",                   # ParallelManager, initialize
    "core.distributed.comm": "comm",                          # split_sequence, gather_sequence, all_to_all_comm, all_to_all_with_pad, set_pad, get_pad
    "core.pipeline.pipeline": "pipeline",                     # VideoSysPipeline, VideoSysPipelineOutput
    "core.engine.engine": "engine",                           # VideoSysEngine
    "core.engine.mp_utils": "engine",                         # ResultHandler, WorkerMonitor, get_open_port
    "schedulers.scheduling_rflow_open_sora": "rflow",         # RFLOW, timestep_transform
    "schedulers.scheduling_ddim_cogvideox": "pipeline_cogvideox",   # CogVideoXDDIMScheduler
    "schedulers.scheduling_dpm_cogvideox": "pipeline_cogvideox",    # CogVideoXDPMScheduler
    "models.modules.normalization": "modules",                # LlamaRMSNorm, get_rms_norm
    "models.transformers.open_sora_transformer_3d": "stdit3",       # STDiT3, STDiT3Config, STDiT3_XL_2
    "models.transformers.latte_transformer_3d": "latte",            # LatteT2V
    "models.transformers.cogvideox_transformer_3d": "cogvideox",    # CogVideoXTransformer3DModel
    "models.autoencoders.autoencoder_kl_open_sora": "vae_open_sora",    # OpenSoraVAE_V1_2
    "models.autoencoders.autoencoder_kl_cogvideox": "vae_cogvideox",    # AutoencoderKLCogVideoX
    "pipelines.open_sora": "pipeline_open_sora",              # OpenSoraConfig, OpenSoraPABConfig, OpenSoraPipeline
    "pipelines.open_sora.pipeline_open_sora": "pipeline_open_sora",
    "pipelines.latte": "pipeline_latte",
    "pipelines.latte.pipeline_latte": "pipeline_latte",
    "pipelines.cogvideox": "pipeline_cogvideox",
    "pipelines.cogvideox.pipeline_cogvideox": "pipeline_cogvideox",
}


def _install_aliases():
    import importlib
    import sys
    import types

    def package(name):
        mod = sys.modules.get(name)
        if mod is None:
            mod = types.ModuleType(name, "namespace of the reference's module tree (videosys/__init__.py alias table)")
            mod.__path__ = []
            sys.modules[name] = mod
            parent, _, leaf = name.rpartition(".")
            setattr(package(parent) if parent != __name__ else sys.modules[__name__], leaf, mod)
        return mod

    for alias, target in sorted(_MODULES.items(), key=lambda kv: (-kv[0].count("."), kv[0])):   # deepest first
        mod = importlib.import_module("videosys_amd." + target)
        full = f"{__name__}.{alias}"
        parent, _, leaf = full.rpartition(".")
        holder = package(parent) if parent != __name__ else sys.modules[__name__]
        existing = sys.modules.get(full)
        if isinstance(existing, types.ModuleType) and getattr(existing, "__path__", None) == []:
            # the alias is also a package of deeper aliases (pipelines.open_sora -> pipeline_open_sora, with
            # pipelines.open_sora.data_process below it): keep the namespace, give it the target's public names
            for k, v in vars(mod).items():
                if not k.startswith("_"):
                    setattr(existing, k, v)
            continue
        sys.modules[full] = mod
        setattr(holder, leaf, mod)
    # data_process.py of the Open-Sora pipeline: resolution / frame-count tables and the reference readers live in two modules here
    from videosys_amd import open_sora_condition, open_sora_geometry

    dp = types.ModuleType(f"{__name__}.pipelines.open_sora.data_process", "pipelines/open_sora/data_process.py names")
    for src in (open_sora_geometry, open_sora_condition):
        for k, v in vars(src).items():
            if not k.startswith("_"):
                setattr(dp, k, v)
    sys.modules[dp.__name__] = dp
    package(f"{__name__}.pipelines.open_sora").data_process = dp


_install_aliases()

_OUT_OF_SCOPE = {"OpenSoraPlanPipeline", "OpenSoraPlanConfig", "OpenSoraPlanV110PABConfig", "OpenSoraPlanV120PABConfig",
                 "VchitectXLPipeline", "VchitectConfig", "VchitectPABConfig"}


def __getattr__(name):
    if name in _OUT_OF_SCOPE:
        raise ImportError(f"videosys.{name}: the Open-Sora-Plan and Vchitect pipelines are outside the MI355X hot path of this "
                          "build (DESIGN.md §7); Open-Sora, Latte and CogVideoX are available")
    raise AttributeError(f"module 'videosys' has no attribute {name!r}")
