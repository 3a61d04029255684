"""TEST INFRASTRUCTURE ONLY — loader for the *real* reference STDiT3 on CPU.

Imports the reference's own Python (``/root/reference/videosys``) with its
``videosys/__init__.py`` bypassed and the un-vendored third-party modules
stubbed (SURVEY.md §8c recipe).  It exists ONLY in the build container: the
GPU box has no ``/root/reference``.  It is used by ``oracle/make_golden.py``
to mint the fixtures under ``tests/golden/`` and by the CPU tests that check
the restatement (``oracle/stdit3_oracle.py``) against the real reference when
the reference tree is present.

Nothing under ``videosys_amd/`` may import this module.

Stubbed third-party pieces (not in /root/reference, not installable here):
  * ``timm.models.vision_transformer.Mlp`` / ``timm.models.layers.DropPath``
    (unpinned dep; call sites open_sora_transformer_3d.py:18-19,130-133,
    embeddings.py:11,197-203): fc1 -> act -> fc2 with biases; DropPath(0) = Identity.
  * ``rotary_embedding_torch.RotaryEmbedding`` (unpinned dep; constructed at
    open_sora_transformer_3d.py:388-390, called attentions.py:76-78): published
    algorithm restated in ``_RotaryEmbedding`` below (freqs_for="lang",
    theta=10000, interleaved pairs, seq_dim=-2).
  * ``colossalai.cluster.process_group_mesh.ProcessGroupMesh``, ``diffusers``
    type names, ``imageio``, ``omegaconf``: no arithmetic, empty shells.
"""
from __future__ import annotations

import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("VSYS_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "videosys", "models"))


class _Mlp(nn.Module):
    """timm.models.vision_transformer.Mlp restated: fc1 -> act -> drop -> fc2 -> drop."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, bias=True, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _RotaryEmbedding(nn.Module):
    """rotary_embedding_torch.RotaryEmbedding(dim) restated (lang freqs, theta 1e4)."""

    def __init__(self, dim, theta=10000):
        super().__init__()
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: (dim // 2)].float() / dim))
        self.freqs = nn.Parameter(freqs, requires_grad=False)

    def rotate_queries_or_keys(self, t, seq_dim=-2):
        seq_len = t.shape[seq_dim]
        seq = torch.arange(seq_len, device=t.device, dtype=t.dtype)
        freqs = torch.einsum("..., f -> ... f", seq.type(self.freqs.dtype), self.freqs)
        freqs = freqs.repeat_interleave(2, dim=-1)  # '... n -> ... (n r)', r=2
        dtype = t.dtype
        x = t.reshape(*t.shape[:-1], -1, 2)
        x1, x2 = x.unbind(dim=-1)
        rot = torch.stack((-x2, x1), dim=-1).reshape(t.shape)
        out = (t * freqs.cos()) + (rot * freqs.sin())
        return out.type(dtype)


class _SPStub:
    sp_size = 1
    cp_size = 1
    sp_group = None
    cp_group = None
    dp_rank = 0


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_INSTALLED = False


def install_stubs():
    global _INSTALLED
    if _INSTALLED:
        return
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    import transformers  # noqa: F401  (must be imported before the fake package is installed)

    # the repo ships an alias package of the same name whose submodule aliases point at videosys_amd (videosys/__init__.py): if a
    # test imported it earlier in this process, drop every entry so that "videosys.*" below can only come from the reference tree
    for k in [k for k in sys.modules if k == "videosys" or k.startswith("videosys.")]:
        del sys.modules[k]
    pkg = types.ModuleType("videosys")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "videosys")]
    sys.modules["videosys"] = pkg

    if "timm" not in sys.modules:
        _mod("timm")
        _mod("timm.models")
        _mod("timm.models.layers", DropPath=lambda *a, **k: nn.Identity())
        _mod("timm.models.vision_transformer", Mlp=_Mlp)
    if "colossalai" not in sys.modules:
        _mod("colossalai")
        _mod("colossalai.cluster")

        class ProcessGroupMesh:  # no arithmetic
            def __init__(self, *a, **k):
                pass

        _mod("colossalai.cluster.process_group_mesh", ProcessGroupMesh=ProcessGroupMesh)
    if "diffusers" not in sys.modules:
        _mod("diffusers")
        _mod("diffusers.models")
        _mod("diffusers.models.attention", Attention=object)
        _mod("diffusers.models.attention_processor", AttnProcessor=object)
    if "rotary_embedding_torch" not in sys.modules:
        _mod("rotary_embedding_torch", RotaryEmbedding=_RotaryEmbedding)
    if "imageio" not in sys.modules:
        _mod("imageio")
    if "omegaconf" not in sys.modules:
        _mod("omegaconf", OmegaConf=object, DictConfig=dict, ListConfig=list)
    _INSTALLED = True


def load_reference_modules():
    """Returns the reference modules used on the hot path (imported from /root/reference)."""
    install_stubs()
    import importlib

    names = {
        "stdit3": "videosys.models.transformers.open_sora_transformer_3d",
        "attentions": "videosys.models.modules.attentions",
        "normalization": "videosys.models.modules.normalization",
        "embeddings": "videosys.models.modules.embeddings",
        "rflow": "videosys.schedulers.scheduling_rflow_open_sora",
        "pab_mgr": "videosys.core.pab.pab_mgr",
        "comm": "videosys.core.distributed.comm",
    }
    mods = {k: importlib.import_module(v) for k, v in names.items()}
    for k, m in mods.items():   # the checker must be the reference's file, never this repo's module of the same dotted name
        assert os.path.abspath(m.__file__).startswith(os.path.abspath(REFERENCE_ROOT) + os.sep), (k, m.__file__)
    return mods


def build_reference_stdit3(cfg_kwargs: dict, state_dict=None, dtype=torch.float32):
    """Instantiate the reference STDiT3 on CPU (sp=cp=1) and optionally load weights."""
    mods = load_reference_modules()
    m = mods["stdit3"]
    model = m.STDiT3(m.STDiT3Config(**cfg_kwargs))
    model.parallel_manager = _SPStub()
    for blk in list(model.spatial_blocks) + list(model.temporal_blocks):
        blk.parallel_manager = _SPStub()
        blk.grad_checkpointing = False  # pure wrapper in no-grad inference (recompute.py:141-153)
    if state_dict is not None:
        missing, unexpected = model.load_state_dict(state_dict, strict=False)
        assert not unexpected, unexpected
        assert all("pos_embed" in k or "inv_freq" in k for k in missing), missing
    return model.to(dtype).eval()


# ---------------------------------------------------------------------------------------------------- Latte
def build_reference_latte(cfg_kwargs: dict, state_dict=None, dtype=torch.float32):
    """Instantiate the reference LatteT2V (videosys/models/transformers/latte_transformer_3d.py:893-1470) on CPU with
    the restated diffusers leaves of oracle/diffusers_stub.py, sp = cp = 1."""
    install_stubs()
    from oracle import diffusers_stub

    diffusers_stub.install()
    import importlib

    m = importlib.import_module("videosys.models.transformers.latte_transformer_3d")
    model = m.LatteT2V(**cfg_kwargs)
    model.parallel_manager = _SPStub()
    for blk in list(model.transformer_blocks) + list(model.temporal_transformer_blocks):
        blk.parallel_manager = _SPStub()
    if state_dict is not None:
        missing, unexpected = model.load_state_dict(state_dict, strict=False)
        assert not unexpected, unexpected
        assert all("pos_embed.pos_embed" in k or "temp_pos_embed" in k for k in missing), missing
    return model.to(dtype).eval()


# ---------------------------------------------------------------------------------------------------- CogVideoX
def build_reference_cogvideox(cfg_kwargs: dict, state_dict=None, dtype=torch.float32):
    """Instantiate the reference CogVideoXTransformer3DModel (cogvideox_transformer_3d.py:315-589) on CPU over the
    restated diffusers leaves, sp = cp = 1."""
    install_stubs()
    from oracle import diffusers_stub

    diffusers_stub.install()
    import importlib

    m = importlib.import_module("videosys.models.transformers.cogvideox_transformer_3d")
    model = m.CogVideoXTransformer3DModel(**cfg_kwargs)
    model.parallel_manager = _SPStub()
    for blk in model.transformer_blocks:
        blk.attn1.parallel_manager = _SPStub()
    if state_dict is not None:
        missing, unexpected = model.load_state_dict(state_dict, strict=False)
        assert not unexpected, unexpected
        assert all("pos_embedding" in k for k in missing), missing
    return model.to(dtype).eval()


def load_reference_cogvideox_scheduler(**kwargs):
    """The in-tree CogVideoXDDIMScheduler (schedulers/scheduling_ddim_cogvideox.py:118-395)."""
    install_stubs()
    from oracle import diffusers_stub

    diffusers_stub.install()
    import importlib

    m = importlib.import_module("videosys.schedulers.scheduling_ddim_cogvideox")
    s = m.CogVideoXDDIMScheduler(**kwargs)
    return s


def load_reference_cogvideox_dpm_scheduler(**kwargs):
    """The in-tree CogVideoXDPMScheduler (schedulers/scheduling_dpm_cogvideox.py:119-483; its ``randn_tensor`` is the restated
    diffusers leaf of oracle/diffusers_stub.py)."""
    install_stubs()
    from oracle import diffusers_stub

    diffusers_stub.install()
    import importlib

    m = importlib.import_module("videosys.schedulers.scheduling_dpm_cogvideox")
    return m.CogVideoXDPMScheduler(**kwargs)


# ---------------------------------------------------------------------------------------------------- Open-Sora VAE
def build_reference_opensora_vae(state_dict=None, dtype=torch.float32, micro_frame_size=17, micro_batch_size=4):
    """Instantiate the reference VideoAutoencoderPipeline (autoencoder_kl_open_sora.py:620-735) on CPU: the temporal VAE
    (VAE_Temporal_SD, reference code) over the restated diffusers AutoencoderKL decoder of oracle/diffusers_stub.py, with the
    OpenSoraVAE_V1_2 normalisation constants (:738-761).  Random init unless ``state_dict`` is given."""
    install_stubs()
    from oracle import diffusers_stub

    diffusers_stub.install()
    import importlib

    m = importlib.import_module("videosys.models.autoencoders.autoencoder_kl_open_sora")
    cfg = m.VideoAutoencoderPipelineConfig(
        vae_2d=dict(type="VideoAutoencoderKL", from_pretrained="stub", subfolder="vae", micro_batch_size=micro_batch_size),
        vae_temporal=dict(type="VAE_Temporal_SD", from_pretrained=None), freeze_vae_2d=False, cal_loss=False,
        micro_frame_size=micro_frame_size, shift=(-0.10, 0.34, 0.27, 0.98), scale=(3.85, 2.32, 2.33, 3.06))
    model = m.VideoAutoencoderPipeline(cfg)
    if state_dict is not None:
        missing, unexpected = model.load_state_dict(state_dict, strict=False)
        assert not unexpected, unexpected
        assert all("encoder" in k or ".quant_conv" in k or k in ("scale", "shift") for k in missing), missing
    return model.to(dtype).eval()


# ---------------------------------------------------------------------------------------------------- CogVideoX VAE
def build_reference_cogvideox_vae(state_dict=None, dtype=torch.float32, **cfg):
    """Instantiate the reference AutoencoderKLCogVideoX (autoencoder_kl_cogvideox.py:872-1257; in-tree code: causal convs with
    conv_cache, spatial norm, 3-D upsampling, frame batching, tiled decode) on CPU.  Only import-level diffusers names are
    stubbed (mixins, DecoderOutput, get_activation)."""
    install_stubs()
    from oracle import diffusers_stub

    diffusers_stub.install()
    import importlib

    m = importlib.import_module("videosys.models.autoencoders.autoencoder_kl_cogvideox")
    model = m.AutoencoderKLCogVideoX(**cfg)
    if state_dict is not None:
        missing, unexpected = model.load_state_dict(state_dict, strict=False)
        assert not unexpected, unexpected
        assert all(k.startswith("encoder.") for k in missing), missing
    return model.to(dtype).eval()
