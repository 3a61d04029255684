"""core/pipeline/pipeline.py mirror: the base class of the pipelines and their output record.

The reference derives ``VideoSysPipeline`` from diffusers' ``DiffusionPipeline`` (third-party; hub download, component registry,
``.to()`` of nn.Modules).  None of that is on the MI355X path — weights are read from local safetensors files into flat HBM
tables — so the base here holds only what the reference class itself adds (pipeline.py:10-30): ``generate`` as the entry point,
``__call__`` forwarding to it, ``set_eval_and_device``; plus the staged cpu_offload and seeding shared by the three pipelines."""
from __future__ import annotations

from dataclasses import dataclass, fields

import torch

from .utils import StagedOffloadMixin, ctor_kwargs, read_component


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

    @staticmethod
    def _resolve_device(device, who: str):
        """The reference's default is ``torch.device("cuda")`` (no index): that means the process's current HIP device."""
        if device is None or (torch.device(device).type == "cuda" and torch.device(device).index is None):
            if not torch.cuda.is_available():
                raise RuntimeError(f"{who} needs a HIP device (videosys_amd has no CPU execution path)")
            return torch.device("cuda", torch.cuda.current_device())
        return torch.device(device)

    @staticmethod
    def _check_dtype(dtype):
        """The reference's ``dtype`` keyword picks the torch dtype of its modules (bf16 for Open-Sora and CogVideoX-5b, fp16 for
        Latte and CogVideoX-2b).  The kernels here hold weights and activations in bf16 with fp32 accumulation and fp32 islands
        whatever is asked; a 16-bit request is accepted (same storage width), anything else is refused."""
        if dtype not in (torch.bfloat16, torch.float16):
            raise NotImplementedError(f"dtype {dtype}: videosys_amd computes in bf16 (fp32 accumulation and islands)")
        return torch.bfloat16

    @staticmethod
    def _check_scheduler(scheduler, needs: str, what: str):
        """A scheduler handed to the constructor must be this build's mirror of the reference class: the step itself is a fused
        kernel driven by the scheduler's per-step coefficients, a foreign ``step()`` cannot be called on HBM buffers."""
        if not callable(getattr(scheduler, needs, None)):
            raise TypeError(f"scheduler: expected {what} (it has no `{needs}`): got {type(scheduler).__name__}")
        return scheduler

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


# ---------------------------------------------------------------------------------------------------------------------------
# Components handed to a pipeline constructor.  The reference pipelines take ready-made ``text_encoder`` / ``tokenizer`` /
# ``vae`` / ``transformer`` / ``scheduler`` objects (torch modules) in place of the ``from_pretrained`` defaults
# (pipeline_open_sora.py:194-228, pipeline_latte.py:192-236, pipeline_cogvideox.py:124-160).  A caller of this build may pass
# either this build's own objects, or a torch module holding the reference's weights: its ``state_dict()`` / ``config`` are read
# once and the MI355X object is built from them (the module itself never runs).
# ---------------------------------------------------------------------------------------------------------------------------
def module_state(obj):
    """(config dict, state dict) of a torch-module-like component: ``state_dict()`` + ``.config`` (a dict, a namespace, or an HF
    config with ``to_dict()``)."""
    cfg = getattr(obj, "config", None)
    if cfg is None:
        cfg = {}
    elif hasattr(cfg, "to_dict"):
        cfg = cfg.to_dict()
    elif not isinstance(cfg, dict):
        cfg = {k: v for k, v in vars(cfg).items() if not k.startswith("_")}
    return dict(cfg), obj.state_dict()


def is_foreign_module(obj, own_types=()) -> bool:
    return obj is not None and not isinstance(obj, own_types) and callable(getattr(obj, "state_dict", None))


def t5_encoder_from(cfg: dict, sd, device):
    """A t5.T5Encoder with the geometry of an HF T5 ``config.json`` and the weights of its state dict."""
    from .t5 import T5Encoder

    enc = T5Encoder(**ctor_kwargs(T5Encoder.__init__, cfg), device=device)
    enc.load_state_dict(sd)
    return enc


def build_text_encoder(spec, tokenizer=None, *, device, caption_channels: int = 4096, max_length: int = 300,
                       use_attention_mask: bool = True, tokenizer_path=None):
    """The ``text_encoder`` callable of a pipeline (prompts -> (embeddings [B, 1, L, d], mask [B, L]), t5.T5TextEncoder) from
    whatever the constructor was given:
      * None                      -> None (generate() then expects ``prompt_embeds``)
      * a callable                -> itself (a T5TextEncoder, or any function with that contract)
      * ``"synthetic:<seed>"``    -> random T5 v1.1 weights of the geometry ``caption_channels`` implies + a byte tokenizer (offline)
      * a LOCAL directory         -> the HF checkpoint in it (config.json + *.safetensors); tokenizer from ``tokenizer`` or, when that
                                     is None, ``AutoTokenizer.from_pretrained(tokenizer_path or the directory)``
      * a torch module (HF ``T5EncoderModel``) -> its weights on the MI355X encoder, with ``tokenizer``
    A hub id cannot be fetched on an offline box: None."""
    import os

    from .t5 import ByteTokenizer, T5Encoder, T5TextEncoder

    if spec is None:
        return None
    if isinstance(spec, str):
        if spec.startswith("synthetic:"):
            d = caption_channels
            geo = dict(d_model=4096, d_ff=10240, num_layers=24, num_heads=64) if d == 4096 else \
                dict(d_model=d, d_ff=2 * d, num_layers=2, num_heads=max(d // 64, 2))
            enc = T5Encoder(device=device, **geo).init_random_(int(spec.split(":", 1)[1]))
            return T5TextEncoder(enc, tokenizer or ByteTokenizer(enc.config.vocab_size), max_length=max_length,
                                 use_attention_mask=use_attention_mask)
        cfg, sd = read_component(spec)
        if sd is None or not cfg:
            return None
        if tokenizer is None:
            from transformers import AutoTokenizer

            tok_dir = tokenizer_path if (tokenizer_path and os.path.isdir(tokenizer_path)) else spec
            tokenizer = AutoTokenizer.from_pretrained(tok_dir)
        return T5TextEncoder(t5_encoder_from(cfg, sd, device), tokenizer, max_length=max_length, use_attention_mask=use_attention_mask)
    if is_foreign_module(spec):
        if tokenizer is None:
            raise ValueError("text_encoder given as a module: pass its tokenizer too")
        cfg, sd = module_state(spec)
        return T5TextEncoder(t5_encoder_from(cfg, sd, device), tokenizer, max_length=max_length, use_attention_mask=use_attention_mask)
    if callable(spec):
        if tokenizer is not None and hasattr(spec, "tokenizer"):
            spec.tokenizer = tokenizer
        return spec
    raise TypeError(f"text_encoder: expected a callable, a torch module, a directory or 'synthetic:<seed>', got {type(spec).__name__}")
