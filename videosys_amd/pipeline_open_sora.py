"""Open-Sora pipeline plug-in — host mirror of videosys/pipelines/open_sora/pipeline_open_sora.py
(OpenSoraPABConfig :32-69, OpenSoraConfig :72-163, OpenSoraPipeline :166-659) for the denoising hot path.

``OpenSoraConfig`` / ``OpenSoraPABConfig`` take the same kwargs with the same defaults, expose ``pipeline_cls`` and
``num_gpus`` and drop into ``VideoSysEngine(config)`` unchanged; ``generate`` keeps the reference's call shape
(``resolution`` / ``aspect_ratio`` / ``num_frames="2s"`` through open_sora_geometry.py).  The T5 text encoder and the VAE
decoder load from local checkpoint directories (no hub access here); ``generate`` also accepts ``prompt_embeds`` /
``prompt_mask`` and returns latents when no VAE is attached, and raises a clear error if a raw prompt string arrives
without a text encoder.
No pretrained weights exist offline: ``transformer="synthetic:<seed>"`` (default when the HF name cannot be resolved
locally) builds seeded random weights of the real STDiT3-XL/2 geometry.
"""
from __future__ import annotations

import glob
import os
from typing import Callable, Optional

import torch

from . import dsp, pab
from .pab import PABConfig
from .rflow import RFLOW
from .stdit3 import STDiT3, STDiT3Config
from .pipeline import (VideoSysPipeline, VideoSysPipelineOutput, build_text_encoder, is_foreign_module,  # noqa: F401 (re-exported)
                       module_state)

os.environ.setdefault("TOKENIZERS_PARALLELISM", "true")   # pipeline_open_sora.py:24 (kept if the application set it)


class OpenSoraPABConfig(PABConfig):
    """pipeline_open_sora.py:32-69 — identical defaults, incl. mlp_broadcast=True with the three default windows (in the
    reference that default raises TypeError because STDiT3.forward drops ``all_timesteps``; here the schedule
    is handed down and the MLP broadcast runs).  mlp_broadcast=False = attention-only PAB, BASELINE config 3."""

    def __init__(
        self,
        spatial_broadcast: bool = True,
        spatial_threshold: list = [450, 930],
        spatial_range: int = 2,
        temporal_broadcast: bool = True,
        temporal_threshold: list = [450, 930],
        temporal_range: int = 4,
        cross_broadcast: bool = True,
        cross_threshold: list = [450, 930],
        cross_range: int = 6,
        mlp_broadcast: bool = True,
        mlp_spatial_broadcast_config: dict = None,
        mlp_temporal_broadcast_config: dict = None,
    ):
        default_mlp = {
            676: {"block": [0, 1, 2, 3, 4], "skip_count": 2},
            788: {"block": [0, 1, 2, 3, 4], "skip_count": 2},
            864: {"block": [0, 1, 2, 3, 4], "skip_count": 2},
        }
        super().__init__(
            spatial_broadcast=spatial_broadcast, spatial_threshold=spatial_threshold, spatial_range=spatial_range,
            temporal_broadcast=temporal_broadcast, temporal_threshold=temporal_threshold, temporal_range=temporal_range,
            cross_broadcast=cross_broadcast, cross_threshold=cross_threshold, cross_range=cross_range,
            mlp_broadcast=mlp_broadcast,
            mlp_spatial_broadcast_config=mlp_spatial_broadcast_config or dict(default_mlp),
            mlp_temporal_broadcast_config=mlp_temporal_broadcast_config or dict(default_mlp),
        )


class OpenSoraConfig:
    """pipeline_open_sora.py:72-163 — identical kwargs/defaults."""

    def __init__(
        self,
        transformer: str = "hpcai-tech/OpenSora-STDiT-v3",
        vae: str = "hpcai-tech/OpenSora-VAE-v1.2",
        text_encoder: str = "DeepFloyd/t5-v1_1-xxl",
        num_gpus: int = 1,
        num_sampling_steps: int = 30,
        cfg_scale: float = 7.0,
        cpu_offload: bool = False,
        tiling_size: int = 4,
        enable_flash_attn: bool = False,
        enable_pab: bool = False,
        pab_config: PABConfig = None,
        **extra,
    ):
        self.pipeline_cls = OpenSoraPipeline
        self.transformer = transformer
        self.vae = vae
        self.text_encoder = text_encoder
        self.num_gpus = num_gpus
        self.num_sampling_steps = num_sampling_steps
        self.cfg_scale = cfg_scale
        self.tiling_size = tiling_size
        self.cpu_offload = cpu_offload
        self.enable_flash_attn = enable_flash_attn
        self.enable_pab = enable_pab
        self.pab_config = pab_config if pab_config is not None else OpenSoraPABConfig()
        # extensions (not in the reference): geometry override for tests, e.g. transformer_config=dict(depth=2, ...)
        self.transformer_config = extra.pop("transformer_config", None)
        # with num_gpus > 1 the VAE decode is sharded over the ranks by output frame and gathered once (the reference decodes the
        # whole video on every rank: autoencoder_kl_open_sora.py:672-695 under engine.py:85-95); False = the reference's behaviour
        self.shard_vae_decode = bool(extra.pop("shard_vae_decode", True))
        if extra:
            raise TypeError(f"unexpected OpenSoraConfig kwargs: {sorted(extra)}")


def get_latent_size(num_frames: int, height: int, width: int):
    """OpenSoraVAE_V1_2.get_latent_size (autoencoder_kl_open_sora.py:706-717): spatial /8; time 17-frame micro
    batches compress x4 with a causal first frame: 64 frames -> 3*5 + 4 = 19."""
    micro = 17

    def tlat(n):
        return -(-n // 4) if n > 1 else 1  # ceil(n/4) per micro batch (time_downsample_factor 4, causal pad)

    if num_frames == 1:
        t = 1
    else:
        t = (num_frames // micro) * tlat(micro) + (tlat(num_frames % micro) if num_frames % micro else 0)
    return (t, height // 8, width // 8)


class OpenSoraPipeline(VideoSysPipeline):
    """The per-rank pipeline object the engine instantiates (engine.py:68-72) and whose ``generate`` it calls."""

    def __init__(self, config: OpenSoraConfig, text_encoder=None, tokenizer=None, vae=None, transformer=None, scheduler=None,
                 device=None, dtype: torch.dtype = torch.bfloat16, *, vae_decoder: Optional[Callable] = None):
        """pipeline_open_sora.py:194-250, same parameter order.  Components left at None are loaded as the reference loads them
        (``config.text_encoder`` / ``config.vae`` / ``config.transformer``: LOCAL checkpoint directories or ``"synthetic:<seed>"``);
        a component may be this build's object or a torch module holding the reference's weights (pipeline.module_state).
        ``dtype``: see VideoSysPipeline._check_dtype.  ``vae_decoder`` = ``vae``."""
        self._config = config
        self._dtype = self._check_dtype(dtype)
        self._device = self._resolve_device(device, "OpenSoraPipeline")
        if transformer is None:
            name = config.transformer   # a local checkpoint directory or "synthetic:<seed>"; a hub id cannot be fetched: seeded weights
            local = isinstance(name, str) and (name.startswith("synthetic:") or bool(glob.glob(os.path.join(name, "*.safetensors"))))
            if not local:
                self._hub_fallback(name, "transformer", "synthetic:1234")
            transformer = STDiT3.from_pretrained(name if local else "synthetic:1234", device=self._device,
                                                 **(config.transformer_config or {}))
        elif is_foreign_module(transformer, STDiT3):   # e.g. the reference's own STDiT3 module: geometry + weights are taken over
            cfg, sd = module_state(transformer)
            known = set(STDiT3Config().__dict__)
            cfg = {k: v for k, v in cfg.items() if k in known}
            cfg.update(config.transformer_config or {})
            transformer = STDiT3(STDiT3Config(**cfg), device=self._device)
            transformer.load_state_dict({k: v for k, v in sd.items() if "pos_embed" not in k and "inv_freq" not in k})
        self.transformer = transformer
        self.scheduler = self._check_scheduler(scheduler, "sample", "videosys_amd.rflow.RFLOW") if scheduler is not None else RFLOW(
            num_sampling_steps=config.num_sampling_steps, cfg_scale=config.cfg_scale, use_timestep_transform=True)
        self.text_encoder = build_text_encoder(text_encoder if text_encoder is not None else config.text_encoder, tokenizer,
                                               device=self._device, caption_channels=self.transformer.config.caption_channels,
                                               max_length=self.transformer.config.model_max_length)
        vae = vae if vae is not None else vae_decoder
        if vae is None:
            vae = self._load_vae(config.vae)
        elif is_foreign_module(vae):
            from .vae_open_sora import OpenSoraVAE

            vae = OpenSoraVAE(module_state(vae)[1], device=self._device, micro_batch_size=config.tiling_size)
        self.vae_decoder = vae
        if config.enable_pab:
            pab.set_pab_manager(config.pab_config)
        else:
            pab.set_pab_manager(None)
        self._set_parallel()
        # cpu_offload (pipeline_open_sora.py:241-244): every stage parks its weights in pinned host memory and holds HBM only
        # while it runs — text encoder -> transformer -> VAE decoder, one at a time
        self._init_stages(config.cpu_offload, self._device, text_encoder=getattr(self.text_encoder, "encoder", None),
                          transformer=self.transformer, vae=self.vae_decoder)

    vae = property(lambda self: self.vae_decoder)                                      # register_modules names (:233-235)
    tokenizer = property(lambda self: getattr(self.text_encoder, "tokenizer", None))

    def _after_onload(self, name):
        if name == "transformer":   # attribute paths the sampler reads alias entries of the weight table
            self.transformer.x_embedder.proj.weight = self.transformer.w["x_embedder.proj.weight"]
            self.transformer.y_embedder.y_embedding = self.transformer.w["y_embedder.y_embedding"]

    def _load_vae(self, name):
        """OpenSoraVAE_V1_2 (autoencoder_kl_open_sora.py:738-761): a local checkpoint directory (model.safetensors with the
        reference's keys) or "synthetic:<seed>"; a hub id cannot be fetched here, so it leaves the pipeline latent-only."""
        from .vae_open_sora import OpenSoraVAE_V1_2

        try:
            return OpenSoraVAE_V1_2(from_pretrained=name, device=self._device)
        except FileNotFoundError:
            self._hub_fallback(name, "vae", "no VAE: generate() returns latents")
            return None

    @staticmethod
    def _hub_fallback(name, what, instead):
        """A component name that is neither a local checkpoint directory nor ``synthetic:<seed>``: a Hugging Face hub id
        (``org/name``, the reference's defaults) cannot be fetched offline and is replaced — loudly; anything that looks like a
        filesystem path and does not exist is a typo and raises."""
        n = str(name)
        pathlike = n.startswith(("/", "./", "../", "~")) or n.count("/") != 1     # a hub id is exactly "org/name"
        # "ckpts/stdit3" has the shape of a hub id; when its first component is a directory here it is a mistyped relative path
        pathlike = pathlike or os.path.isdir(n.split("/", 1)[0])
        if pathlike:
            raise FileNotFoundError(f"config.{what} = {name!r}: no such checkpoint directory (expected *.safetensors inside, or 'synthetic:<seed>')")
        import logging

        logging.getLogger("videosys_amd").warning("config.%s = %r is a hub id and cannot be fetched offline: using %s", what, name, instead)

    def _set_parallel(self, dp_size: Optional[int] = None, sp_size: Optional[int] = None, enable_cp: bool = False):
        """pipeline_open_sora.py:253-267: dp=1, sp=world."""
        import torch.distributed as dist

        world = dist.get_world_size() if dist.is_initialized() else 1
        if world > 1:
            self.transformer.enable_parallel(dp_size or 1, sp_size or world, enable_cp)

    def _decode_group(self):
        """The ranks that share one video's VAE decode: the sequence-parallel group of the transformer (every one of its ranks holds
        the same final latents), None on one GPU or with ``shard_vae_decode=False``."""
        if not getattr(self._config, "shard_vae_decode", True) or os.environ.get("VSYS_VAE_SHARD") == "0":
            return None
        sp = getattr(self.transformer, "_sp", None)
        if sp is not None and sp.P > 1:
            return sp.group
        return None

    def null(self, n):
        """pipeline_open_sora.py:294-296."""
        return self.transformer.y_embedder.y_embedding[None].repeat(n, 1, 1)[:, None]

    null_embed = null   # the reference's name (:294)

    def get_text_embeddings(self, texts):
        """pipeline_open_sora.py:269-287: (T5 last hidden state [B, L, 4096], attention mask [B, L]) of already cleaned texts."""
        if self.text_encoder is None:
            raise RuntimeError("no text encoder attached (OpenSoraConfig(text_encoder=<local T5 checkpoint directory>))")
        self._enter_stage("text_encoder")
        emb, mask = self.text_encoder(texts)
        return emb.reshape(emb.shape[0], emb.shape[-2], emb.shape[-1]), mask

    def encode_prompt(self, text):
        """pipeline_open_sora.py:289-292: dict(y=[B, 1, L, 4096], mask=[B, L]) as RFLOW.sample's model_args take it."""
        emb, mask = self.get_text_embeddings(text)
        return dict(y=emb[:, None], mask=mask)

    @staticmethod
    def _basic_clean(text):
        from .caption import basic_clean

        return basic_clean(text)

    @staticmethod
    def _clean_caption(caption):
        from .caption import clean_caption

        return clean_caption(caption)

    @staticmethod
    def text_preprocessing(text: str, use_text_preprocessing: bool = True) -> str:
        """pipeline_open_sora.py:417-424: the training-time caption cleaner applied twice (caption.py; its html / mojibake steps
        use bs4 / ftfy when they are installed and standard-library equivalents otherwise), or lower-case + strip."""
        from .caption import text_preprocessing

        return text_preprocessing(text, use_text_preprocessing)

    @classmethod
    def prepare_prompt(cls, prompt: str, aes: Optional[float] = 6.5, flow: Optional[float] = None, camera_motion=None,
                       loop_i: int = 0) -> str:
        """What generate() feeds the tokenizer for loop ``loop_i`` (pipeline_open_sora.py:548-615, 705-792): an optional JSON
        tail (``{"reference_path": ..., "mask_strategy": ...}``, consumed by generate() through
        open_sora_condition.extract_json_from_prompts) is split off, a ``|0| text |k| text`` schedule is resolved to the segment
        that covers the loop, the score tags the Open-Sora 1.2 checkpoints were trained with are appended, and the text is
        lower-cased."""
        import json

        text, brace, tail = prompt.partition("{")
        if brace:
            unknown = set(json.loads(brace + tail)) - {"reference_path", "mask_strategy"}
            assert not unknown, f"Invalid key: {sorted(unknown)[0]}"
        if text.startswith("|0|"):
            fields = text.split("|")[1:]            # start, text, start, text, ...
            starts = [int(v) for v in fields[0::2]]
            texts = [v.strip() for v in fields[1::2]]
            text = texts[max(i for i, s0 in enumerate(starts) if s0 <= loop_i)]
        for tag, val, fmt in (("aesthetic score:", aes, "{:.1f}"), ("motion score:", flow, "{:.1f}"), ("camera motion:", camera_motion, "{}")):
            if val is not None and tag not in text:
                text = f"{text} {tag} {fmt.format(val)}."
        return cls.text_preprocessing(text)

    @torch.no_grad()
    def generate(self, prompt=None, resolution="480p", aspect_ratio="9:16", num_frames="2s", loop: int = 1,
                 llm_refine: bool = False, negative_prompt: str = "", seed: int = -1, ms: Optional[str] = "",
                 refs: Optional[str] = "", aes: Optional[float] = 6.5, flow: Optional[float] = None, camera_motion=None,
                 condition_frame_length: int = 5, align: int = 5, condition_frame_edit: float = 0.0, return_dict: bool = True,
                 verbose: bool = True, *, height: Optional[int] = None, width: Optional[int] = None,
                 prompt_embeds: Optional[torch.Tensor] = None, prompt_mask: Optional[torch.Tensor] = None,
                 fps: float = 24.0, output_type: str = "auto"):
        """pipeline_open_sora.py:426-656 with the reference's positional / keyword call shape:
        ``engine.generate(prompt, resolution="480p", aspect_ratio="9:16", num_frames="2s")`` (examples/open_sora/sample.py).

        Geometry comes from the reference's vocabulary (open_sora_geometry.get_image_size / get_num_frames); keyword-only
        ``height=`` / ``width=`` override it — an extension, needed because the reference's own tables cannot name 512x512
        (("512", "1:1") fails its assert).  ``prompt_embeds`` / ``prompt_mask`` bypass the text encoder.

        Image / video conditioning (:528-535,607-645): ``refs`` names the reference clips of the prompt (';'-separated image
        paths as in the reference, or a list of paths / pixel tensors [3, T, H, W] in [-1, 1] / latents [4, T, h, w]), ``ms`` the
        mask strategy that pastes their latent frames into the start noise (open_sora_condition.py); both may also arrive as a
        JSON tail of the prompt.  ``loop`` > 1 generates that many clips, each one conditioned on the last
        ``condition_frame_length`` latent frames of the previous (encoded again by the VAE), and returns them joined in time with
        the overlap removed.  (As written, the reference's ``video_clips[i][:, dframe_to_frame(...):]`` / ``torch.cat(dim=1)``
        at :641-643 slice and join the CHANNEL axis of [B, C, T, H, W] clips, which drops every clip after the first; the
        time axis — what the upstream Open-Sora code these lines come from operates on — is used here.)"""
        from . import open_sora_condition as K
        from . import open_sora_geometry as G

        image_size = (int(height), int(width)) if height is not None and width is not None else G.get_image_size(resolution, aspect_ratio)
        num_frames = G.get_num_frames(num_frames)
        seed = self._set_seed(seed)   # (:253-257) -1 draws a fresh seed on rank 0 and broadcasts it; + dp_rank in a process group

        # ---- conditioning inputs: one entry per prompt (:528-535)
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
        prompts = None if prompt is None else ([prompt] if isinstance(prompt, str) else list(prompt))
        given_embeds = prompt_embeds is not None        # the caller's embeddings win over `prompt` for EVERY clip of a loop
        n = len(prompts) if prompts is not None else prompt_embeds.shape[0]
        per_prompt = lambda v: list(v) if isinstance(v, (list, tuple)) and len(v) == n and not torch.is_tensor(v) \
            and all(isinstance(e, (str, list, tuple, type(None))) for e in v) else [v] * n
        ms_list = per_prompt(ms if ms is not None else "")
        ref_list = per_prompt(refs if refs is not None else "")
        if prompts is not None:
            _, ref_list, ms_list = K.extract_json_from_prompts(prompts, ref_list, ms_list)
        conditioned = loop != 1 or any(ms_list) or any(len(r) > 0 for r in ref_list if r is not None)
        vae = self.vae_decoder
        encode = None
        if getattr(vae, "has_encoder", False):
            encode = lambda v: vae.encode(v.to(self._device))      # pixels [B, 3, T, H, W] in [-1, 1] -> latents (seeded: set_seed)
        if loop > 1 and (vae is None or encode is None or output_type == "latent"):
            raise RuntimeError("loop > 1 conditions every clip on the decoded + re-encoded previous one: it needs a VAE with "
                               "encoder weights attached and pixel output")
        if conditioned:
            self._enter_stage("vae")
        refs_x = K.collect_references_batch(ref_list, encode, image_size) if conditioned else None

        pab.update_steps(self._config.num_sampling_steps)
        T, Hl, Wl = get_latent_size(num_frames, *image_size)
        g = torch.Generator(device="cpu")
        g.manual_seed(seed)
        clips = []
        samples = None
        for loop_i in range(loop):
            if prompt_embeds is None or (loop_i > 0 and prompts is not None and not given_embeds):
                if self.text_encoder is None:
                    raise RuntimeError("no text encoder attached: pass prompt_embeds=[B,1,L,4096] (+ prompt_mask), or give "
                                       "OpenSoraConfig(text_encoder=<local T5 checkpoint directory>)")
                texts = [self.prepare_prompt(q, aes=aes, flow=flow, camera_motion=camera_motion, loop_i=loop_i) for q in prompts]
                self._enter_stage("text_encoder")
                prompt_embeds, prompt_mask = self.text_encoder(texts[0] if len(texts) == 1 else texts)
            if loop_i > 0:   # the clip just decoded becomes a reference of every sample (:613-617)
                self._enter_stage("vae")
                refs_x, ms_list = K.append_generated(encode, clips[-1], refs_x, ms_list, loop_i, condition_frame_length,
                                                     condition_frame_edit)
            self._enter_stage("transformer")
            self.transformer.reset_pab_state()
            self.transformer.reset_text_cache()   # per-prompt projections never outlive a sampling run
            B = prompt_embeds.shape[0]
            z = torch.randn(B, self.transformer.in_channels, T, Hl, Wl, generator=g, dtype=torch.float32)
            z = z.to(torch.bfloat16).float()  # the reference draws z in bf16 (pipeline_open_sora.py:622-624)
            # the conditioning scalars are tensors of the MODEL dtype, as in the reference (data_process.py:798-805): 854 is 856 there
            margs = dict(y=prompt_embeds, mask=prompt_mask)
            margs.update(G.prepare_multi_resolution_info(B, image_size, num_frames, fps, dtype=self.transformer.dtype))
            masks = None
            if conditioned:
                masks = K.apply_mask_strategy(z, [[r.to("cpu", torch.float32) for r in rs] for rs in refs_x], ms_list, loop_i, align=align)
                if masks is not None and bool((masks == 1).all()):
                    masks = None      # nothing pasted for this loop: the plain sampler (an all-one mask changes no value)
            samples = self.scheduler.sample(self.transformer, z, margs, self.null(B), device=self._device, progress=verbose,
                                            mask=masks)
            dsp.check_exchange(self.transformer)   # a timed-out peer-to-peer exchange left stale rows: raise here, not a corrupt video
            if vae is None or output_type == "latent":
                break
            self._enter_stage("vae")
            grp = self._decode_group() if loop == 1 else None      # (loop > 1 re-encodes the decoded clip: every rank needs its pixels)
            if grp is not None and hasattr(vae, "decode_sharded"):
                # one video over N GPUs: every rank decodes its block of output frames, one all-gather of uint8 frames
                video = vae.decode_sharded(samples.to(torch.bfloat16), num_frames, grp, to_uint8=True).to("cpu")
                self._enter_stage(None)
                return VideoSysPipelineOutput(video=video) if return_dict else (video,)
            clips.append(vae(samples.to(torch.bfloat16), num_frames=num_frames))
        self._enter_stage(None)
        if not clips:
            out = VideoSysPipelineOutput(video=samples)
            return out if return_dict else (samples,)
        skip = K.dframe_to_frame(condition_frame_length) if loop > 1 else 0
        video = clips[0] if loop == 1 else torch.cat([clips[0]] + [c[:, :, skip:] for c in clips[1:]], dim=2)
        video = (video.clamp(-1, 1) * 0.5 + 0.5).mul(255).add_(0.5).clamp_(0, 255).permute(0, 2, 3, 4, 1).to("cpu", torch.uint8)
        return VideoSysPipelineOutput(video=video) if return_dict else (video,)

    def save_video(self, video, output_path):
        from .utils import save_video

        return save_video(video, output_path, fps=24)   # the reference's frame rate for this pipeline
