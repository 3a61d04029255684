"""Latte pipeline plug-in — host mirror of videosys/pipelines/latte/pipeline_latte.py (LattePABConfig :35-76,
LatteConfig :79-160, LattePipeline :163-960) for the denoising hot path: text embeddings + noise -> latents.

``LatteConfig`` / ``LattePABConfig`` take the same kwargs with the same defaults and drop into ``VideoSysEngine``.
The T5 text encoder and the (temporal-decoder) VAE are pluggable callables like in the Open-Sora pipeline (no weights
offline).  The scheduler is a host-side restatement of diffusers' DDIMScheduler as the pipeline uses it (set_timesteps
"leading", eta = 0, epsilon prediction; diffusers==0.30.0 is pinned by the reference but not vendored) whose per-step
CFG combine + update is one HIP kernel (vsys_cfg_linear_step).
"""
from __future__ import annotations

import math
import os
from typing import Callable, List, Optional

import torch

from .utils import check_prompt_args as _check_prompt_args, ctor_kwargs, progress_wrap, randn_tensor as _randn, read_component
from . import ops, pab
from . import dsp as _dsp
from .latte import LatteT2V, synth_state_dict
from .pab import PABConfig
from .pipeline import VideoSysPipeline, VideoSysPipelineOutput, build_text_encoder, is_foreign_module, module_state

_MLP = {k: {"block": [0, 1, 2, 3, 4], "skip_count": 2} for k in (720, 640, 560, 480, 400)}


class LattePABConfig(PABConfig):
    """pipeline_latte.py:35-76 — identical defaults."""

    def __init__(self, spatial_broadcast: bool = True, spatial_threshold: list = [100, 800], spatial_range: int = 2,
                 temporal_broadcast: bool = True, temporal_threshold: list = [100, 800], temporal_range: int = 3,
                 cross_broadcast: bool = True, cross_threshold: list = [100, 800], cross_range: int = 6,
                 mlp_broadcast: bool = True, mlp_spatial_broadcast_config: dict = None,
                 mlp_temporal_broadcast_config: dict = None):
        super().__init__(
            spatial_broadcast=spatial_broadcast, spatial_threshold=spatial_threshold, spatial_range=spatial_range,
            temporal_broadcast=temporal_broadcast, temporal_threshold=temporal_threshold, temporal_range=temporal_range,
            cross_broadcast=cross_broadcast, cross_threshold=cross_threshold, cross_range=cross_range,
            mlp_broadcast=mlp_broadcast,
            mlp_spatial_broadcast_config=mlp_spatial_broadcast_config or {k: dict(v) for k, v in _MLP.items()},
            mlp_temporal_broadcast_config=mlp_temporal_broadcast_config or {k: dict(v) for k, v in _MLP.items()},
        )


class LatteConfig:
    """pipeline_latte.py:79-160 — identical kwargs/defaults."""

    def __init__(self, model_path: str = "maxin-cn/Latte-1", num_gpus: int = 1, enable_vae_temporal_decoder: bool = True,
                 beta_start: float = 0.0001, beta_end: float = 0.02, beta_schedule: str = "linear",
                 variance_type: str = "learned_range", cpu_offload: bool = False, enable_pab: bool = False,
                 pab_config: PABConfig = None, **extra):
        self.model_path = model_path
        self.pipeline_cls = LattePipeline
        self.num_gpus = num_gpus
        self.enable_vae_temporal_decoder = enable_vae_temporal_decoder
        self.cpu_offload = cpu_offload
        self.beta_start, self.beta_end, self.beta_schedule, self.variance_type = beta_start, beta_end, beta_schedule, variance_type
        self.enable_pab = enable_pab
        self.pab_config = pab_config if pab_config is not None else LattePABConfig()
        self.transformer_config = extra.pop("transformer_config", None)  # extension: geometry override for tests
        if extra:
            raise TypeError(f"unexpected LatteConfig kwargs: {sorted(extra)}")


class DDIMScheduler:
    """diffusers DDIMScheduler as pipeline_latte.py:224-233,802-803,876 uses it."""

    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02,
                 beta_schedule: str = "linear", variance_type: str = "learned_range", set_alpha_to_one: bool = True,
                 steps_offset: int = 0):
        if beta_schedule != "linear":
            raise NotImplementedError("Latte uses the linear beta schedule")
        self.num_train_timesteps = num_train_timesteps
        betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = 1.0 if set_alpha_to_one else float(self.alphas_cumprod[0])
        self.steps_offset = steps_offset
        self.num_inference_steps = None
        self.timesteps: List[int] = []

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        self.timesteps = [int(round(i * ratio)) + self.steps_offset for i in range(num_inference_steps)][::-1]

    def coeffs(self, t: int):
        """prev_sample = c_z * sample + c_eps * eps (eta = 0, no clipping)."""
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_prev = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else self.final_alpha_cumprod
        return math.sqrt(a_prev / a_t), math.sqrt(1 - a_prev) - math.sqrt(a_prev * (1 - a_t) / a_t)


class LattePipeline(VideoSysPipeline):
    vae_scale_factor = 8   # 2 ** (len(vae.config.block_out_channels) - 1) (pipeline_latte.py:239)

    def __init__(self, config: LatteConfig, tokenizer=None, text_encoder=None, vae=None, transformer=None, scheduler=None,
                 device=None, dtype: torch.dtype = torch.bfloat16, *, vae_decoder: Optional[Callable] = None):
        """pipeline_latte.py:192-254, same parameter order.  Components left at None are read from ``config.model_path`` when that
        is a LOCAL checkpoint directory in the published layout (``transformer/``, ``vae_temporal_decoder/`` or ``vae/``,
        ``text_encoder/`` + ``tokenizer/``), else seeded synthetic weights for the transformer and no text encoder / VAE
        (generate() then takes ``prompt_embeds`` and returns latents).  A component may be this build's object, or a torch module
        holding the reference's weights (pipeline.module_state); ``text_encoder="synthetic:<seed>"`` builds an offline stand-in.
        ``dtype``: see VideoSysPipeline._check_dtype (the reference's default here is fp16).  ``vae_decoder`` = ``vae``."""
        self._config = config
        self._dtype = self._check_dtype(dtype)
        self._device = self._resolve_device(device, "LattePipeline")
        name = config.model_path
        if transformer is None or is_foreign_module(transformer, LatteT2V):
            # <model_path>/transformer/{config.json, *.safetensors} (:208-210 LatteT2V.from_pretrained(model_path,
            # subfolder="transformer", video_length=16)); else seeded weights
            file_cfg, sd = module_state(transformer) if transformer is not None else read_component(name, "transformer")
            tcfg = ctor_kwargs(LatteT2V.__init__, file_cfg)
            tcfg.update(config.transformer_config or {})
            transformer = LatteT2V(**tcfg, device=self._device)
            if sd is None:
                seed = int(name.split(":", 1)[1]) if isinstance(name, str) and name.startswith("synthetic:") else 4321
                c = transformer.config
                sd = synth_state_dict(c.num_layers, c.num_attention_heads, c.attention_head_dim, c.caption_channels, c.in_channels,
                                      c.out_channels, c.patch_size, seed=seed)
            transformer.load_state_dict(sd)
        self.transformer = transformer
        self.scheduler = self._check_scheduler(scheduler, "coeffs", "videosys_amd.pipeline_latte.DDIMScheduler") if scheduler is not None else DDIMScheduler(
            beta_start=config.beta_start, beta_end=config.beta_end, beta_schedule=config.beta_schedule, variance_type=config.variance_type)
        vae = vae if vae is not None else vae_decoder
        if vae is None:
            vae = self._load_vae(config)
        elif is_foreign_module(vae):
            vae = self._vae_from_state(config, *module_state(vae))
        if text_encoder is None and isinstance(name, str) and os.path.isdir(os.path.join(name, "text_encoder")):
            text_encoder = os.path.join(name, "text_encoder")   # (:219-223) T5EncoderModel + T5Tokenizer of the checkpoint
        self.text_encoder = build_text_encoder(text_encoder, tokenizer, device=self._device,
                                               caption_channels=self.transformer.config.caption_channels, max_length=120,
                                               tokenizer_path=os.path.join(name, "tokenizer") if isinstance(name, str) else None)
        self.vae_decoder = vae
        pab.set_pab_manager(config.pab_config if config.enable_pab else None)
        self._set_parallel()
        # cpu_offload: each stage's weights live in pinned host memory and are resident only while the stage runs
        self._init_stages(config.cpu_offload, self._device, text_encoder=getattr(self.text_encoder, "encoder", None),
                          transformer=self.transformer, vae=self.vae_decoder)

    vae = property(lambda self: self.vae_decoder)                                      # register_modules names (:238-240)
    tokenizer = property(lambda self: getattr(self.text_encoder, "tokenizer", None))

    def _vae_from_state(self, config, cfg, sd):
        """The decoder object for a VAE state dict (pipeline_latte.py:211-217: the SVD temporal decoder when
        ``enable_vae_temporal_decoder``, else the plain AutoencoderKL)."""
        if config.enable_vae_temporal_decoder:
            from .vae_svd_temporal import AutoencoderKLTemporalDecoder

            return AutoencoderKLTemporalDecoder(sd, device=self._device)
        from .vae_open_sora import AutoencoderKLDecoder

        return AutoencoderKLDecoder(sd, device=self._device, scaling_factor=cfg.get("scaling_factor", 0.18215))

    def _load_vae(self, config):
        """pipeline_latte.py:211-217: ``enable_vae_temporal_decoder=True`` (the reference default) -> the SVD
        ``AutoencoderKLTemporalDecoder`` from ``<model_path>/vae_temporal_decoder``, else the plain ``AutoencoderKL`` from
        ``<model_path>/vae`` — local ``diffusion_pytorch_model.safetensors`` files, or ``"synthetic:<seed>"`` weights (no hub
        access here).  Returns None (latents out) when neither is available."""
        name = config.model_path
        if not isinstance(name, str):
            return None
        if config.enable_vae_temporal_decoder:
            from .vae_svd_temporal import AutoencoderKLTemporalDecoder, synth_state_dict as svd_synth

            if name.startswith("synthetic:"):
                return AutoencoderKLTemporalDecoder(svd_synth(int(name.split(":", 1)[1])), device=self._device)
            sd = read_component(name, "vae_temporal_decoder")[1]
            return AutoencoderKLTemporalDecoder(sd, device=self._device) if sd is not None else None
        from .vae_open_sora import AutoencoderKLDecoder, synth_state_dict as vae_synth

        if name.startswith("synthetic:"):
            pre = "spatial_vae.module."
            sd = {k[len(pre):]: v for k, v in vae_synth(int(name.split(":", 1)[1])).items() if k.startswith(pre)}
            return AutoencoderKLDecoder(sd, device=self._device)
        cfg, sd = read_component(name, "vae")
        return AutoencoderKLDecoder(sd, device=self._device, scaling_factor=cfg.get("scaling_factor", 0.18215)) if sd is not None else None

    def _set_parallel(self, dp_size: Optional[int] = None, sp_size: Optional[int] = None, enable_cp: Optional[bool] = False):
        """pipeline_latte.py:236-250: sp = world size unless given (then dp = world / sp)."""
        import torch.distributed as dist

        world = dist.get_world_size() if dist.is_initialized() else 1
        if world == 1:
            return
        if sp_size is None:
            sp_size, dp_size = world, 1
        else:
            assert world % sp_size == 0, f"world_size {world} must be divisible by sp_size"
            dp_size = world // sp_size
        self.transformer.enable_parallel(dp_size, sp_size, enable_cp)

    def _text_preprocessing(self, text, clean_caption: bool = False):
        """pipeline_latte.py:519-531: a string or a list of strings -> a list; the IF caption cleaner twice, or lower + strip."""
        from .caption import text_preprocessing

        text = [text] if not isinstance(text, (list, tuple)) else text
        return [text_preprocessing(t, clean_caption, mid_strip=False) for t in text]

    def _clean_caption(self, caption):
        """pipeline_latte.py:534-647 (one application; no strip inside the ftfy / unescape step)."""
        from .caption import clean_caption

        return clean_caption(caption, mid_strip=False)

    def mask_text_embeddings(self, emb, mask):
        """pipeline_latte.py:278-284 on ``emb`` [B, 1, L, d], ``mask`` [B, L]: one prompt -> embeddings cut to its token count;
        a batch -> zeroed past each prompt's tokens.  Returns (embeddings, kept length)."""
        if emb.shape[0] == 1:
            keep = int(mask.sum().item())
            return emb[:, :, :keep, :], keep
        return emb * mask[:, None, :, None].to(device=emb.device, dtype=emb.dtype), emb.shape[2]

    def encode_prompt(self, prompt, negative_prompt="", do_classifier_free_guidance: bool = True, num_images_per_prompt: int = 1,
                      device=None, prompt_embeds: Optional[torch.Tensor] = None,
                      negative_prompt_embeds: Optional[torch.Tensor] = None, clean_caption: bool = False,
                      mask_feature: bool = True, dtype=None):
        """pipeline_latte.py:287-445 -> (prompt_embeds, negative_prompt_embeds) [B, L', 4096].  The negative prompt is encoded
        once per prompt at the prompt's padded length (:391-409); with ``mask_feature`` a single prompt's embeddings — and the
        negative ones with them — are cut to the prompt's token count, a batch's prompt embeddings are zeroed past each prompt's
        tokens (:425-442).  The attached text encoder (tokenizer + T5, t5.py) takes the place of :331-363."""
        if num_images_per_prompt != 1:
            raise NotImplementedError("num_images_per_prompt > 1: pass the prompt that many times")
        flat = lambda e: e.reshape(e.shape[0], e.shape[-2], e.shape[-1])     # encoders may return [B, 1, L, d]
        pm = None
        if prompt_embeds is None:
            if self.text_encoder is None:
                raise RuntimeError("no text encoder attached: pass prompt_embeds / negative_prompt_embeds [B, L, 4096]")
            prompt_embeds, pm = self.text_encoder(self._text_preprocessing(prompt, clean_caption))
        prompt_embeds = flat(prompt_embeds)
        nb = prompt_embeds.shape[0]
        if do_classifier_free_guidance and negative_prompt_embeds is None:
            neg = self._text_preprocessing(negative_prompt, clean_caption)
            negative_prompt_embeds, _ = self.text_encoder(neg * nb if len(neg) == 1 else neg)
        if not do_classifier_free_guidance:
            negative_prompt_embeds = None
        elif negative_prompt_embeds is not None:
            negative_prompt_embeds = flat(negative_prompt_embeds)
        if mask_feature and pm is not None:
            prompt_embeds, keep = self.mask_text_embeddings(prompt_embeds[:, None], pm.reshape(nb, -1))
            prompt_embeds = prompt_embeds[:, 0]
            if negative_prompt_embeds is not None:
                negative_prompt_embeds = negative_prompt_embeds[:, :keep]
        return prompt_embeds, negative_prompt_embeds

    def decode_latents(self, latents):
        """pipeline_latte.py:916-927 / :929-948: latents [B, 4, F, h, w] -> uint8 video [B, F, H, W, 3] through the VAE the
        pipeline was built with (``enable_vae_temporal_decoder`` chose it, :211-217); the 1 / scaling_factor and the chunking live
        in the decoder objects (vae_open_sora.AutoencoderKLDecoder, vae_svd_temporal.AutoencoderKLTemporalDecoder)."""
        if self.vae_decoder is None:
            raise RuntimeError("no VAE attached")
        return self.vae_decoder(latents)

    decode_latents_with_temporal_decoder = decode_latents

    def prepare_extra_step_kwargs(self, generator, eta):
        """pipeline_latte.py:448-463: the keywords the scheduler's ``step`` takes beyond (model_output, t, sample) — the DDIM step of diffusers
        (third-party) takes both ``eta`` and ``generator``.  The step here is the fused ops.cfg_linear_step at eta = 0, which draws
        no noise: generate() refuses another eta and uses ``generator`` for the start latents only."""
        return {"eta": eta, "generator": generator}

    def check_inputs(self, prompt, height, width, negative_prompt, callback_steps, prompt_embeds=None, negative_prompt_embeds=None):
        """pipeline_latte.py:465-516: the argument combinations the reference refuses, with its ValueErrors."""
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if not isinstance(callback_steps, int) or isinstance(callback_steps, bool) or callback_steps <= 0:
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps} of type {type(callback_steps)}.")
        _check_prompt_args(prompt, negative_prompt, prompt_embeds, negative_prompt_embeds)

    def prepare_latents(self, batch_size, num_channels_latents, video_length, height, width, dtype, device, generator, latents=None):
        """pipeline_latte.py:649-672: start latents [B, C, F, h / 8, w / 8] * init_noise_sigma, drawn from ``generator`` (the
        reference's ``randn_tensor``: on the generator's device, then moved) unless handed in."""
        shape = (batch_size, num_channels_latents, video_length, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            latents = _randn(shape, generator, dtype)
        return latents.to(device=device) * self.scheduler.init_noise_sigma

    def decode_latents_image(self, latents):
        """pipeline_latte.py:904-914 (``video_length`` 1): float frames [B, F, 3, H, W] in [0, 1], left on the device."""
        if self.vae_decoder is None:
            raise RuntimeError("no VAE attached")
        v = self.vae_decoder.decode(latents).float()      # [B, 3, F, H, W]
        return (v / 2.0 + 0.5).clamp(0, 1).permute(0, 2, 1, 3, 4).contiguous()

    @torch.no_grad()
    def generate(self, prompt=None, negative_prompt: str = "", num_inference_steps: int = 50, guidance_scale: float = 7.5,
                 seed: int = -1, verbose: bool = True, *, prompt_embeds: Optional[torch.Tensor] = None,
                 negative_prompt_embeds: Optional[torch.Tensor] = None, prompt_mask: Optional[torch.Tensor] = None,
                 negative_mask: Optional[torch.Tensor] = None, latents: Optional[torch.Tensor] = None,
                 video_length: int = 16, height: int = 512, width: int = 512, output_type: str = "auto",
                 clean_caption: bool = True, mask_feature: bool = True, num_images_per_prompt: int = 1, eta: float = 0.0,
                 generator: Optional[torch.Generator] = None, return_dict: bool = True, callback: Optional[Callable] = None,
                 callback_steps: int = 1, enable_temporal_attentions: bool = True):
        """pipeline_latte.py:675-900 for text-to-video: CFG batch [negative | prompt], learned-sigma half dropped,
        DDIM eta = 0.  ``height/width/video_length`` are the reference's hard-coded 512/512/16 by default (:764-766);
        BASELINE config 1 (256x256) passes them explicitly.  ``clean_caption`` (default True as :692): prompt and negative prompt
        go through _text_preprocessing (:519-531: the IF caption cleaner twice, caption.py) before the tokenizer.  Text prompts
        follow encode_prompt (:287-445): the negative prompt is encoded once per prompt at the same length, and with
        ``mask_feature`` (default) a single prompt's embeddings — and the negative ones with them — are CUT to the prompt's token
        count, a batch's are zeroed past each prompt's tokens; the transformer then sees no attention mask, as in the reference.
        ``prompt_mask`` / ``negative_mask`` (an extension) apply to embeddings the caller hands over.  The remaining reference
        keywords keep their meaning: ``generator`` draws the start latents, ``callback(i, t, latents)`` every ``callback_steps``
        steps, ``return_dict=False`` returns a tuple, ``eta`` must be 0 (DDIM as the reference runs it), ``num_images_per_prompt`` 1."""
        if eta != 0.0:
            raise NotImplementedError("the scheduler step is DDIM with eta = 0 (what the reference pipeline runs)")
        if num_images_per_prompt != 1:
            raise NotImplementedError("num_images_per_prompt > 1: pass the prompt that many times")
        # (:768) the default negative_prompt "" only counts next to a text prompt: embeddings handed in come with their own negatives
        self.check_inputs(prompt, height, width, negative_prompt if prompt is not None else None, callback_steps, prompt_embeds,
                          negative_prompt_embeds)
        if prompt_embeds is None:
            self._enter_stage("text_encoder")
            prompt_embeds, negative_prompt_embeds = self.encode_prompt(prompt, negative_prompt, guidance_scale > 1.0,
                                                                       clean_caption=clean_caption, mask_feature=mask_feature)
            prompt_mask = negative_mask = None
        cfg = guidance_scale > 1.0        # do_classifier_free_guidance (:749): without it the model runs on the prompt batch alone
        seed = self._set_seed(seed)   # -1: a fresh seed per call, drawn on rank 0 and broadcast; + dp_rank in a process group
        self._enter_stage("transformer")
        pab.update_steps(num_inference_steps)
        self.transformer.reset_pab_state()
        self.transformer.reset_text_cache()   # per-prompt projections never outlive a generate()
        B = prompt_embeds.shape[0]
        if cfg:
            emb = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0)
            mask = None if prompt_mask is None else torch.cat([negative_mask, prompt_mask], dim=0)
        else:
            emb, mask = prompt_embeds, prompt_mask
        nb = (2 if cfg else 1) * B
        self.scheduler.set_timesteps(num_inference_steps)
        ts = self.scheduler.timesteps
        cin = self.transformer.in_channels
        if latents is None and generator is None:
            generator = torch.Generator(device="cpu").manual_seed(seed)
        z = self.prepare_latents(B, cin, video_length, height, width, torch.float32, self._device, generator,
                                 None if latents is None else latents.float()).contiguous().clone()
        all_ts = torch.tensor(ts)
        for step_i, t in progress_wrap(list(enumerate(ts)), verbose):   # (:834-835) tqdm on rank 0
            tt = torch.full((nb,), t, dtype=torch.int64)
            out = self.transformer(z, timestep=tt, all_timesteps=all_ts, encoder_hidden_states=emb,
                                   encoder_attention_mask=mask, added_cond_kwargs={"resolution": None, "aspect_ratio": None},
                                   enable_temporal_attentions=enable_temporal_attentions, return_dict=False)[0]
            c_z, c_eps = self.scheduler.coeffs(t)
            if not cfg:   # the step kernel combines two halves: hand it the prediction twice at guidance 1 (u + 1 (u - u) = u exactly)
                out = torch.cat([out, out], 0)
            ops.cfg_linear_step(z, out, guidance_scale if cfg else 1.0, c_z, c_eps, cond_first=False)
            if callback is not None and step_i % callback_steps == 0:   # (:880-885)
                callback(step_i, t, z)
        _dsp.check_exchange(self.transformer)   # a timed-out peer-to-peer exchange left stale rows: raise here, not a corrupt video
        if self.vae_decoder is None or output_type in ("latent", "latents"):
            self._enter_stage(None)
            return VideoSysPipelineOutput(video=z) if return_dict else (z,)
        self._enter_stage("vae")
        # (:884-891) one frame: float image frames; else uint8 video through the decoder the config chose
        video = self.decode_latents_image(z) if z.shape[2] == 1 and hasattr(self.vae_decoder, "decode") else self.decode_latents(z)
        self._enter_stage(None)
        return VideoSysPipelineOutput(video=video) if return_dict else (video,)

    def save_video(self, video, output_path):
        from .utils import save_video

        return save_video(video, output_path, fps=8)   # the reference's frame rate for this pipeline
