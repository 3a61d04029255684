"""CogVideoX pipeline plug-in — host mirror of videosys/pipelines/cogvideox/pipeline_cogvideox.py (CogVideoXPABConfig
:33-45, CogVideoXConfig :48-118, CogVideoXPipeline :121-755) for the denoising hot path: text embeddings + noise ->
latents.  ``CogVideoXConfig`` / ``CogVideoXPABConfig`` take the same kwargs with the same defaults and drop into
``VideoSysEngine``.  T5 and the causal 3-D VAE (tiled decode) are pluggable callables (no weights offline; the VAE decode
kernels are a later row of SURVEY.md §8f).

Scheduler: host restatement of schedulers/scheduling_ddim_cogvideox.py (scaled-linear betas, SNR shift, zero terminal
SNR, "trailing" spacing, v-prediction, eta = 0) whose per-step CFG combine + update is one HIP kernel
(vsys_cfg_linear_step); the dynamic-CFG guidance (:702-705) is a host scalar per step.
"""
from __future__ import annotations

import math
import os
from typing import Callable, List, Optional

import numpy as np
import torch

from .utils import check_prompt_args as _check_prompt_args, ctor_kwargs, progress_wrap, randn_tensor as _randn, read_component
from . import ops, pab
from . import dsp as _dsp
from .cogvideox import CogVideoXTransformer3DModel, synth_state_dict
from .pab import PABConfig
from .pipeline import VideoSysPipeline, VideoSysPipelineOutput, build_text_encoder, is_foreign_module, module_state


class CogVideoXPABConfig(PABConfig):
    """pipeline_cogvideox.py:33-45 — identical defaults (spatial broadcast only)."""

    def __init__(self, spatial_broadcast: bool = True, spatial_threshold: list = [100, 850], spatial_range: int = 2):
        super().__init__(spatial_broadcast=spatial_broadcast, spatial_threshold=spatial_threshold, spatial_range=spatial_range)


class CogVideoXConfig:
    """pipeline_cogvideox.py:48-118 — identical kwargs/defaults."""

    def __init__(self, model_path: str = "THUDM/CogVideoX-2b", num_gpus: int = 1, cpu_offload: bool = False,
                 vae_tiling: bool = True, enable_pab: bool = False, pab_config=None, **extra):
        self.model_path = model_path
        self.pipeline_cls = CogVideoXPipeline
        self.num_gpus = num_gpus
        self.cpu_offload = cpu_offload
        self.vae_tiling = vae_tiling
        self.enable_pab = enable_pab
        self.pab_config = pab_config if pab_config is not None else CogVideoXPABConfig()
        self.transformer_config = extra.pop("transformer_config", None)  # extension: geometry override for tests
        if extra:
            raise TypeError(f"unexpected CogVideoXConfig kwargs: {sorted(extra)}")


# geometry of the two published checkpoints (HF transformer/config.json; the code defaults are the 2B values)
_GEOMETRY = {
    "THUDM/CogVideoX-2b": dict(num_attention_heads=30, num_layers=30, use_rotary_positional_embeddings=False),
    "THUDM/CogVideoX-5b": dict(num_attention_heads=48, num_layers=42, use_rotary_positional_embeddings=True),
}


class CogVideoXDDIMScheduler:
    """schedulers/scheduling_ddim_cogvideox.py:118-393 with the values of the checkpoints' scheduler_config.json
    (v_prediction, trailing spacing, zero terminal SNR, snr_shift_scale 3.0 for 2b / 1.0 for 5b)."""

    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 set_alpha_to_one=True, steps_offset=0, prediction_type="v_prediction", timestep_spacing="trailing",
                 rescale_betas_zero_snr=True, snr_shift_scale=3.0):
        if beta_schedule != "scaled_linear" or prediction_type != "v_prediction":
            raise NotImplementedError("CogVideoX uses scaled_linear betas with v-prediction")
        self.num_train_timesteps = num_train_timesteps
        betas = torch.linspace(beta_start**0.5, beta_end**0.5, num_train_timesteps, dtype=torch.float64) ** 2
        ac = torch.cumprod(1.0 - betas, dim=0)
        ac = ac / (snr_shift_scale + (1 - snr_shift_scale) * ac)
        if rescale_betas_zero_snr:  # :87-115
            s = ac.sqrt()
            s0, sT = s[0].clone(), s[-1].clone()
            ac = ((s - sT) * (s0 / (s0 - sT))) ** 2
        self.alphas_cumprod = ac
        self.final_alpha_cumprod = 1.0 if set_alpha_to_one else float(ac[0])
        self.timestep_spacing, self.steps_offset = timestep_spacing, steps_offset
        self.num_inference_steps = None
        self.timesteps: List[int] = []

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        n = self.num_train_timesteps
        if self.timestep_spacing == "trailing":
            ts = np.round(np.arange(n, 0, -n / num_inference_steps)).astype(np.int64) - 1
        elif self.timestep_spacing == "leading":
            ts = (np.arange(0, num_inference_steps) * (n // num_inference_steps)).round()[::-1].astype(np.int64) + self.steps_offset
        else:
            ts = np.linspace(0, n - 1, num_inference_steps).round()[::-1].astype(np.int64)
        self.timesteps = [int(v) for v in ts]

    def coeffs(self, t: int):
        """prev_sample = c_z * sample + c_v * v_pred (:358-388)."""
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_prev = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else self.final_alpha_cumprod
        a_coef = math.sqrt((1 - a_prev) / (1 - a_t))
        b = math.sqrt(a_prev) - math.sqrt(a_t) * a_coef
        return a_coef + b * math.sqrt(a_t), -b * math.sqrt(1 - a_t)


class CogVideoXDPMScheduler(CogVideoXDDIMScheduler):
    """schedulers/scheduling_dpm_cogvideox.py:119-483 — the stochastic DPM-Solver++ (2M) step the reference pipeline runs when a
    caller hands this scheduler in (pipeline_cogvideox.py:679-680,711-721): same noise schedule and timesteps as the DDIM class;
    per step  x0 = sqrt(a_t) x - sqrt(1 - a_t) v  and, with h the log-SNR step,
        first step / last step:   x <- m1 x - m2 x0 + mn n1
        otherwise:                x <- m1 x - m2 ((1 + 1/2r) x0 - (1/2r) x0_prev) + mn n2      (n1 is drawn and discarded, :439-447)
    with fresh standard-normal noise every step (two draws on the second-order steps, in this order — the generator's stream is part
    of the result).  ``multipliers`` holds the coefficient arithmetic in float64 (the reference's 0-dim float64 tensors, :402-415)."""

    def multipliers(self, t: int, t_back: Optional[int]):
        """(sa, sb, m1, m2, m3, m4, mn, second_order) for timestep ``t`` with the previous step's timestep ``t_back`` (None: first)."""
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        one = torch.tensor(1.0, dtype=torch.float64)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else one * self.final_alpha_cumprod
        lam = ((a_t / (1 - a_t)) ** 0.5).log()
        lam_next = ((a_prev / (1 - a_prev)) ** 0.5).log()
        h = lam_next - lam
        m1 = ((1 - a_prev) / (1 - a_t)) ** 0.5 * (-h).exp()
        m2 = (-2 * h).expm1() * a_prev ** 0.5
        mn = (1 - a_prev) ** 0.5 * (1 - (-2 * h).exp()) ** 0.5
        m3 = m4 = None
        if t_back is not None:
            a_back = self.alphas_cumprod[t_back]
            r = (lam - ((a_back / (1 - a_back)) ** 0.5).log()) / h
            m3, m4 = float(1 + 1 / (2 * r)), float(1 / (2 * r))
        return float(a_t ** 0.5), float((1 - a_t) ** 0.5), float(m1), float(m2), m3, m4, float(mn), (t_back is not None and prev_t >= 0)

    @staticmethod
    def _noise(shape, generator, device, dtype):
        """diffusers' randn_tensor (third-party): a CPU generator draws on the CPU, then the noise moves to ``device``."""
        gdev = device if generator is None else generator.device
        return torch.randn(tuple(shape), generator=generator, device=gdev, dtype=dtype).to(device)

    def step(self, model_output, old_pred_original_sample, timestep, timestep_back, sample, eta: float = 0.0,
             use_clipped_model_output: bool = False, generator=None, variance_noise=None, return_dict: bool = False):
        """The reference's signature and return value (:348-457): (prev_sample, pred_original_sample)."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        sa, sb, m1, m2, m3, m4, mn, second = self.multipliers(int(timestep), None if timestep_back is None else int(timestep_back))
        x0 = sa * sample - sb * model_output
        noise = self._noise(sample.shape, generator, sample.device, sample.dtype)
        prev = m1 * sample - m2 * x0 + mn * noise
        if old_pred_original_sample is None or not second:
            return prev, x0
        d = m3 * x0 - m4 * old_pred_original_sample
        noise = self._noise(sample.shape, generator, sample.device, sample.dtype)
        return m1 * sample - m2 * d + mn * noise, x0


def get_resize_crop_region_for_grid(src, tgt_width, tgt_height):
    """pipeline_cogvideox.py:757-775."""
    h, w = src
    if h / w > tgt_height / tgt_width:
        rh, rw = tgt_height, int(round(tgt_height / h * w))
    else:
        rw, rh = tgt_width, int(round(tgt_width / w * h))
    top, left = int(round((tgt_height - rh) / 2.0)), int(round((tgt_width - rw) / 2.0))
    return (top, left), (top + rh, left + rw)


def get_3d_rotary_pos_embed(embed_dim, crops_coords, grid_size, temporal_size, theta: float = 10000.0):
    """modules/embeddings.py:283-355 (use_real): constant cos / sin tables [T*H*W, embed_dim], built once on the host."""
    (s0, s1), (e0, e1) = crops_coords
    gh = torch.from_numpy(np.linspace(s0, e0, grid_size[0], endpoint=False, dtype=np.float32))
    gw = torch.from_numpy(np.linspace(s1, e1, grid_size[1], endpoint=False, dtype=np.float32))
    gt = torch.from_numpy(np.linspace(0, temporal_size, temporal_size, endpoint=False, dtype=np.float32))
    dt, dh, dw = embed_dim // 4, embed_dim // 8 * 3, embed_dim // 8 * 3

    def axis(grid, d):
        f = 1.0 / (theta ** (torch.arange(0, d, 2).float() / d))
        return torch.einsum("n,f->nf", grid, f).repeat_interleave(2, dim=-1)

    ft, fh, fw = axis(gt, dt), axis(gh, dh), axis(gw, dw)
    T, H, W = temporal_size, grid_size[0], grid_size[1]
    freqs = torch.cat([ft[:, None, None, :].expand(T, H, W, dt), fh[None, :, None, :].expand(T, H, W, dh),
                       fw[None, None, :, :].expand(T, H, W, dw)], dim=-1).reshape(T * H * W, -1)
    return freqs.cos().contiguous(), freqs.sin().contiguous()


class _CustomTimestepsUnsupported(ValueError, NotImplementedError):
    """ValueError as the reference raises it; also a NotImplementedError for callers of earlier versions of this package."""


class CogVideoXPipeline(VideoSysPipeline):
    vae_scale_factor_spatial = 8
    vae_scale_factor_temporal = 4

    def __init__(self, config: CogVideoXConfig, tokenizer=None, text_encoder=None, vae=None, transformer=None, scheduler=None,
                 device=None, dtype: torch.dtype = torch.bfloat16, *, vae_decoder: Optional[Callable] = None):
        """pipeline_cogvideox.py:124-187, same parameter order.  Components left at None are read from ``config.model_path`` when
        that is a LOCAL checkpoint directory in the published layout (``transformer/``, ``scheduler/``, ``vae/``, ``text_encoder/`` +
        ``tokenizer/``); a hub id selects the published geometry with seeded synthetic weights and no text encoder / VAE
        (generate() then takes ``prompt_embeds`` and returns latents).  A component may be this build's object, or a torch module
        holding the reference's weights (pipeline.module_state); ``text_encoder="synthetic:<seed>"`` builds an offline stand-in.
        ``dtype``: see VideoSysPipeline._check_dtype (the reference switches the 2b model to fp16).  ``vae_decoder`` = ``vae``."""
        self._config = config
        self._dtype = self._check_dtype(dtype)
        self._device = self._resolve_device(device, "CogVideoXPipeline")
        name = config.model_path
        base = name.split("@", 1)[0] if isinstance(name, str) else ""
        if transformer is None or is_foreign_module(transformer, CogVideoXTransformer3DModel):
            # <model_path>/transformer/{config.json, *.safetensors} (:141-144 from_pretrained); else the published geometry of the
            # hub id + seeded weights
            file_cfg, sd = module_state(transformer) if transformer is not None else read_component(name, "transformer")
            tcfg = dict(_GEOMETRY.get(base, {}))
            tcfg.update(ctor_kwargs(CogVideoXTransformer3DModel.__init__, file_cfg))
            tcfg.update(config.transformer_config or {})
            transformer = CogVideoXTransformer3DModel(**tcfg, device=self._device)
            if sd is None:
                seed = int(name.rsplit(":", 1)[1]) if isinstance(name, str) and ":" in name and name.rsplit(":", 1)[1].isdigit() else 777
                c = transformer.config
                sd = synth_state_dict(c.num_layers, c.num_attention_heads, c.attention_head_dim, c.text_embed_dim, c.in_channels,
                                      c.out_channels, c.time_embed_dim, c.patch_size, seed=seed)
            transformer.load_state_dict(sd)
        self.transformer = transformer
        is_5b = base.endswith("5b") or self.transformer.config.num_layers == 42
        if scheduler is None:   # <model_path>/scheduler/scheduler_config.json (:155-158)
            sched_cfg = ctor_kwargs(CogVideoXDDIMScheduler.__init__, read_component(name, "scheduler")[0])
            sched_cfg.setdefault("snr_shift_scale", 1.0 if is_5b else 3.0)
            scheduler = CogVideoXDDIMScheduler(**sched_cfg)
        self.scheduler = self._check_scheduler(scheduler, "coeffs", "videosys_amd.pipeline_cogvideox.CogVideoXDDIMScheduler")
        vae = vae if vae is not None else vae_decoder
        if vae is None:
            vae = self._load_vae(config, base, is_5b)
        elif is_foreign_module(vae):
            from .vae_cogvideox import CogVideoXVAE

            cfg, sd = module_state(vae)
            vae = CogVideoXVAE(sd, device=self._device, scaling_factor=cfg.get("scaling_factor", 0.7 if is_5b else 1.15258426),
                               use_tiling=config.vae_tiling)
        if text_encoder is None and isinstance(name, str) and os.path.isdir(os.path.join(name, "text_encoder")):
            text_encoder = os.path.join(name, "text_encoder")   # (:149-153) T5EncoderModel + T5Tokenizer of the checkpoint
        # the reference hands the encoder no attention mask (:244): the padding is attended
        self.text_encoder = build_text_encoder(text_encoder, tokenizer, device=self._device,
                                               caption_channels=self.transformer.config.text_embed_dim, max_length=226,
                                               use_attention_mask=False,
                                               tokenizer_path=os.path.join(name, "tokenizer") if isinstance(name, str) else None)
        self.vae_decoder = vae
        pab.set_pab_manager(config.pab_config if config.enable_pab else None)
        self._set_parallel()
        # cpu_offload: each stage's weights live in pinned host memory and are resident only while the stage runs
        self._init_stages(config.cpu_offload, self._device, text_encoder=getattr(self.text_encoder, "encoder", None),
                          transformer=self.transformer, vae=self.vae_decoder)

    vae = property(lambda self: self.vae_decoder)                                      # register_modules names (:161-163)
    tokenizer = property(lambda self: getattr(self.text_encoder, "tokenizer", None))

    def _load_vae(self, config, base, is_5b=False):
        """pipeline_cogvideox.py:146-147,170-172: AutoencoderKLCogVideoX from ``<model_path>/vae`` (local config.json + safetensors)
        or ``...@synthetic:<seed>`` random weights; tiling per ``config.vae_tiling``.  scaling_factor from the checkpoint's
        config.json, else as published: 1.15258426 (2B), 0.7 (5B)."""
        from .vae_cogvideox import CogVideoXVAE, synth_state_dict as vae_synth

        name = config.model_path
        sf = 0.7 if is_5b else 1.15258426
        if isinstance(name, str) and "@synthetic:" in name:
            return CogVideoXVAE(vae_synth(int(name.rsplit(":", 1)[1])), device=self._device, scaling_factor=sf, use_tiling=config.vae_tiling)
        cfg, sd = read_component(name, "vae")
        if sd is not None:
            return CogVideoXVAE(sd, device=self._device, scaling_factor=cfg.get("scaling_factor", sf), use_tiling=config.vae_tiling)
        return None

    def _set_parallel(self, dp_size: Optional[int] = None, sp_size: Optional[int] = None, enable_cp: Optional[bool] = False):
        """pipeline_cogvideox.py:195-209: sp = world size unless given (then dp = world / sp)."""
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

    def _prepare_rotary_positional_embeddings(self, height: int, width: int, num_frames: int):
        """pipeline_cogvideox.py:449-474."""
        c = self.transformer.config
        gh = height // (self.vae_scale_factor_spatial * c.patch_size)
        gw = width // (self.vae_scale_factor_spatial * c.patch_size)
        crops = get_resize_crop_region_for_grid((gh, gw), 720 // (self.vae_scale_factor_spatial * c.patch_size),
                                                480 // (self.vae_scale_factor_spatial * c.patch_size))
        return get_3d_rotary_pos_embed(c.attention_head_dim, crops, (gh, gw), num_frames)

    _callback_tensor_inputs = ["latents", "prompt_embeds", "negative_prompt_embeds"]   # pipeline_cogvideox.py:117-121
    _guidance_scale, _num_timesteps, _interrupt, fusing_transformer = None, 0, False, False

    guidance_scale = property(lambda self: self._guidance_scale)     # (:476-486) values of the generate() call in flight / last run
    num_timesteps = property(lambda self: self._num_timesteps)
    interrupt = property(lambda self: self._interrupt)

    def prepare_extra_step_kwargs(self, generator, eta):
        """pipeline_cogvideox.py:367-383: the keywords the scheduler's ``step`` takes beyond (model_output, t, sample) — the DDIM step of diffusers
        (third-party) takes both ``eta`` and ``generator``.  The step here is the fused ops.cfg_linear_step at eta = 0, which draws
        no noise: generate() refuses another eta and uses ``generator`` for the start latents only."""
        return {"eta": eta, "generator": generator}

    def check_inputs(self, prompt, height, width, negative_prompt, callback_on_step_end_tensor_inputs, prompt_embeds=None,
                     negative_prompt_embeds=None):
        """pipeline_cogvideox.py:385-434: the argument combinations the reference refuses, with its ValueErrors."""
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        bad = [k for k in (callback_on_step_end_tensor_inputs or ()) if k not in self._callback_tensor_inputs]
        if bad:
            raise ValueError(f"`callback_on_step_end_tensor_inputs` has to be in {self._callback_tensor_inputs}, but found {bad}")
        _check_prompt_args(prompt, negative_prompt, prompt_embeds, negative_prompt_embeds)

    def prepare_latents(self, batch_size, num_channels_latents, num_frames, height, width, dtype, device, generator, latents=None):
        """pipeline_cogvideox.py:334-357: start latents [B, (F - 1) / 4 + 1, C, h / 8, w / 8] * init_noise_sigma, drawn from
        ``generator`` (utils.randn_tensor) unless handed in."""
        shape = (batch_size, (num_frames - 1) // self.vae_scale_factor_temporal + 1, num_channels_latents,
                 height // self.vae_scale_factor_spatial, width // self.vae_scale_factor_spatial)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            latents = _randn(shape, generator, dtype)
        return latents.to(device=device) * self.scheduler.init_noise_sigma

    def fuse_qkv_projections(self) -> None:
        """pipeline_cogvideox.py:436-439.  The transformer here always runs q, k and v as one [3C, C] GEMM (cogvideox.py packs the
        three weights at load), so there is nothing to switch; the flag is kept for call sites that toggle it."""
        self.fusing_transformer = True

    def unfuse_qkv_projections(self) -> None:
        """pipeline_cogvideox.py:441-447 (see fuse_qkv_projections: the packed GEMM stays)."""
        self.fusing_transformer = False

    def _get_t5_prompt_embeds(self, prompt=None, num_videos_per_prompt: int = 1, max_sequence_length: int = 226, device=None,
                              dtype=None):
        """pipeline_cogvideox.py:211-251: prompts -> T5 states [B * num_videos_per_prompt, 226, 4096].  The attached text encoder
        (tokenizer padded / truncated to its ``max_length`` + T5, t5.py; for this pipeline build it with
        ``T5TextEncoder(..., max_length=226, use_attention_mask=False)``: the reference hands the encoder no attention mask, :244) takes the place of :225-244."""
        if self.text_encoder is None:
            raise RuntimeError("no text encoder attached: pass prompt_embeds / negative_prompt_embeds [B, 226, 4096]")
        want = getattr(self.text_encoder, "max_length", max_sequence_length)
        if want != max_sequence_length:
            raise ValueError(f"max_sequence_length {max_sequence_length}: the attached text encoder pads to {want}")
        prompt = [prompt] if isinstance(prompt, str) else list(prompt)
        e = self.text_encoder(prompt)
        e = e[0] if isinstance(e, tuple) else e
        e = e.reshape(len(prompt), e.shape[-2], e.shape[-1])
        return e.repeat_interleave(num_videos_per_prompt, 0) if num_videos_per_prompt > 1 else e

    def encode_prompt(self, prompt, negative_prompt=None, do_classifier_free_guidance: bool = True, num_videos_per_prompt: int = 1,
                      prompt_embeds: Optional[torch.Tensor] = None, negative_prompt_embeds: Optional[torch.Tensor] = None,
                      max_sequence_length: int = 226, device=None, dtype=None):
        """pipeline_cogvideox.py:253-332 -> (prompt_embeds, negative_prompt_embeds); a string negative prompt (default "") is
        used once per prompt of the batch, a list must have the batch's length and the prompt's type."""
        prompt = [prompt] if isinstance(prompt, str) else prompt
        B = len(prompt) if prompt is not None else prompt_embeds.shape[0]
        if prompt_embeds is None:
            prompt_embeds = self._get_t5_prompt_embeds(prompt, num_videos_per_prompt, max_sequence_length)
        if do_classifier_free_guidance and negative_prompt_embeds is None:
            negative_prompt = negative_prompt or ""
            negative_prompt = B * [negative_prompt] if isinstance(negative_prompt, str) else negative_prompt
            if prompt is not None and type(prompt) is not type(negative_prompt):
                raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} != {type(prompt)}.")
            if B != len(negative_prompt):
                raise ValueError(f"`negative_prompt` has batch size {len(negative_prompt)}, but `prompt` has batch size {B}")
            negative_prompt_embeds = self._get_t5_prompt_embeds(negative_prompt, num_videos_per_prompt, max_sequence_length)
        return prompt_embeds, negative_prompt_embeds

    def decode_latents(self, latents: torch.Tensor) -> torch.Tensor:
        """pipeline_cogvideox.py:359-364: latents [B, T, 16, h, w] -> frames [B, 3, T', H, W]; the permute and the
        1 / scaling_factor live in vae_cogvideox.CogVideoXVAE.__call__."""
        if self.vae_decoder is None:
            raise RuntimeError("no VAE attached")
        # under sequence parallelism every rank holds the same final latents: the tiles of the tiled decode are shared out over the
        # ranks and gathered once (VSYS_VAE_SHARD=0: every rank decodes every tile, as the reference does)
        sp = getattr(self.transformer, "_sp", None)
        if sp is not None and sp.P > 1 and os.environ.get("VSYS_VAE_SHARD") != "0" and hasattr(self.vae_decoder, "_tiles_over_ranks"):
            return self.vae_decoder.decode_latents(latents.to(torch.bfloat16), group=sp.group)
        return self.vae_decoder(latents.to(torch.bfloat16))

    @torch.no_grad()
    def generate(self, prompt=None, negative_prompt=None, height: int = 480, width: int = 720, num_frames: int = 49,
                 num_inference_steps: int = 50, guidance_scale: float = 6, use_dynamic_cfg: bool = False, seed: int = -1,
                 verbose: bool = True, *, prompt_embeds: Optional[torch.Tensor] = None,
                 negative_prompt_embeds: Optional[torch.Tensor] = None, latents: Optional[torch.Tensor] = None,
                 output_type: str = "auto", timesteps=None, num_videos_per_prompt: int = 1, eta: float = 0.0,
                 generator: Optional[torch.Generator] = None, return_dict: bool = True, callback_on_step_end=None,
                 callback_on_step_end_tensor_inputs=("latents",), max_sequence_length: int = 226):
        """pipeline_cogvideox.py:498-755 for text-to-video (CFG batch [negative | prompt], eta = 0).  The reference's other
        keywords keep their meaning where the path has them: ``generator`` draws the start latents, ``callback_on_step_end(self,
        i, t, {...})`` sees the tensors named in ``callback_on_step_end_tensor_inputs`` and may hand back new ones (:725-734), ``return_dict=False`` returns a tuple; ``timesteps`` (a
        custom schedule) and ``eta`` != 0 are refused; ``num_videos_per_prompt`` is overridden to 1 exactly as the reference does (:603)."""
        if timesteps is not None:   # the reference's retrieve_timesteps (:47-88) raises ValueError for this scheduler too
            raise _CustomTimestepsUnsupported("The current scheduler class CogVideoXDDIMScheduler's `set_timesteps` does not support "
                                              "custom timestep schedules: the schedule is set from num_inference_steps")
        if eta != 0.0:
            raise NotImplementedError("the scheduler step is DDIM with eta = 0 (what the reference pipeline runs)")
        self.check_inputs(prompt, height, width, negative_prompt, callback_on_step_end_tensor_inputs, prompt_embeds,
                          negative_prompt_embeds)
        self._guidance_scale, self._interrupt = guidance_scale, False
        if prompt_embeds is None:
            if self.text_encoder is None:
                raise RuntimeError("no text encoder attached: pass prompt_embeds / negative_prompt_embeds [B, 226, 4096]")
            self._enter_stage("text_encoder")
            prompt_embeds, negative_prompt_embeds = self.encode_prompt(prompt, negative_prompt, guidance_scale > 1.0,
                                                                       max_sequence_length=max_sequence_length)
        cfg = guidance_scale > 1.0        # do_classifier_free_guidance (:627): without it the model runs on the prompt batch alone
        seed = self._set_seed(seed)   # -1: a fresh seed per call, drawn on rank 0 and broadcast; + dp_rank in a process group
        self._enter_stage("transformer")
        pab.update_steps(num_inference_steps)
        self.transformer.reset_pab_state()
        self.transformer.reset_text_cache()   # per-prompt projections never outlive a generate()
        B = prompt_embeds.shape[0]
        emb = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0) if cfg else prompt_embeds
        nb = (2 if cfg else 1) * B
        self.scheduler.set_timesteps(num_inference_steps)
        c = self.transformer.config
        if latents is None and generator is None:
            generator = torch.Generator(device="cpu").manual_seed(seed)
        z = self.prepare_latents(B, c.in_channels, num_frames, height, width, torch.float32, self._device, generator,
                                 None if latents is None else latents.float()).contiguous().clone()
        self._num_timesteps = len(self.scheduler.timesteps)
        rope = self._prepare_rotary_positional_embeddings(height, width, z.shape[1]) if c.use_rotary_positional_embeddings else None
        zf = z.view(B, 1, -1)  # the step kernel sees [Bz, Cin = 1, thw]: CogVideoX predicts all 16 channels (no sigma half)
        dpm = isinstance(self.scheduler, CogVideoXDPMScheduler)      # (:679-680) DPM-solver++: the previous step's x0 prediction
        # INTENTIONAL DEVIATION (precision, not semantics): the latents stay fp32 between steps and every scheduler product is rounded
        # once per step.  The reference hands its scheduler bf16 latents (latents.to(prompt_embeds.dtype)), so mult[0] * sample,
        # sqrt(alpha) * sample and mult_noise * noise are each rounded to bf16 there (a 0-dim float64 tensor times a bf16 tensor is
        # bf16): this path is MORE precise than the reference's own trajectory and not bit-equal to it; the golden
        # (oracle/make_golden_dpm.py) drives the reference scheduler with fp32 samples for the same reason.
        x0_old, t_back = None, None
        ndt = getattr(self.transformer, "dtype", torch.bfloat16)    # the noise is drawn in the latents' dtype (randn_tensor(sample.dtype), :437)
        # (:678,737) the reference runs diffusers' progress bar on every call; ``verbose=False`` (an extension) turns it off
        for step_i, t in progress_wrap(list(enumerate(self.scheduler.timesteps)), verbose):
            if self._interrupt:   # (:682-683) a callback may set pipe._interrupt: the remaining steps are skipped
                continue
            out = self.transformer(z, emb, torch.full((nb,), t, dtype=torch.int64), image_rotary_emb=rope,
                                   return_dict=False)[0]
            g_t = guidance_scale
            if use_dynamic_cfg:  # :702-705
                g_t = 1 + guidance_scale * ((1 - math.cos(math.pi * ((num_inference_steps - t) / num_inference_steps) ** 5.0)) / 2)
            if not cfg:   # (:706-708 skipped) the step kernel combines two halves: the prediction twice at guidance 1 is the prediction
                out, g_t = torch.cat([out, out], 0), 1.0
            if dpm:
                # (:711-721 -> scheduling_dpm_cogvideox.py:402-447) x0 = sa z - sb v and the update are linear in (z, v): two launches of
                # the fused guidance + step kernel, then the x0_prev and noise terms (1.1 M latent values) as plain tensor adds
                sa, sb, m1, m2, m3, m4, mn, second = self.scheduler.multipliers(t, t_back)
                x0 = z.clone()
                ops.cfg_linear_step(x0.view(B, 1, -1), out.view(2 * B, 1, -1), g_t, sa, -sb, cond_first=False)
                noise = self.scheduler._noise(z.shape, generator, z.device, ndt).float()
                if x0_old is None or not second:
                    ops.cfg_linear_step(zf, out.view(2 * B, 1, -1), g_t, m1 - m2 * sa, m2 * sb, cond_first=False)
                else:
                    noise = self.scheduler._noise(z.shape, generator, z.device, ndt).float()   # the second draw is used
                    ops.cfg_linear_step(zf, out.view(2 * B, 1, -1), g_t, m1 - m2 * m3 * sa, m2 * m3 * sb, cond_first=False)
                    z.add_(x0_old, alpha=m2 * m4)
                z.add_(noise, alpha=mn)
                x0_old, t_back = x0, t
            else:
                c_z, c_v = self.scheduler.coeffs(t)
                ops.cfg_linear_step(zf, out.view(2 * B, 1, -1), g_t, c_z, c_v, cond_first=False)
            z.copy_(z.to(torch.bfloat16).float())  # latents = latents.to(prompt_embeds.dtype) (:723)
            if callback_on_step_end is not None:   # (:725-734) "prompt_embeds" is the [negative | prompt] batch the loop runs on
                have = {"latents": z, "prompt_embeds": emb, "negative_prompt_embeds": negative_prompt_embeds}
                back = callback_on_step_end(self, step_i, t, {k: have[k] for k in callback_on_step_end_tensor_inputs})
                back = back if isinstance(back, dict) else {}
                if back.get("latents") is not None and back["latents"] is not z:
                    z.copy_(back["latents"].to(z.device, z.dtype))
                if back.get("prompt_embeds") is not None and back["prompt_embeds"] is not emb:
                    emb = back["prompt_embeds"]
                    self.transformer.reset_text_cache()
                negative_prompt_embeds = back.get("negative_prompt_embeds", negative_prompt_embeds)
        _dsp.check_exchange(self.transformer)   # a timed-out peer-to-peer exchange left stale rows: raise here, not a corrupt video
        if self.vae_decoder is None or output_type in ("latent", "latents"):
            self._enter_stage(None)
            return VideoSysPipelineOutput(video=z) if return_dict else (z,)
        self._enter_stage("vae")
        frames = self.decode_latents(z)   # (:359-364) -> [B, 3, T, H, W]
        self._enter_stage(None)
        if not torch.is_tensor(frames) or frames.dtype == torch.uint8:
            return VideoSysPipelineOutput(video=frames) if return_dict else (frames,)
        # VideoProcessor.postprocess_video (diffusers, third-party) denormalises to [0, 1]; here: uint8 [B, T, H, W, C] on the CPU
        video = ((frames.float() / 2.0 + 0.5).clamp(0, 1) * 255).round().permute(0, 2, 3, 4, 1).to("cpu", torch.uint8)
        return VideoSysPipelineOutput(video=video) if return_dict else (video,)

    def save_video(self, video, output_path):
        from .utils import save_video

        return save_video(video, output_path, fps=8)   # the reference's frame rate for this pipeline
