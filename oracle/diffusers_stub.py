"""TEST INFRASTRUCTURE ONLY — restatement of the diffusers==0.30.0 leaf modules the reference's Latte / CogVideoX
transformers import (diffusers is pinned in /root/reference/requirements.txt but is NOT installed here and NOT vendored
in the reference tree: SURVEY.md §8c).

With these stubs installed, ``oracle/ref_loader.py`` can import and run the reference's OWN
``videosys/models/transformers/latte_transformer_3d.py`` (block algebra, rearranges, PAB hooks, final layer — all
reference code) on CPU; only the leaves below are restated from the published diffusers 0.30.0 source:

  diffusers.models.attention_processor.Attention (+ AttnProcessor2_0)   to_q/to_k/to_v/to_out, SDPA with additive mask
  diffusers.models.activations.GELU / GEGLU / ApproximateGELU
  diffusers.models.embeddings.PatchEmbed, get_2d_sincos_pos_embed, get_1d_sincos_pos_embed_from_grid,
      Timesteps / get_timestep_embedding, TimestepEmbedding, PixArtAlphaCombinedTimestepSizeEmbeddings,
      PixArtAlphaTextProjection
  diffusers.models.lora.LoRACompatibleLinear / LoRACompatibleConv (plain Linear / Conv2d taking an ignored ``scale``)
  diffusers.configuration_utils.ConfigMixin / register_to_config, diffusers.models.modeling_utils.ModelMixin
  diffusers.schedulers.DDIMScheduler (set_timesteps "leading", step with eta = 0, epsilon prediction)
  diffusers.models.AutoencoderKL — decode side only (post_quant_conv + vae.Decoder: ResnetBlock2D, UNetMidBlock2D with the
      single-head Attention, UpDecoderBlock2D, Upsample2D) at the SDXL-VAE config the Open-Sora pipeline loads

Parity status: the reference holds no test that pins these leaves, so the Latte goldens are "reference block code over
restated diffusers leaves" (parity unpinned for the leaves, stated in DESIGN.md).
Nothing under ``videosys_amd/`` may import this module.
"""
from __future__ import annotations

import functools
import inspect
import math
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------- config / model mixins
class ConfigMixin:
    config_name = "config.json"

    @property
    def config(self):
        return self._internal_dict


def register_to_config(init):
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        sig = inspect.signature(init)
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        self._internal_dict = SimpleNamespace(**cfg)
        init(self, *args, **kwargs)

    return inner


class ModelMixin(nn.Module):
    pass


class BaseOutput(dict):
    pass


def deprecate(*a, **k):
    pass


def maybe_allow_in_graph(cls):
    return cls


# ----------------------------------------------------------------------------------------------- lora shims
class LoRACompatibleLinear(nn.Linear):
    def forward(self, x, scale: float = 1.0):
        return F.linear(x, self.weight, self.bias)


class LoRACompatibleConv(nn.Conv2d):
    def forward(self, x, scale: float = 1.0):
        return F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)


# ----------------------------------------------------------------------------------------------- activations
class GELU(nn.Module):
    """diffusers.models.activations.GELU: gelu(Linear(x)) with optional tanh approximation."""

    def __init__(self, dim_in, dim_out, approximate: str = "none", bias: bool = True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=bias)
        self.approximate = approximate

    def forward(self, x):
        return F.gelu(self.proj(x), approximate=self.approximate)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out, bias: bool = True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2, bias=bias)

    def forward(self, x, scale: float = 1.0):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)


class ApproximateGELU(nn.Module):
    def __init__(self, dim_in, dim_out, bias: bool = True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=bias)

    def forward(self, x):
        x = self.proj(x)
        return x * torch.sigmoid(1.702 * x)


# ----------------------------------------------------------------------------------------------- attention
class Attention(nn.Module):
    """diffusers.models.attention_processor.Attention.  Default forward = AttnProcessor2_0 (torch >= 2); when a
    ``processor`` object is passed (CogVideoXAttnProcessor2_0 of the reference) forward delegates to it, as diffusers does.
    qk_norm="layer_norm" adds norm_q / norm_k = nn.LayerNorm(dim_head, eps=eps) (affine).  Default path:
    q = to_q(x), k = to_k(ctx), v = to_v(ctx), heads split as view(B, L, H, D).transpose(1, 2),
    F.scaled_dot_product_attention(q, k, v, attn_mask=additive mask [B, H, Lq|1, Lk]), to_out[0] (Linear), to_out[1]
    (Dropout).  qk_norm / added projections are not used by the Latte blocks."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, out_bias=True, qk_norm=None, eps=1e-5, processor=None, **unused):
        super().__init__()
        inner = heads * dim_head
        self.processor = processor
        self.norm_q = nn.LayerNorm(dim_head, eps=eps) if qk_norm == "layer_norm" else None
        self.norm_k = nn.LayerNorm(dim_head, eps=eps) if qk_norm == "layer_norm" else None
        assert qk_norm in (None, "layer_norm")
        self.heads = heads
        self.inner_dim = inner
        self.is_cross_attention = cross_attention_dim is not None
        cdim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(cdim, inner, bias=bias)
        self.to_v = nn.Linear(cdim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=out_bias), nn.Dropout(dropout)])

    def prepare_attention_mask(self, attention_mask, target_length, batch_size):
        if attention_mask is None:
            return None
        # [B, 1, Lk] -> [B*H, 1, Lk] (repeat_interleave over heads), as diffusers does for the SDPA processor
        return attention_mask.repeat_interleave(self.heads, dim=0)

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kwargs):
        if self.processor is not None:
            return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                                  attention_mask=attention_mask, **kwargs)
        B, Lq, _ = hidden_states.shape
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        Lk = ctx.shape[1]
        if attention_mask is not None:
            attention_mask = self.prepare_attention_mask(attention_mask, Lk, B)
            attention_mask = attention_mask.view(B, self.heads, -1, attention_mask.shape[-1])
        q = self.to_q(hidden_states)
        k = self.to_k(ctx)
        v = self.to_v(ctx)
        D = self.inner_dim // self.heads
        q = q.view(B, Lq, self.heads, D).transpose(1, 2)
        k = k.view(B, Lk, self.heads, D).transpose(1, 2)
        v = v.view(B, Lk, self.heads, D).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=attention_mask, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(B, Lq, self.inner_dim).to(q.dtype)
        o = self.to_out[0](o)
        return self.to_out[1](o)


# ----------------------------------------------------------------------------------------------- embeddings
def get_1d_sincos_pos_embed_from_grid(embed_dim, pos):
    """diffusers.models.embeddings.get_1d_sincos_pos_embed_from_grid (numpy, float64 omega)."""
    if isinstance(pos, torch.Tensor):
        pos = pos.numpy()
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000**omega
    pos = pos.reshape(-1)
    out = np.einsum("m,d->md", pos, omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_2d_sincos_pos_embed_from_grid(embed_dim, grid):
    emb_h = get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[0])
    emb_w = get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[1])
    return np.concatenate([emb_h, emb_w], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False, extra_tokens=0, interpolation_scale=1.0, base_size=16):
    if isinstance(grid_size, int):
        grid_size = (grid_size, grid_size)
    grid_h = np.arange(grid_size[0], dtype=np.float32) / (grid_size[0] / base_size) / interpolation_scale
    grid_w = np.arange(grid_size[1], dtype=np.float32) / (grid_size[1] / base_size) / interpolation_scale
    grid = np.meshgrid(grid_w, grid_h)  # w goes first
    grid = np.stack(grid, axis=0)
    grid = grid.reshape([2, 1, grid_size[1], grid_size[0]])
    return get_2d_sincos_pos_embed_from_grid(embed_dim, grid)


class PatchEmbed(nn.Module):
    """diffusers.models.embeddings.PatchEmbed (pos_embed_type="sincos", no layer norm, flatten)."""

    def __init__(self, height=224, width=224, patch_size=16, in_channels=3, embed_dim=768, layer_norm=False, flatten=True,
                 bias=True, interpolation_scale=1, pos_embed_type="sincos", pos_embed_max_size=None):
        super().__init__()
        num_patches = (height // patch_size) * (width // patch_size)
        self.proj = nn.Conv2d(in_channels, embed_dim, kernel_size=(patch_size, patch_size), stride=patch_size, bias=bias)
        self.patch_size = patch_size
        self.height, self.width = height // patch_size, width // patch_size
        self.base_size = height // patch_size
        self.interpolation_scale = interpolation_scale
        pos = get_2d_sincos_pos_embed(embed_dim, int(num_patches**0.5), base_size=self.base_size,
                                      interpolation_scale=self.interpolation_scale)
        self.register_buffer("pos_embed", torch.from_numpy(pos).float().unsqueeze(0), persistent=False)

    def forward(self, latent):
        height, width = latent.shape[-2] // self.patch_size, latent.shape[-1] // self.patch_size
        latent = self.proj(latent).flatten(2).transpose(1, 2)
        if self.height != height or self.width != width:
            pos = get_2d_sincos_pos_embed(self.pos_embed.shape[-1], (height, width), base_size=self.base_size,
                                          interpolation_scale=self.interpolation_scale)
            pos = torch.from_numpy(pos).float().unsqueeze(0).to(latent.device)
        else:
            pos = self.pos_embed
        return (latent + pos).to(latent.dtype)


def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False, downscale_freq_shift=1.0, scale=1.0,
                           max_period=10000):
    half = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - downscale_freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift, scale=1):
        super().__init__()
        self.num_channels, self.flip, self.shift, self.scale = num_channels, flip_sin_to_cos, downscale_freq_shift, scale

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels, self.flip, self.shift, self.scale)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu"):
        super().__init__()
        assert act_fn == "silu"
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


class PixArtAlphaCombinedTimestepSizeEmbeddings(nn.Module):
    def __init__(self, embedding_dim, size_emb_dim, use_additional_conditions: bool = False):
        super().__init__()
        assert not use_additional_conditions, "Latte-1 (sample_size 64) does not use the size/aspect-ratio conditions"
        self.time_proj = Timesteps(num_channels=256, flip_sin_to_cos=True, downscale_freq_shift=0)
        self.timestep_embedder = TimestepEmbedding(in_channels=256, time_embed_dim=embedding_dim)

    def forward(self, timestep, resolution, aspect_ratio, batch_size, hidden_dtype):
        proj = self.time_proj(timestep)
        return self.timestep_embedder(proj.to(dtype=hidden_dtype))


class PixArtAlphaTextProjection(nn.Module):
    def __init__(self, in_features, hidden_size, out_features=None, act_fn="gelu_tanh"):
        super().__init__()
        out_features = out_features or hidden_size
        self.linear_1 = nn.Linear(in_features, hidden_size, bias=True)
        self.act_1 = nn.GELU(approximate="tanh")
        self.linear_2 = nn.Linear(hidden_size, out_features, bias=True)

    def forward(self, caption):
        return self.linear_2(self.act_1(self.linear_1(caption)))


class FeedForward(nn.Module):
    """diffusers.models.attention.FeedForward (0.30.0) for activation_fn="gelu-approximate"."""

    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False, inner_dim=None,
                 bias=True):
        super().__init__()
        assert activation_fn == "gelu-approximate"
        inner_dim = int(dim * mult) if inner_dim is None else inner_dim
        dim_out = dim_out if dim_out is not None else dim
        self.net = nn.ModuleList([GELU(dim, inner_dim, approximate="tanh", bias=bias), nn.Dropout(dropout),
                                  nn.Linear(inner_dim, dim_out, bias=bias)])
        if final_dropout:
            self.net.append(nn.Dropout(dropout))

    def forward(self, hidden_states, *args, **kwargs):
        for m in self.net:
            hidden_states = m(hidden_states)
        return hidden_states


def get_3d_sincos_pos_embed(embed_dim, spatial_size, temporal_size, spatial_interpolation_scale=1.0,
                            temporal_interpolation_scale=1.0):
    """diffusers.models.embeddings.get_3d_sincos_pos_embed (0.30.0): [T, H*W, D] = [temporal D/4 | spatial 3D/4]."""
    if isinstance(spatial_size, int):
        spatial_size = (spatial_size, spatial_size)
    ds, dt = 3 * embed_dim // 4, embed_dim // 4
    grid_h = np.arange(spatial_size[1], dtype=np.float32) / spatial_interpolation_scale
    grid_w = np.arange(spatial_size[0], dtype=np.float32) / spatial_interpolation_scale
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape([2, 1, spatial_size[1], spatial_size[0]])
    pos_s = get_2d_sincos_pos_embed_from_grid(ds, grid)
    grid_t = np.arange(temporal_size, dtype=np.float32) / temporal_interpolation_scale
    pos_t = get_1d_sincos_pos_embed_from_grid(dt, grid_t)
    pos_s = np.repeat(pos_s[np.newaxis], temporal_size, axis=0)
    pos_t = np.repeat(pos_t[:, np.newaxis], spatial_size[0] * spatial_size[1], axis=1)
    return np.concatenate([pos_t, pos_s], axis=-1)


class SchedulerMixin:
    pass


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    """diffusers.utils.torch_utils.randn_tensor (third-party, restated from its published behaviour): standard normal noise of
    ``shape``; a CPU generator draws on the CPU and the result moves to ``device`` (so a seed gives the same noise on every device); a
    list of generators draws one batch element each."""
    device = torch.device(device) if device is not None else torch.device("cpu")
    layout = layout or torch.strided
    rand_device = device
    if generator is not None:
        gtype = (generator[0] if isinstance(generator, (list, tuple)) else generator).device.type
        if gtype != device.type and gtype == "cpu":
            rand_device = torch.device("cpu")
        elif gtype != device.type and gtype == "cuda":
            raise ValueError(f"Cannot generate a {device} tensor from a generator of type {gtype}.")
    if isinstance(generator, (list, tuple)) and len(generator) == 1:
        generator = generator[0]
    if isinstance(generator, (list, tuple)):
        one = (1,) + tuple(shape[1:])
        lat = torch.cat([torch.randn(one, generator=generator[i], device=rand_device, dtype=dtype, layout=layout)
                         for i in range(shape[0])], dim=0)
        return lat.to(device)
    return torch.randn(tuple(shape), generator=generator, device=rand_device, dtype=dtype, layout=layout).to(device)


class _Unused(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("diffusers class outside the Latte ada_norm_single path")


# ----------------------------------------------------------------------------------------------- DDIM scheduler
class DDIMScheduler:
    """diffusers.schedulers.DDIMScheduler, the subset pipeline_latte.py uses: linear betas, timestep_spacing "leading",
    steps_offset 0, set_alpha_to_one True, epsilon prediction, eta 0, clip_sample False (latent diffusion; the value
    lives in the HF repo's scheduler_config.json, not in the reference tree)."""

    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 clip_sample=False, set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon", **unused):
        assert beta_schedule == "linear" and prediction_type == "epsilon"
        self.num_train_timesteps = num_train_timesteps
        self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.steps_offset = steps_offset
        self.clip_sample = clip_sample
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        step_ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
        ts += self.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def step(self, model_output, timestep, sample, eta=0.0, return_dict=False, **unused):
        t = int(timestep)
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        beta_t = 1 - a_t
        x0 = (sample - beta_t**0.5 * model_output) / a_t**0.5
        if self.clip_sample:
            x0 = x0.clamp(-1.0, 1.0)
        pred_dir = (1 - a_prev) ** 0.5 * model_output
        prev = a_prev**0.5 * x0 + pred_dir
        return (prev,)



# ----------------------------------------------------------------------------------------------- AutoencoderKL (decode side)
class _VaeResnetBlock2D(nn.Module):
    """diffusers.models.resnet.ResnetBlock2D with temb_channels=None, time_embedding_norm="default", output_scale_factor=1:
    norm1 -> SiLU -> conv1 -> norm2 -> SiLU -> dropout(0) -> conv2; 1x1 ``conv_shortcut`` when the channel count changes."""

    def __init__(self, in_channels, out_channels, groups=32, eps=1e-6):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, stride=1, padding=1)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, stride=1, padding=1)
        self.nonlinearity = nn.SiLU()
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, x, temb=None):
        h = self.conv1(self.nonlinearity(self.norm1(x)))
        h = self.conv2(self.nonlinearity(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class _VaeAttention(nn.Module):
    """diffusers Attention as UNetMidBlock2D builds it for the VAE (heads = C // attention_head_dim = 1, norm_num_groups=32,
    eps=1e-6, residual_connection=True, bias=True, rescale_output_factor=1) run by AttnProcessor2_0 on a 4-D input."""

    def __init__(self, channels, groups=32, eps=1e-6):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, channels, eps=eps, affine=True)
        self.to_q = nn.Linear(channels, channels, bias=True)
        self.to_k = nn.Linear(channels, channels, bias=True)
        self.to_v = nn.Linear(channels, channels, bias=True)
        self.to_out = nn.ModuleList([nn.Linear(channels, channels, bias=True), nn.Dropout(0.0)])

    def forward(self, x, temb=None):
        b, c, h, w = x.shape
        res = x
        t = x.view(b, c, h * w).transpose(1, 2)
        t = self.group_norm(t.transpose(1, 2)).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        o = F.scaled_dot_product_attention(q.unsqueeze(1), k.unsqueeze(1), v.unsqueeze(1)).squeeze(1)
        o = self.to_out[0](o)
        return o.transpose(-1, -2).reshape(b, c, h, w) + res


class _VaeMidBlock(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.resnets = nn.ModuleList([_VaeResnetBlock2D(channels, channels), _VaeResnetBlock2D(channels, channels)])
        self.attentions = nn.ModuleList([_VaeAttention(channels)])

    def forward(self, x):
        x = self.resnets[0](x)
        x = self.attentions[0](x)
        return self.resnets[1](x)


class _VaeUpsample2D(nn.Module):
    """Upsample2D(use_conv=True): F.interpolate(scale_factor=2, mode="nearest") then a 3x3 conv."""

    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _VaeUpDecoderBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, num_layers, add_upsample):
        super().__init__()
        self.resnets = nn.ModuleList([_VaeResnetBlock2D(in_channels if i == 0 else out_channels, out_channels)
                                      for i in range(num_layers)])
        self.upsamplers = nn.ModuleList([_VaeUpsample2D(out_channels)]) if add_upsample else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class _VaeDecoder(nn.Module):
    """diffusers.models.autoencoders.vae.Decoder: conv_in -> mid_block -> up_blocks (reversed block_out_channels,
    layers_per_block + 1 resnets each, upsample on all but the last) -> conv_norm_out -> SiLU -> conv_out."""

    def __init__(self, in_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2):
        super().__init__()
        rev = list(reversed(block_out_channels))
        self.conv_in = nn.Conv2d(in_channels, rev[0], 3, padding=1)
        self.mid_block = _VaeMidBlock(rev[0])
        self.up_blocks = nn.ModuleList()
        out_ch = rev[0]
        for i, ch in enumerate(rev):
            prev, out_ch = out_ch, ch
            self.up_blocks.append(_VaeUpDecoderBlock2D(prev, out_ch, layers_per_block + 1, i != len(rev) - 1))
        self.conv_norm_out = nn.GroupNorm(32, block_out_channels[0], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(block_out_channels[0], out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(self.conv_act(self.conv_norm_out(x)))


class _VaeDownsample2D(nn.Module):
    """Downsample2D(use_conv=True, padding=0): F.pad(x, (0, 1, 0, 1)) (right / bottom only) then a 3x3 stride-2 conv."""

    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))


class _VaeDownEncoderBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, num_layers, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList([_VaeResnetBlock2D(in_channels if i == 0 else out_channels, out_channels)
                                      for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([_VaeDownsample2D(out_channels)]) if add_downsample else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        return x


class _VaeEncoder(nn.Module):
    """diffusers.models.autoencoders.vae.Encoder (double_z=True): conv_in -> down_blocks (block_out_channels, layers_per_block
    resnets each, downsample on all but the last) -> mid_block -> conv_norm_out -> SiLU -> conv_out (2 * latent channels)."""

    def __init__(self, in_channels=3, out_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2):
        super().__init__()
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        out_ch = block_out_channels[0]
        for i, ch in enumerate(block_out_channels):
            prev, out_ch = out_ch, ch
            self.down_blocks.append(_VaeDownEncoderBlock2D(prev, out_ch, layers_per_block, i != len(block_out_channels) - 1))
        self.mid_block = _VaeMidBlock(block_out_channels[-1])
        self.conv_norm_out = nn.GroupNorm(32, block_out_channels[-1], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(block_out_channels[-1], 2 * out_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        return self.conv_out(self.conv_act(self.conv_norm_out(self.mid_block(x))))


class _DiagonalGaussian:
    """diffusers.models.autoencoders.vae.DiagonalGaussianDistribution: moments = [mean | logvar] along dim 1, logvar clamped to
    [-30, 20], sample() = mean + exp(0.5 logvar) * randn."""

    def __init__(self, parameters):
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None):
        return self.mean + self.std * torch.randn(self.mean.shape, generator=generator, dtype=self.mean.dtype)

    def mode(self):
        return self.mean


class AutoencoderKL(nn.Module):
    """diffusers.models.AutoencoderKL at the SDXL-VAE config the reference loads
    ("PixArt-alpha/pixart_sigma_sdxlvae_T5_diffusers", subfolder "vae": block_out_channels (128, 256, 512, 512),
    layers_per_block 2, latent_channels 4, norm_num_groups 32): decode(z) = decoder(post_quant_conv(z)); encode(x).latent_dist =
    DiagonalGaussian(quant_conv(encoder(x))).  ``from_pretrained`` returns randomly initialised weights (no network here)."""

    def __init__(self, block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4):
        super().__init__()
        self.config = SimpleNamespace(latent_channels=latent_channels, block_out_channels=tuple(block_out_channels),
                                      layers_per_block=layers_per_block, scaling_factor=0.13025)
        self.encoder = _VaeEncoder(3, latent_channels, block_out_channels, layers_per_block)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.decoder = _VaeDecoder(latent_channels, 3, block_out_channels, layers_per_block)

    @classmethod
    def from_pretrained(cls, *args, **kwargs):
        return cls()

    def encode(self, x, return_dict=True):
        return SimpleNamespace(latent_dist=_DiagonalGaussian(self.quant_conv(self.encoder(x))))

    def decode(self, z, return_dict=True):
        return SimpleNamespace(sample=self.decoder(self.post_quant_conv(z)))


def get_activation(name):
    """diffusers.models.activations.get_activation for the names the reference uses."""
    return {"swish": nn.SiLU, "silu": nn.SiLU, "mish": nn.Mish, "gelu": nn.GELU, "relu": nn.ReLU}[name.lower()]()


class DecoderOutput:
    """diffusers.models.autoencoders.vae.DecoderOutput (a dataclass with one field)."""

    def __init__(self, sample):
        self.sample = sample


# ----------------------------------------------------------------------------------------------- installation
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    """Install (or extend) the fake ``diffusers`` package in sys.modules."""
    d = sys.modules.get("diffusers") or _mod("diffusers")
    d.__dict__.setdefault("__path__", [])
    _mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)
    models = sys.modules.get("diffusers.models") or _mod("diffusers.models")
    models.__dict__.setdefault("__path__", [])
    models.__dict__["AutoencoderKL"] = AutoencoderKL
    models.__dict__["AutoencoderKLTemporalDecoder"] = _Unused
    _mod("diffusers.models.activations", GEGLU=GEGLU, GELU=GELU, ApproximateGELU=ApproximateGELU)
    ap = _mod("diffusers.models.attention_processor", Attention=Attention, AttnProcessor=object)
    sys.modules["diffusers.models.attention"] = _mod("diffusers.models.attention", Attention=Attention, FeedForward=FeedForward)
    _mod("diffusers.models.modeling_outputs", Transformer2DModelOutput=BaseOutput)
    _mod("diffusers.schedulers.scheduling_utils", KarrasDiffusionSchedulers=[], SchedulerMixin=SchedulerMixin)
    _mod("diffusers.models.embeddings", ImagePositionalEmbeddings=_Unused, PatchEmbed=PatchEmbed,
         PixArtAlphaCombinedTimestepSizeEmbeddings=PixArtAlphaCombinedTimestepSizeEmbeddings,
         PixArtAlphaTextProjection=PixArtAlphaTextProjection, SinusoidalPositionalEmbedding=_Unused,
         get_1d_sincos_pos_embed_from_grid=get_1d_sincos_pos_embed_from_grid, Timesteps=Timesteps,
         TimestepEmbedding=TimestepEmbedding, get_2d_sincos_pos_embed=get_2d_sincos_pos_embed,
         get_3d_sincos_pos_embed=get_3d_sincos_pos_embed)
    _mod("diffusers.models.lora", LoRACompatibleConv=LoRACompatibleConv, LoRACompatibleLinear=LoRACompatibleLinear)
    _mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    _mod("diffusers.models.normalization", AdaLayerNorm=_Unused, AdaLayerNormContinuous=_Unused, AdaLayerNormZero=_Unused)
    _mod("diffusers.utils", USE_PEFT_BACKEND=False, BaseOutput=BaseOutput, deprecate=deprecate,
         is_torch_version=lambda *a, **k: True)
    _mod("diffusers.utils.torch_utils", maybe_allow_in_graph=maybe_allow_in_graph, randn_tensor=randn_tensor)
    # what videosys/models/autoencoders/autoencoder_kl_cogvideox.py imports (:18-24): no arithmetic except get_activation
    loaders = sys.modules.get("diffusers.loaders") or _mod("diffusers.loaders")
    loaders.__dict__.setdefault("__path__", [])
    _mod("diffusers.loaders.single_file_model", FromOriginalModelMixin=type("FromOriginalModelMixin", (), {}))
    sys.modules["diffusers.models.activations"].__dict__["get_activation"] = get_activation
    ae = sys.modules.get("diffusers.models.autoencoders") or _mod("diffusers.models.autoencoders")
    ae.__dict__.setdefault("__path__", [])
    _mod("diffusers.models.autoencoders.vae", DecoderOutput=DecoderOutput, DiagonalGaussianDistribution=_Unused)
    sys.modules["diffusers.models.modeling_outputs"].__dict__["AutoencoderKLOutput"] = BaseOutput
    utils = sys.modules["diffusers.utils"]
    utils.__dict__.setdefault("__path__", [])
    _mod("diffusers.utils.accelerate_utils", apply_forward_hook=lambda f: f)
    sch = _mod("diffusers.schedulers", DDIMScheduler=DDIMScheduler)
    sch.__dict__["__path__"] = []
    return ap
