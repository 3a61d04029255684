"""TEST INFRASTRUCTURE ONLY — fp32 CPU restatement of the Latte denoise path.

Follows /root/reference/videosys/models/transformers/latte_transformer_3d.py (LatteT2V.forward :1144-1466,
BasicTransformerBlock.forward :357-517, BasicTransformerBlock_.forward :680-824, AdaLayerNormSingle :846-878) and the
published diffusers==0.30.0 semantics of the leaf modules it imports (Attention + AttnProcessor2_0, GELU-tanh
FeedForward, PatchEmbed with sincos position table, PixArt-alpha timestep / caption embedders, DDIMScheduler) —
see oracle/diffusers_stub.py for the leaf-by-leaf citations.  Pinned by tests/test_oracle_vs_golden.py against
goldens minted from the reference's own LatteT2V class running over those restated leaves (oracle/make_golden.py);
the leaves themselves have no reference-side test: parity unpinned for them.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ------------------------------------------------------------------------------------------------- leaves
def linear(x: Tensor, sd: Dict[str, Tensor], prefix: str) -> Tensor:
    return F.linear(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"))


def layer_norm(x: Tensor, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), None, None, eps)


def sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    """diffusers get_1d_sincos_pos_embed_from_grid: [sin | cos], float64 frequencies."""
    omega = np.arange(embed_dim // 2, dtype=np.float64) / (embed_dim / 2.0)
    omega = 1.0 / 10000**omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_2d(embed_dim: int, gh: int, gw: int, base_size: int, interpolation_scale: float = 1.0) -> Tensor:
    """diffusers get_2d_sincos_pos_embed (note: the half named emb_h is built from the W coordinate — meshgrid(w, h))."""
    grid_h = np.arange(gh, dtype=np.float32) / (gh / base_size) / interpolation_scale
    grid_w = np.arange(gw, dtype=np.float32) / (gw / base_size) / interpolation_scale
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape([2, 1, gw, gh])
    emb = np.concatenate([sincos_1d(embed_dim // 2, grid[0]), sincos_1d(embed_dim // 2, grid[1])], axis=1)
    return torch.from_numpy(emb).float()


def timestep_embedding(t: Tensor, dim: int = 256) -> Tensor:
    """Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    a = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)


def attention(x: Tensor, ctx: Tensor, sd, prefix: str, heads: int, bias_mask: Optional[Tensor] = None) -> Tensor:
    """diffusers Attention / AttnProcessor2_0; bias_mask [B, 1, Lk] additive (0 keep / -10000 discard)."""
    B, Lq, C = x.shape
    Lk = ctx.shape[1]
    D = C // heads
    q = linear(x, sd, prefix + ".to_q").view(B, Lq, heads, D).transpose(1, 2)
    k = linear(ctx, sd, prefix + ".to_k").view(B, Lk, heads, D).transpose(1, 2)
    v = linear(ctx, sd, prefix + ".to_v").view(B, Lk, heads, D).transpose(1, 2)
    if x.dtype != torch.float32:  # low-precision run: AttnProcessor2_0 calls F.scaled_dot_product_attention
        am = None if bias_mask is None else bias_mask[:, None].to(q.dtype).expand(B, 1, Lq, Lk)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=am).transpose(1, 2).reshape(B, Lq, C)
        return linear(o, sd, prefix + ".to_out.0")
    s = q @ k.transpose(-1, -2) / math.sqrt(D)
    if bias_mask is not None:
        s = s + bias_mask[:, None]
    o = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, Lq, C)
    return linear(o, sd, prefix + ".to_out.0")


def feed_forward(x: Tensor, sd, prefix: str) -> Tensor:
    """FeedForward(activation_fn="gelu-approximate"): net.0 = GELU(tanh)(Linear), net.2 = Linear."""
    return linear(F.gelu(linear(x, sd, prefix + ".net.0.proj"), approximate="tanh"), sd, prefix + ".net.2")


# ------------------------------------------------------------------------------------------------- blocks
def _chunks6(sd, prefix, timestep6: Tensor):
    B = timestep6.shape[0]
    return (sd[prefix + ".scale_shift_table"][None] + timestep6.reshape(B, 6, -1)).chunk(6, dim=1)


def spatial_block(x, sd, prefix, heads, timestep6, ctx, ctx_bias, eps):
    """BasicTransformerBlock.forward, ada_norm_single branch (latte_transformer_3d.py:398-404,413-417,431-432,
    440-469,500-505,510,522)."""
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = _chunks6(sd, prefix, timestep6)
    h = layer_norm(x, eps) * (1 + scale_msa) + shift_msa
    x = gate_msa * attention(h, h, sd, prefix + ".attn1", heads) + x
    x = attention(x, ctx, sd, prefix + ".attn2", heads, ctx_bias) + x  # no norm2 here for ada_norm_single (:447-450)
    h = layer_norm(x, eps) * (1 + scale_mlp) + shift_mlp
    return gate_mlp * feed_forward(h, sd, prefix + ".ff") + x


def temporal_block(x, sd, prefix, heads, timestep6, eps):
    """BasicTransformerBlock_.forward, ada_norm_single branch (:721-727,738-753,759,786-789,803,808,819)."""
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = _chunks6(sd, prefix, timestep6)
    h = layer_norm(x, eps) * (1 + scale_msa) + shift_msa
    x = gate_msa * attention(h, h, sd, prefix + ".attn1", heads) + x
    h = layer_norm(x, eps) * (1 + scale_mlp) + shift_mlp
    return gate_mlp * feed_forward(h, sd, prefix + ".ff") + x


class LatteOracle:
    def __init__(self, sd: Dict[str, Tensor], num_layers: int, num_heads: int, head_dim: int, patch_size: int = 2,
                 sample_size: int = 64, out_channels: int = 8, video_length: int = 16, norm_eps: float = 1e-6,
                 device=None, dtype=torch.float32):
        # device / dtype: the full-depth parity tests run this oracle on the GPU as the checker (fp32) and once more in the
        # low-precision dtype the reference runs in (its distance from fp32 = the noise floor); see stdit3_oracle.py header
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.dtype = dtype
        self.sd = {k: v.to(device=self.device, dtype=dtype) for k, v in sd.items()}
        self.L, self.H, self.D = num_layers, num_heads, head_dim
        self.C = num_heads * head_dim
        self.p = patch_size
        self.sample_size = sample_size
        self.out_channels = out_channels
        self.video_length = video_length
        self.eps = norm_eps

    def forward(self, hidden_states: Tensor, timestep: Tensor, encoder_hidden_states: Tensor,
                encoder_attention_mask: Optional[Tensor] = None, enable_temporal_attentions: bool = True,
                on_hidden=None) -> Tensor:
        sd, C, p = self.sd, self.C, self.p
        dev, dt = self.device, self.dtype
        hidden_states, timestep, encoder_hidden_states = hidden_states.to(dev), timestep.to(dev), encoder_hidden_states.to(dev)
        B, cin, Fr, Hh, Ww = hidden_states.shape
        x = hidden_states.to(dt).permute(0, 2, 1, 3, 4).reshape(B * Fr, cin, Hh, Ww)  # b c f h w -> (b f) c h w (:1219)
        bias = None
        if encoder_attention_mask is not None:  # :1245-1248
            bias = ((1 - encoder_attention_mask.to(dev).to(dt)) * -10000.0).unsqueeze(1)
            bias = bias.repeat_interleave(Fr, dim=0)
        # PatchEmbed (:1266): conv patchify + sincos table for this grid
        gh, gw = Hh // p, Ww // p
        S = gh * gw
        x = F.conv2d(x, sd["pos_embed.proj.weight"], sd["pos_embed.proj.bias"], stride=p).flatten(2).transpose(1, 2)
        interp = max(self.sample_size // 64, 1)
        x = x + sincos_2d(C, gh, gw, self.sample_size // p, interp)[None].to(device=dev, dtype=dt)
        # AdaLayerNormSingle (:846-878): embedded = MLP(sinusoid(t)); timestep6 = Linear(SiLU(embedded))
        emb = linear(F.silu(linear(timestep_embedding(timestep.float()).to(dt), sd, "adaln_single.emb.timestep_embedder.linear_1")),
                     sd, "adaln_single.emb.timestep_embedder.linear_2")
        t6 = linear(F.silu(emb), sd, "adaln_single.linear")
        # caption projection (:1284) then one copy per frame (:1297-1299)
        y = linear(F.gelu(linear(encoder_hidden_states.to(dt), sd, "caption_projection.linear_1"), approximate="tanh"),
                   sd, "caption_projection.linear_2")
        y_sp = y.repeat_interleave(Fr, dim=0)
        t_sp = t6.repeat_interleave(Fr, dim=0)  # (b f) d
        t_tp = t6.repeat_interleave(S, dim=0)  # (b p) d
        tpe = torch.from_numpy(sincos_1d(C, np.arange(0, self.video_length)[:, None].astype(np.float64))).float()[None].to(
            device=dev, dtype=dt)
        for i in range(self.L):
            x = spatial_block(x, sd, f"transformer_blocks.{i}", self.H, t_sp, y_sp, bias, self.eps)
            if enable_temporal_attentions:
                x = x.view(B, Fr, S, C).permute(0, 2, 1, 3).reshape(B * S, Fr, C)  # (b f) t d -> (b t) f d (:1391)
                if i == 0 and Fr > 1:
                    x = x + tpe[:, :Fr]
                x = temporal_block(x, sd, f"temporal_transformer_blocks.{i}", self.H, t_tp, self.eps)
                x = x.view(B, S, Fr, C).permute(0, 2, 1, 3).reshape(B * Fr, S, C)
            if on_hidden is not None:
                on_hidden(i, x)   # [(b f), S, C] after block pair i
        # final (:1443-1449) + unpatchify (:1452-1460)
        e = emb.repeat_interleave(Fr, dim=0)
        shift, scale = (sd["scale_shift_table"][None] + e[:, None]).chunk(2, dim=1)
        x = layer_norm(x, 1e-6) * (1 + scale) + shift
        x = linear(x, sd, "proj_out")
        co = self.out_channels
        x = x.reshape(-1, gh, gw, p, p, co)
        x = torch.einsum("nhwpqc->nchpwq", x).reshape(-1, co, gh * p, gw * p)
        return x.view(B, Fr, co, gh * p, gw * p).permute(0, 2, 1, 3, 4).contiguous().float()

    __call__ = forward


# ------------------------------------------------------------------------------------------------- DDIM sampling
def ddim_tables(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02):
    betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    return torch.cumprod(1.0 - betas, dim=0)


def ddim_timesteps(num_inference_steps: int, num_train_timesteps: int = 1000) -> List[int]:
    ratio = num_train_timesteps // num_inference_steps
    return [int(v) for v in (np.arange(0, num_inference_steps) * ratio).round()[::-1]]


def ddim_coeffs(t: int, num_inference_steps: int, alphas_cumprod: Tensor, num_train_timesteps: int = 1000):
    """prev = c_z * sample + c_eps * eps  (DDIMScheduler.step, eta = 0, epsilon prediction, no clipping)."""
    prev_t = t - num_train_timesteps // num_inference_steps
    a_t = float(alphas_cumprod[t])
    a_prev = float(alphas_cumprod[prev_t]) if prev_t >= 0 else 1.0
    c_z = math.sqrt(a_prev / a_t)
    c_eps = math.sqrt(1 - a_prev) - math.sqrt(a_prev * (1 - a_t) / a_t)
    return c_z, c_eps


def latte_sample(model, latents: Tensor, prompt_embeds: Tensor, negative_embeds: Tensor, prompt_mask: Optional[Tensor],
                 negative_mask: Optional[Tensor], num_inference_steps: int = 20, guidance_scale: float = 7.5) -> Tensor:
    """The denoising loop of LattePipeline.generate (pipeline_latte.py:798-876): CFG batch [negative | prompt],
    learned-sigma half dropped, DDIM eta = 0."""
    ac = ddim_tables()
    z = latents.float().clone()
    emb = torch.cat([negative_embeds, prompt_embeds], dim=0)
    mask = None if prompt_mask is None else torch.cat([negative_mask, prompt_mask], dim=0)
    for t in ddim_timesteps(num_inference_steps):
        zin = torch.cat([z, z], dim=0)
        tt = torch.full((zin.shape[0],), t, dtype=torch.int64)
        out = model(zin, tt, emb, mask)
        unc, txt = out.chunk(2)
        eps = (unc + guidance_scale * (txt - unc))[:, : z.shape[1]]
        c_z, c_eps = ddim_coeffs(t, num_inference_steps, ac)
        z = c_z * z + c_eps * eps
    return z


def synth_state_dict(num_layers, num_heads, head_dim, caption_channels=4096, in_channels=4, out_channels=8, patch_size=2,
                     seed: int = 4321) -> Dict[str, Tensor]:
    """Seeded random weights with the HF checkpoint's (maxin-cn/Latte-1 transformer) key names and shapes."""
    g = torch.Generator().manual_seed(seed)
    C = num_heads * head_dim
    sd: Dict[str, Tensor] = {}

    def lin(name, n_out, n_in):
        s = min(0.08, 1.0 / math.sqrt(n_in))
        sd[name + ".weight"] = torch.randn(n_out, n_in, generator=g) * s
        sd[name + ".bias"] = torch.randn(n_out, generator=g) * 0.02

    sd["pos_embed.proj.weight"] = torch.randn(C, in_channels, patch_size, patch_size, generator=g) * 0.1
    sd["pos_embed.proj.bias"] = torch.randn(C, generator=g) * 0.02
    lin("adaln_single.emb.timestep_embedder.linear_1", C, 256)
    lin("adaln_single.emb.timestep_embedder.linear_2", C, C)
    lin("adaln_single.linear", 6 * C, C)
    lin("caption_projection.linear_1", C, caption_channels)
    lin("caption_projection.linear_2", C, C)
    for kind, cross in (("transformer_blocks", True), ("temporal_transformer_blocks", False)):
        for i in range(num_layers):
            p = f"{kind}.{i}"
            sd[p + ".scale_shift_table"] = torch.randn(6, C, generator=g) / C**0.5
            for a in (("attn1", "attn2") if cross else ("attn1",)):
                for l in ("to_q", "to_k", "to_v", "to_out.0"):
                    lin(f"{p}.{a}.{l}", C, C)
            lin(p + ".ff.net.0.proj", 4 * C, C)
            lin(p + ".ff.net.2", C, 4 * C)
    sd["scale_shift_table"] = torch.randn(2, C, generator=g) / C**0.5
    lin("proj_out", patch_size * patch_size * out_channels, C)
    return sd
