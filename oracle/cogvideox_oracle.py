"""TEST INFRASTRUCTURE ONLY — fp32 CPU restatement of the CogVideoX denoise path.

Follows /root/reference/videosys: models/transformers/cogvideox_transformer_3d.py (CogVideoXAttnProcessor2_0 :35-175,
CogVideoXBlock :179-312, CogVideoXTransformer3DModel.forward :479-589), models/modules/normalization.py
(CogVideoXLayerNormZero :36-60, AdaLayerNorm :62-114), models/modules/embeddings.py (CogVideoXPatchEmbed :14-51,
get_3d_rotary_pos_embed :283-355, apply_rotary_emb :358-412), schedulers/scheduling_ddim_cogvideox.py (:87-115,
:172-216, :258-297, :299-393) and pipelines/cogvideox/pipeline_cogvideox.py (:449-474, :675-720, :757-775); the
diffusers==0.30.0 leaves (Attention with LayerNorm qk-norm, FeedForward gelu-approximate, Timesteps,
TimestepEmbedding, get_3d_sincos_pos_embed) are restated as in oracle/diffusers_stub.py.  Pinned by
tests/test_cogvideox_cpu.py against goldens minted from the reference's own classes (oracle/make_golden_cogvideox.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def linear(x: Tensor, sd: Dict[str, Tensor], prefix: str) -> Tensor:
    return F.linear(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"))


def ln(x: Tensor, sd, prefix: str, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def timestep_embedding(t: Tensor, dim: int) -> Tensor:
    """Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    a = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)


def sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    omega = np.arange(embed_dim // 2, dtype=np.float64) / (embed_dim / 2.0)
    omega = 1.0 / 10000**omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_3d(embed_dim: int, w: int, h: int, t: int, spatial_scale: float, temporal_scale: float) -> Tensor:
    """diffusers get_3d_sincos_pos_embed(embed_dim, (w, h), t, ...) flattened to [t*h*w, embed_dim]."""
    ds, dt = 3 * embed_dim // 4, embed_dim // 4
    gh = np.arange(h, dtype=np.float32) / spatial_scale
    gw = np.arange(w, dtype=np.float32) / spatial_scale
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape([2, 1, h, w])
    pos_s = np.concatenate([sincos_1d(ds // 2, grid[0]), sincos_1d(ds // 2, grid[1])], axis=1)
    pos_t = sincos_1d(dt, np.arange(t, dtype=np.float32) / temporal_scale)
    pos_s = np.repeat(pos_s[np.newaxis], t, axis=0)
    pos_t = np.repeat(pos_t[:, np.newaxis], w * h, axis=1)
    return torch.from_numpy(np.concatenate([pos_t, pos_s], axis=-1)).float().flatten(0, 1)


def crop_region(src, tgt_width, tgt_height):
    """pipeline_cogvideox.py:757-775."""
    h, w = src
    if h / w > tgt_height / tgt_width:
        rh, rw = tgt_height, int(round(tgt_height / h * w))
    else:
        rw, rh = tgt_width, int(round(tgt_width / w * h))
    top, left = int(round((tgt_height - rh) / 2.0)), int(round((tgt_width - rw) / 2.0))
    return (top, left), (top + rh, left + rw)


def rope_3d(head_dim: int, crops, grid_size, temporal_size: int, theta: float = 10000.0) -> Tuple[Tensor, Tensor]:
    """modules/embeddings.py:283-355 (use_real): cos, sin [T*H*W, head_dim], pairs repeat-interleaved."""
    (s0, s1), (e0, e1) = crops
    gh = torch.from_numpy(np.linspace(s0, e0, grid_size[0], endpoint=False, dtype=np.float32))
    gw = torch.from_numpy(np.linspace(s1, e1, grid_size[1], endpoint=False, dtype=np.float32))
    gt = torch.from_numpy(np.linspace(0, temporal_size, temporal_size, endpoint=False, dtype=np.float32))
    dt, dh, dw = head_dim // 4, head_dim // 8 * 3, head_dim // 8 * 3

    def axis(grid, d):
        f = 1.0 / (theta ** (torch.arange(0, d, 2).float() / d))
        return torch.einsum("n,f->nf", grid, f).repeat_interleave(2, dim=-1)

    ft, fh, fw = axis(gt, dt), axis(gh, dh), axis(gw, dw)
    T, H, W = temporal_size, grid_size[0], grid_size[1]
    freqs = torch.cat([ft[:, None, None, :].expand(T, H, W, dt), fh[None, :, None, :].expand(T, H, W, dh),
                       fw[None, None, :, :].expand(T, H, W, dw)], dim=-1).reshape(T * H * W, -1)
    return freqs.cos(), freqs.sin()


def prepare_rope(height: int, width: int, num_frames: int, head_dim: int, patch: int = 2, vae_scale: int = 8):
    """pipeline_cogvideox.py:449-474 (latent frames = num_frames)."""
    gh, gw = height // (vae_scale * patch), width // (vae_scale * patch)
    crops = crop_region((gh, gw), 720 // (vae_scale * patch), 480 // (vae_scale * patch))
    return rope_3d(head_dim, crops, (gh, gw), num_frames)


def apply_rope(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(-2)
    return x * cos + rot * sin


class CogVideoXOracle:
    def __init__(self, sd: Dict[str, Tensor], num_layers: int, num_heads: int, head_dim: int = 64, patch_size: int = 2,
                 out_channels: int = 16, max_text_seq_length: int = 226, sample_width: int = 90, sample_height: int = 60,
                 sample_frames: int = 49, temporal_compression_ratio: int = 4, spatial_interpolation_scale: float = 1.875,
                 temporal_interpolation_scale: float = 1.0, use_rotary_positional_embeddings: bool = False,
                 norm_eps: float = 1e-5, device=None, dtype: torch.dtype = torch.float32):
        """device / dtype: the checker may run on the GPU; dtype = bfloat16 executes the same module graph with torch's bf16
        kernels, i.e. the way the reference itself runs the model (its distance from the fp32 run is the reference's own noise
        floor — tests/fulldepth_util.py)."""
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.dtype = dtype
        self.sd = {k: v.to(device=self.device, dtype=dtype) for k, v in sd.items()}
        self.L, self.H, self.D, self.C = num_layers, num_heads, head_dim, num_heads * head_dim
        self.p, self.co = patch_size, out_channels
        self.max_text = max_text_seq_length
        self.use_rope = use_rotary_positional_embeddings
        self.eps = norm_eps
        pf = (sample_frames - 1) // temporal_compression_ratio + 1
        self.pos3d = sincos_3d(self.C, sample_width // patch_size, sample_height // patch_size, pf,
                               spatial_interpolation_scale, temporal_interpolation_scale).to(device=self.device, dtype=dtype)

    def attn(self, x: Tensor, prefix: str, text_len: int, rope) -> Tensor:
        """CogVideoXAttnProcessor2_0 on the joint sequence x [B, Lt + Lv, C]."""
        sd, H, D = self.sd, self.H, self.D
        B, L, C = x.shape
        q, k, v = [linear(x, sd, f"{prefix}.{n}").view(B, L, H, D).transpose(1, 2) for n in ("to_q", "to_k", "to_v")]
        q = F.layer_norm(q, (D,), sd[prefix + ".norm_q.weight"], sd[prefix + ".norm_q.bias"], 1e-6)
        k = F.layer_norm(k, (D,), sd[prefix + ".norm_k.weight"], sd[prefix + ".norm_k.bias"], 1e-6)
        if rope is not None:
            cos, sin = rope
            n = cos.shape[0]
            q = torch.cat([q[:, :, :text_len], apply_rope(q[:, :, text_len:text_len + n], cos, sin), q[:, :, text_len + n:]], 2)
            k = torch.cat([k[:, :, :text_len], apply_rope(k[:, :, text_len:text_len + n], cos, sin), k[:, :, text_len + n:]], 2)
        if L > 4096 or self.dtype != torch.float32:
            # long sequences (config 5: 17 776 rows) / low precision: F.scaled_dot_product_attention, which is what the reference's
            # CogVideoXAttnProcessor2_0 calls (same softmax(q k^T / sqrt(D)) v; no [B, H, L, L] score tensor)
            o = torch.cat([F.scaled_dot_product_attention(q[:, h0:h0 + 4], k[:, h0:h0 + 4], v[:, h0:h0 + 4]) for h0 in range(0, H, 4)], 1)
        else:
            o = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(D), -1) @ v
        return linear(o.transpose(1, 2).reshape(B, L, C), sd, prefix + ".to_out.0")

    def forward(self, hidden_states: Tensor, encoder_hidden_states: Tensor, timestep: Tensor, image_rotary_emb=None,
                on_hidden=None) -> Tensor:
        """on_hidden(i, x): called with the joint hidden state after block i (full-depth parity tables)."""
        sd, C, p = self.sd, self.C, self.p
        dd = dict(device=self.device, dtype=self.dtype)
        hidden_states, encoder_hidden_states = hidden_states.to(**dd), encoder_hidden_states.to(**dd)
        if image_rotary_emb is not None:
            image_rotary_emb = tuple(t.to(**dd) for t in image_rotary_emb)
        B, Fr, cin, Hh, Ww = hidden_states.shape
        emb = linear(F.silu(linear(timestep_embedding(timestep.float(), C).to(**dd), sd, "time_embedding.linear_1")), sd,
                     "time_embedding.linear_2")
        txt = linear(encoder_hidden_states, sd, "patch_embed.text_proj")
        Lt = txt.shape[1]
        img = F.conv2d(hidden_states.reshape(-1, cin, Hh, Ww), sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"],
                       stride=p)
        img = img.view(B, Fr, C, -1).transpose(2, 3).flatten(1, 2)
        Lv = img.shape[1]
        if not self.use_rope:  # CogVideoX-2B: learned-free sincos table, zeros on the text slots (:505-517)
            img = img + self.pos3d[None, :Lv]
        x = torch.cat([txt, img], dim=1)
        silu_emb = F.silu(emb)
        for i in range(self.L):
            pre = f"transformer_blocks.{i}"
            sh, sc, g, esh, esc, eg = linear(silu_emb, sd, pre + ".norm1.linear").chunk(6, dim=1)
            n = ln(x, sd, pre + ".norm1.norm", self.eps)
            h = torch.cat([n[:, :Lt] * (1 + esc)[:, None] + esh[:, None], n[:, Lt:] * (1 + sc)[:, None] + sh[:, None]], 1)
            a = self.attn(h, pre + ".attn1", Lt, image_rotary_emb if self.use_rope else None)
            x = torch.cat([x[:, :Lt] + eg[:, None] * a[:, :Lt], x[:, Lt:] + g[:, None] * a[:, Lt:]], 1)
            sh, sc, g, esh, esc, eg = linear(silu_emb, sd, pre + ".norm2.linear").chunk(6, dim=1)
            n = ln(x, sd, pre + ".norm2.norm", self.eps)
            h = torch.cat([n[:, :Lt] * (1 + esc)[:, None] + esh[:, None], n[:, Lt:] * (1 + sc)[:, None] + sh[:, None]], 1)
            f = linear(F.gelu(linear(h, sd, pre + ".ff.net.0.proj"), approximate="tanh"), sd, pre + ".ff.net.2")
            x = torch.cat([x[:, :Lt] + eg[:, None] * f[:, :Lt], x[:, Lt:] + g[:, None] * f[:, Lt:]], 1)
            if on_hidden is not None:
                on_hidden(i, x)
        # norm_final: on the video rows (2B) or on the joint sequence then sliced (5B) — LayerNorm is row-wise: identical
        v = ln(x[:, Lt:], sd, "norm_final", self.eps)
        shift, scale = linear(silu_emb, sd, "norm_out.linear").chunk(2, dim=1)  # chunk_dim=1: (shift, scale)
        v = ln(v, sd, "norm_out.norm", self.eps) * (1 + scale)[:, None] + shift[:, None]
        v = linear(v, sd, "proj_out")
        out = v.reshape(B, Fr, Hh // p, Ww // p, self.co, p, p).permute(0, 1, 4, 2, 5, 3, 6).flatten(5, 6).flatten(3, 4)
        return out

    __call__ = forward


# ------------------------------------------------------------------------------------------------- scheduler / sampling
def ddim_alphas(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, snr_shift_scale=3.0,
                rescale_betas_zero_snr=True) -> Tensor:
    """scheduling_ddim_cogvideox.py:193-213 (scaled_linear) + :87-115."""
    betas = torch.linspace(beta_start**0.5, beta_end**0.5, num_train_timesteps, dtype=torch.float64) ** 2
    ac = torch.cumprod(1.0 - betas, dim=0)
    ac = ac / (snr_shift_scale + (1 - snr_shift_scale) * ac)
    if rescale_betas_zero_snr:
        s = ac.sqrt()
        s0, sT = s[0].clone(), s[-1].clone()
        s = (s - sT) * (s0 / (s0 - sT))
        ac = s**2
    return ac


def ddim_timesteps(num_inference_steps: int, num_train_timesteps: int = 1000, spacing: str = "trailing") -> List[int]:
    if spacing == "trailing":
        ratio = num_train_timesteps / num_inference_steps
        return [int(v) - 1 for v in np.round(np.arange(num_train_timesteps, 0, -ratio)).astype(np.int64)]
    ratio = num_train_timesteps // num_inference_steps
    return [int(v) for v in (np.arange(0, num_inference_steps) * ratio).round()[::-1]]


def ddim_coeffs_v(t: int, num_inference_steps: int, ac: Tensor, num_train_timesteps: int = 1000):
    """prev = c_z * sample + c_v * v  (step(), v_prediction, :358-388)."""
    prev_t = t - num_train_timesteps // num_inference_steps
    a_t = float(ac[t])
    a_prev = float(ac[prev_t]) if prev_t >= 0 else 1.0
    a_coef = math.sqrt((1 - a_prev) / (1 - a_t))
    b = math.sqrt(a_prev) - math.sqrt(a_t) * a_coef
    return a_coef + b * math.sqrt(a_t), -b * math.sqrt(1 - a_t)


def dynamic_cfg(guidance_scale: float, t: int, num_inference_steps: int) -> float:
    """pipeline_cogvideox.py:702-705."""
    return 1 + guidance_scale * ((1 - math.cos(math.pi * ((num_inference_steps - t) / num_inference_steps) ** 5.0)) / 2)


def synth_state_dict(num_layers, num_heads, head_dim=64, text_embed_dim=4096, in_channels=16, out_channels=16,
                     time_embed_dim=512, patch_size=2, seed: int = 777) -> Dict[str, Tensor]:
    """Seeded random weights with the THUDM/CogVideoX-* transformer key names and shapes."""
    g = torch.Generator().manual_seed(seed)
    C = num_heads * head_dim
    sd: Dict[str, Tensor] = {}

    def lin(name, n_out, n_in, scale=None):
        s = min(0.08, 1.0 / math.sqrt(n_in)) if scale is None else scale
        sd[name + ".weight"] = torch.randn(n_out, n_in, generator=g) * s
        sd[name + ".bias"] = torch.randn(n_out, generator=g) * 0.02

    def norm(name, n):
        sd[name + ".weight"] = 1 + 0.1 * torch.randn(n, generator=g)
        sd[name + ".bias"] = 0.05 * torch.randn(n, generator=g)

    sd["patch_embed.proj.weight"] = torch.randn(C, in_channels, patch_size, patch_size, generator=g) * 0.1
    sd["patch_embed.proj.bias"] = torch.randn(C, generator=g) * 0.02
    lin("patch_embed.text_proj", C, text_embed_dim)
    lin("time_embedding.linear_1", time_embed_dim, C)
    lin("time_embedding.linear_2", time_embed_dim, time_embed_dim)
    for i in range(num_layers):
        p = f"transformer_blocks.{i}"
        for n in ("norm1", "norm2"):
            lin(f"{p}.{n}.linear", 6 * C, time_embed_dim, scale=0.02)
            norm(f"{p}.{n}.norm", C)
        norm(p + ".attn1.norm_q", head_dim)
        norm(p + ".attn1.norm_k", head_dim)
        for l in ("to_q", "to_k", "to_v", "to_out.0"):
            lin(f"{p}.attn1.{l}", C, C)
        lin(p + ".ff.net.0.proj", 4 * C, C)
        lin(p + ".ff.net.2", C, 4 * C)
    norm("norm_final", C)
    lin("norm_out.linear", 2 * C, time_embed_dim, scale=0.02)
    norm("norm_out.norm", C)
    lin("proj_out", patch_size * patch_size * out_channels, C)
    return sd
