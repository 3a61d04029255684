"""TEST INFRASTRUCTURE ONLY — CPU restatement (plain PyTorch, any dtype) of the Open-Sora v1.2 VAE decode path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; nothing under videosys_amd/ does.

Restates, function by function, from a flat reference-keyed state dict:
  * VideoAutoencoderPipeline.decode              /root/reference/videosys/models/autoencoders/autoencoder_kl_open_sora.py:672-695
  * VAE_Temporal.decode                          :453-462
  * Decoder.forward (temporal)                   :349-376   (construction order :310-345)
  * ResBlock.forward                             :152-164
  * CausalConv3d.forward                         :120-124   (padding :108-113)
  * VideoAutoencoderKL.decode                    :522-538
  * diffusers==0.30.0 AutoencoderKL.decode / vae.Decoder / UNetMidBlock2D / Attention(AttnProcessor2_0) / UpDecoderBlock2D /
    ResnetBlock2D / Upsample2D — THIRD-PARTY, not vendored in the reference tree and not installable here: restated from the
    published source; **parity unpinned** for these leaves.

Encode side (image / video conditioning, SURVEY.md §8 "callers either side"): VideoAutoencoderPipeline.encode :653-670,
VAE_Temporal.encode :441-451 + Encoder.forward :258-272 (construction :203-256; the strided CausalConv3d :107-118),
DiagonalGaussianDistribution :21-40, VideoAutoencoderKL.encode :503-520; and the THIRD-PARTY diffusers AutoencoderKL.encode /
vae.Encoder / DownEncoderBlock2D / Downsample2D (restated, **parity unpinned**, as above).

Pinning: tests/test_vae_cpu.py checks this file against the reference's own classes (imported through oracle/ref_loader.py
with the diffusers leaves stubbed) when /root/reference is present, and against tests/golden/opensora_vae_small.pt (minted by
oracle/make_golden_vae.py from the reference classes) everywhere.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

SHIFT = (-0.10, 0.34, 0.27, 0.98)   # OpenSoraVAE_V1_2, autoencoder_kl_open_sora.py:756-757
SCALE = (3.85, 2.32, 2.33, 3.06)
SD_SCALE = 0.18215


def causal_conv3d(x, w, b):
    """CausalConv3d: zero pad (kt - 1) frames in front, k // 2 pixels around, then Conv3d (:108-124; stride 1, dilation 1)."""
    kt, kh, kw = w.shape[2:]
    x = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2, kt - 1, 0), mode="constant")
    return F.conv3d(x, w, b)


def _gn(x, sd, k, eps):
    return F.group_norm(x, 32, sd[k + ".weight"], sd[k + ".bias"], eps)


def res_block_3d(sd, k, x):
    """ResBlock.forward :152-164 (GroupNorm default eps 1e-5, SiLU, bias-free convs, conv3 = 1x1x1 when channels change)."""
    h = causal_conv3d(F.silu(_gn(x, sd, k + ".norm1", 1e-5)), sd[k + ".conv1.conv.weight"], None)
    h = causal_conv3d(F.silu(_gn(h, sd, k + ".norm2", 1e-5)), sd[k + ".conv2.conv.weight"], None)
    if (k + ".conv3.conv.weight") in sd:
        x = causal_conv3d(x, sd[k + ".conv3.conv.weight"], None)
    return h + x


def temporal_decode(sd, z, num_frames):
    """VAE_Temporal.decode :453-462 with the VAE_Temporal_SD geometry (:465-478)."""
    t = "temporal_vae."
    tpad = 0 if num_frames % 4 == 0 else 4 - num_frames % 4
    x = causal_conv3d(z, sd[t + "post_quant_conv.conv.weight"], sd[t + "post_quant_conv.conv.bias"])
    d = t + "decoder."
    x = causal_conv3d(x, sd[d + "conv1.conv.weight"], sd[d + "conv1.conv.bias"])
    for i in range(4):
        x = res_block_3d(sd, f"{d}res_blocks.{i}", x)
    for i in (3, 2, 1, 0):
        for j in range(4):
            x = res_block_3d(sd, f"{d}block_res_blocks.{i}.{j}", x)
        if i > 0 and f"{d}conv_blocks.{i - 1}.conv.weight" in sd:
            x = causal_conv3d(x, sd[f"{d}conv_blocks.{i - 1}.conv.weight"], sd[f"{d}conv_blocks.{i - 1}.conv.bias"])
            B, C2, T, H, W = x.shape   # "B (C ts) T H W -> B C (T ts) H W", ts = 2 (:362-368)
            x = x.view(B, C2 // 2, 2, T, H, W).permute(0, 1, 3, 2, 4, 5).reshape(B, C2 // 2, 2 * T, H, W)
    x = causal_conv3d(F.silu(_gn(x, sd, d + "norm1", 1e-5)), sd[d + "conv_out.conv.weight"], sd[d + "conv_out.conv.bias"])
    return x[:, :, tpad:]


def _resnet_2d(sd, k, x):
    h = F.conv2d(F.silu(_gn(x, sd, k + ".norm1", 1e-6)), sd[k + ".conv1.weight"], sd[k + ".conv1.bias"], padding=1)
    h = F.conv2d(F.silu(_gn(h, sd, k + ".norm2", 1e-6)), sd[k + ".conv2.weight"], sd[k + ".conv2.bias"], padding=1)
    if (k + ".conv_shortcut.weight") in sd:
        x = F.conv2d(x, sd[k + ".conv_shortcut.weight"], sd[k + ".conv_shortcut.bias"])
    return x + h


def _attn_2d(sd, k, x):
    b, c, h, w = x.shape
    t = _gn(x.view(b, c, h * w), sd, k + ".group_norm", 1e-6).transpose(1, 2)
    q = F.linear(t, sd[k + ".to_q.weight"], sd[k + ".to_q.bias"])
    kk = F.linear(t, sd[k + ".to_k.weight"], sd[k + ".to_k.bias"])
    v = F.linear(t, sd[k + ".to_v.weight"], sd[k + ".to_v.bias"])
    p = torch.softmax((q.float() @ kk.float().transpose(1, 2)) / (c ** 0.5), dim=-1).to(x.dtype)
    o = F.linear(p @ v, sd[k + ".to_out.0.weight"], sd[k + ".to_out.0.bias"])
    return o.transpose(1, 2).reshape(b, c, h, w) + x


def spatial_decode(sd, z):
    """diffusers AutoencoderKL.decode at the SDXL-VAE config: post_quant_conv, conv_in, mid (resnet, attention, resnet),
    4 up blocks of 3 resnets (512, 512, 256, 128; nearest-2x + conv after the first three), GroupNorm, SiLU, conv_out."""
    s = "spatial_vae.module."
    x = F.conv2d(z, sd[s + "post_quant_conv.weight"], sd[s + "post_quant_conv.bias"])
    d = s + "decoder."
    x = F.conv2d(x, sd[d + "conv_in.weight"], sd[d + "conv_in.bias"], padding=1)
    x = _resnet_2d(sd, d + "mid_block.resnets.0", x)
    x = _attn_2d(sd, d + "mid_block.attentions.0", x)
    x = _resnet_2d(sd, d + "mid_block.resnets.1", x)
    for i in range(4):
        for j in range(3):
            x = _resnet_2d(sd, f"{d}up_blocks.{i}.resnets.{j}", x)
        if f"{d}up_blocks.{i}.upsamplers.0.conv.weight" in sd:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, sd[f"{d}up_blocks.{i}.upsamplers.0.conv.weight"], sd[f"{d}up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    x = F.silu(_gn(x, sd, d + "conv_norm_out", 1e-6))
    return F.conv2d(x, sd[d + "conv_out.weight"], sd[d + "conv_out.bias"], padding=1)


@torch.no_grad()
def decode(sd, z, num_frames, micro_frame_size=17, micro_batch_size=4):
    """VideoAutoencoderPipeline.decode :672-695 (cal_loss False) + VideoAutoencoderKL.decode :522-538."""
    dt = z.dtype
    scale = torch.tensor(SCALE)[None, :, None, None, None].to(dt)
    shift = torch.tensor(SHIFT)[None, :, None, None, None].to(dt)
    z = z * scale + shift
    mz = (micro_frame_size + 3) // 4
    parts = []
    left = num_frames
    for i in range(0, z.shape[2], mz):
        parts.append(temporal_decode(sd, z[:, :, i:i + mz], min(micro_frame_size, left)))
        left -= micro_frame_size
    x_z = torch.cat(parts, dim=2)
    B, _, T, H, W = x_z.shape
    x = x_z.permute(0, 2, 1, 3, 4).reshape(B * T, 4, H, W)
    out = []
    for i in range(0, x.shape[0], micro_batch_size):
        out.append(spatial_decode(sd, x[i:i + micro_batch_size] / SD_SCALE))
    x = torch.cat(out, 0)
    return x.view(B, T, 3, x.shape[-2], x.shape[-1]).permute(0, 2, 1, 3, 4).contiguous()


# ---------------------------------------------------------------------------------------------------- encode
def gaussian_sample(moments, noise):
    """DiagonalGaussianDistribution(moments).sample() with the noise handed in: mean + exp(0.5 clamp(logvar, -30, 20)) * noise
    (autoencoder_kl_open_sora.py:21-40; the diffusers class of the 2-D VAE is the same arithmetic)."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    return mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * noise


def spatial_encode_moments(sd, x):
    """diffusers AutoencoderKL.encode up to the distribution parameters, SDXL-VAE config: conv_in, 4 down blocks of 2 resnets
    (128, 256, 512, 512; pad-right/bottom + 3x3 stride-2 conv after the first three), mid (resnet, attention, resnet),
    GroupNorm, SiLU, conv_out (8 channels), quant_conv.  x [N, 3, H, W] -> [N, 8, H/8, W/8]."""
    s = "spatial_vae.module."
    e = s + "encoder."
    x = F.conv2d(x, sd[e + "conv_in.weight"], sd[e + "conv_in.bias"], padding=1)
    for i in range(4):
        for j in range(2):
            x = _resnet_2d(sd, f"{e}down_blocks.{i}.resnets.{j}", x)
        k = f"{e}down_blocks.{i}.downsamplers.0.conv"
        if (k + ".weight") in sd:
            x = F.conv2d(F.pad(x, (0, 1, 0, 1)), sd[k + ".weight"], sd[k + ".bias"], stride=2)
    x = _resnet_2d(sd, e + "mid_block.resnets.0", x)
    x = _attn_2d(sd, e + "mid_block.attentions.0", x)
    x = _resnet_2d(sd, e + "mid_block.resnets.1", x)
    x = F.conv2d(F.silu(_gn(x, sd, e + "conv_norm_out", 1e-6)), sd[e + "conv_out.weight"], sd[e + "conv_out.bias"], padding=1)
    return F.conv2d(x, sd[s + "quant_conv.weight"], sd[s + "quant_conv.bias"])


def strided_causal_conv3d(x, w, b, t_stride):
    """CausalConv3d with strides (t_stride, 1, 1) (:107-118): (kt - 1) + (1 - t_stride) zero frames in front."""
    kt, kh, kw = w.shape[2:]
    x = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2, (kt - 1) + (1 - t_stride), 0), mode="constant")
    return F.conv3d(x, w, b, stride=(t_stride, 1, 1))


def temporal_encode_moments(sd, x):
    """VAE_Temporal.encode :441-451 up to the distribution parameters (VAE_Temporal_SD geometry :465-478): zero frames in front
    up to a multiple of 4, conv_in, 4 x 4 ResBlocks at 128 / 256 / 256 / 512 channels with a stride-2-in-time conv after the
    second and third group (temporal_downsample (False, True, True): the first slot is an Identity), 4 ResBlocks, GroupNorm,
    SiLU, 1x1x1 conv to 8 channels, quant_conv.  x [B, 4, T, h, w] -> [B, 8, ceil(T/4), h, w]."""
    t = "temporal_vae."
    T = x.shape[2]
    tpad = 0 if T % 4 == 0 else 4 - T % 4
    x = F.pad(x, (0, 0, 0, 0, tpad, 0))
    e = t + "encoder."
    x = causal_conv3d(x, sd[e + "conv_in.conv.weight"], None)
    for i in range(4):
        for j in range(4):
            x = res_block_3d(sd, f"{e}block_res_blocks.{i}.{j}", x)
        k = f"{e}conv_blocks.{i}.conv"
        if i < 3 and (k + ".weight") in sd:
            x = strided_causal_conv3d(x, sd[k + ".weight"], sd[k + ".bias"], 2)
    for i in range(4):
        x = res_block_3d(sd, f"{e}res_blocks.{i}", x)
    x = causal_conv3d(F.silu(_gn(x, sd, e + "norm1", 1e-5)), sd[e + "conv2.conv.weight"], sd[e + "conv2.conv.bias"])
    return causal_conv3d(x, sd[t + "quant_conv.conv.weight"], sd[t + "quant_conv.conv.bias"])


@torch.no_grad()
def encode(sd, x, noise_fn=None, micro_frame_size=17, micro_batch_size=4):
    """VideoAutoencoderPipeline.encode :653-670 (cal_loss False) + VideoAutoencoderKL.encode :503-520: x [B, 3, T, H, W] in
    [-1, 1] -> normalised latents [B, 4, Tz, H/8, W/8].  ``noise_fn(shape)`` is drawn once per 2-D micro batch of frames and
    once per temporal micro batch, in that order — the order of the reference's randn calls (default torch.randn)."""
    noise_fn = noise_fn or (lambda shape: torch.randn(shape))
    dt = x.dtype
    B, _, T, H, W = x.shape
    fr = x.permute(0, 2, 1, 3, 4).reshape(B * T, 3, H, W)
    lat = []
    for i in range(0, fr.shape[0], micro_batch_size):
        m = spatial_encode_moments(sd, fr[i:i + micro_batch_size])
        lat.append(gaussian_sample(m, noise_fn(m[:, :4].shape).to(dt)) * SD_SCALE)
    x_z = torch.cat(lat, 0).view(B, T, 4, H // 8, W // 8).permute(0, 2, 1, 3, 4)
    zs = []
    for i in range(0, T, micro_frame_size):
        m = temporal_encode_moments(sd, x_z[:, :, i:i + micro_frame_size]).to(dt)
        zs.append(gaussian_sample(m, noise_fn(m[:, :4].shape).to(dt)))
    z = torch.cat(zs, dim=2)
    scale = torch.tensor(SCALE)[None, :, None, None, None].to(dt)
    shift = torch.tensor(SHIFT)[None, :, None, None, None].to(dt)
    return (z - shift) / scale
