"""TEST INFRASTRUCTURE ONLY — fp32 restatement of the decode side of diffusers' ``AutoencoderKLTemporalDecoder`` (the Stable Video
Diffusion VAE), the DEFAULT VAE of the reference's Latte pipeline (/root/reference/videosys/pipelines/latte/pipeline_latte.py:24,
211-217 ``enable_vae_temporal_decoder=True`` -> ``vae_temporal_decoder`` subfolder; used :888-889, 929-948
``decode_latents_with_temporal_decoder``: latents / scaling_factor, chunks of 14 frames through ``vae.decode(x, num_frames=n)``,
``(x / 2 + 0.5).clamp(0, 1) * 255`` as uint8 ``b f h w c``).

PARITY UNPINNED LEAF: the class lives in diffusers==0.30.0 (pinned in the reference's requirements.txt, absent from
/root/reference and not installable here), and no reference test covers it.  What follows restates the published algorithm of
that release, module by module, with the checkpoint's state-dict keys:
  * ``TemporalDecoder`` (models/autoencoders/autoencoder_kl_temporal_decoder.py): conv_in 4->512, ``MidBlockTemporalDecoder``
    (SpatioTemporalResBlock, then [Attention, SpatioTemporalResBlock]), four ``UpBlockTemporalDecoder`` (3 SpatioTemporalResBlock
    each, nearest-2x + 3x3 conv on the first three), GroupNorm(32, eps 1e-6) + SiLU + conv_out 128->3, then ``time_conv_out``
    = Conv3d(3, 3, (3, 1, 1), padding (1, 0, 0)) over the frames of the chunk.  No post_quant_conv on this class.
  * ``SpatioTemporalResBlock`` (models/resnet.py): ResnetBlock2D (GroupNorm eps 1e-6) per frame, then ``TemporalResnetBlock``
    (GroupNorm(32, eps 1e-5) over (channels-in-group, FRAMES, H, W) -> SiLU -> Conv3d (3, 1, 1) -> GroupNorm -> SiLU -> Conv3d,
    + input), blended by ``AlphaBlender`` (merge_strategy "learned", switch_spatial_to_temporal_mix=True):
    alpha = 1 - sigmoid(mix_factor);  out = alpha * x_spatial + (1 - alpha) * x_temporal.
  * mid-block ``Attention``: one head of dim 512, GroupNorm(32, eps 1e-6) in front, residual connection.
Only tests/ may import this module; the product (videosys_amd/vae_svd_temporal.py) never does.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
BLOCK_OUT = (128, 256, 512, 512)
SCALING_FACTOR = 0.18215


def _gn(x, sd, p, eps):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def resnet2d(x: Tensor, sd, p: str) -> Tensor:
    """diffusers ResnetBlock2D (temb None, eps 1e-6, output_scale_factor 1)."""
    h = F.conv2d(F.silu(_gn(x, sd, p + ".norm1", 1e-6)), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = F.conv2d(F.silu(_gn(h, sd, p + ".norm2", 1e-6)), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if (p + ".conv_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
    return x + h


def temporal_resnet(x5: Tensor, sd, p: str) -> Tensor:
    """diffusers TemporalResnetBlock on [B, C, F, H, W] (eps 1e-5, kernel (3, 1, 1), padding (1, 0, 0))."""
    h = F.conv3d(F.silu(_gn(x5, sd, p + ".norm1", 1e-5)), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=(1, 0, 0))
    h = F.conv3d(F.silu(_gn(h, sd, p + ".norm2", 1e-5)), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=(1, 0, 0))
    return x5 + h


def spatio_temporal_res(x: Tensor, sd, p: str, num_frames: int) -> Tensor:
    """SpatioTemporalResBlock.forward on [(B F), C, H, W]."""
    x = resnet2d(x, sd, p + ".spatial_res_block")
    bf, c, h, w = x.shape
    b = bf // num_frames
    xs = x.reshape(b, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
    xt = temporal_resnet(xs, sd, p + ".temporal_res_block")
    alpha = 1.0 - torch.sigmoid(sd[p + ".time_mixer.mix_factor"].reshape(()))
    out = alpha * xs + (1.0 - alpha) * xt
    return out.permute(0, 2, 1, 3, 4).reshape(bf, c, h, w)


def attention(x: Tensor, sd, p: str) -> Tensor:
    """diffusers Attention(heads 1, dim_head 512, group norm, residual_connection=True) on [(B F), C, H, W]."""
    n, c, h, w = x.shape
    t = _gn(x, sd, p + ".group_norm", 1e-6).reshape(n, c, h * w).transpose(1, 2)
    q = F.linear(t, sd[p + ".to_q.weight"], sd[p + ".to_q.bias"])
    k = F.linear(t, sd[p + ".to_k.weight"], sd[p + ".to_k.bias"])
    v = F.linear(t, sd[p + ".to_v.weight"], sd[p + ".to_v.bias"])
    a = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(c), dim=-1) @ v
    a = F.linear(a, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return x + a.transpose(1, 2).reshape(n, c, h, w)


def decode_chunk(z: Tensor, sd: Dict[str, Tensor], num_frames: int) -> Tensor:
    """AutoencoderKLTemporalDecoder.decode(z [(B F), 4, H, W], num_frames) -> sample [(B F), 3, 8H, 8W]."""
    d = "decoder."
    x = F.conv2d(z, sd[d + "conv_in.weight"], sd[d + "conv_in.bias"], padding=1)
    x = spatio_temporal_res(x, sd, d + "mid_block.resnets.0", num_frames)
    x = attention(x, sd, d + "mid_block.attentions.0")
    x = spatio_temporal_res(x, sd, d + "mid_block.resnets.1", num_frames)
    for i in range(4):
        for j in range(3):
            x = spatio_temporal_res(x, sd, f"{d}up_blocks.{i}.resnets.{j}", num_frames)
        up = f"{d}up_blocks.{i}.upsamplers.0.conv"
        if (up + ".weight") in sd:
            x = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), sd[up + ".weight"], sd[up + ".bias"], padding=1)
    x = F.conv2d(F.silu(_gn(x, sd, d + "conv_norm_out", 1e-6)), sd[d + "conv_out.weight"], sd[d + "conv_out.bias"], padding=1)
    bf, c, h, w = x.shape
    b = bf // num_frames
    x5 = x.reshape(b, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
    x5 = F.conv3d(x5, sd[d + "time_conv_out.weight"], sd[d + "time_conv_out.bias"], padding=(1, 0, 0))
    return x5.permute(0, 2, 1, 3, 4).reshape(bf, c, h, w)


def decode_latents_with_temporal_decoder(latents: Tensor, sd: Dict[str, Tensor], scaling_factor: float = SCALING_FACTOR,
                                         decode_chunk_size: int = 14, as_uint8: bool = True) -> Tensor:
    """pipeline_latte.py:929-948 on latents [b, 4, f, h, w] -> uint8 [b, f, 8h, 8w, 3] (or the fp32 sample [b, f, 3, 8h, 8w])."""
    b, c, f, h, w = latents.shape
    x = (latents.float() / scaling_factor).permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    sd = {k: v.float() for k, v in sd.items()}
    outs = []
    for i in range(0, b * f, decode_chunk_size):
        chunk = x[i:i + decode_chunk_size]
        outs.append(decode_chunk(chunk, sd, chunk.shape[0]))
    v = torch.cat(outs).reshape(b, f, 3, 8 * h, 8 * w)
    if not as_uint8:
        return v
    return ((v / 2.0 + 0.5).clamp(0, 1) * 255).permute(0, 1, 3, 4, 2).to(torch.uint8).contiguous()


def param_shapes() -> Dict[str, tuple]:
    """Decode-side parameters of the ``vae_temporal_decoder`` checkpoint (block_out_channels (128, 256, 512, 512), layers_per_block 2)."""
    p: Dict[str, tuple] = {}

    def norm(n, c):
        p[n + ".weight"] = (c,)
        p[n + ".bias"] = (c,)

    def conv2(n, ci, co, k):
        p[n + ".weight"] = (co, ci, k, k)
        p[n + ".bias"] = (co,)

    def st_res(n, ci, co):
        s = n + ".spatial_res_block"
        norm(s + ".norm1", ci); conv2(s + ".conv1", ci, co, 3); norm(s + ".norm2", co); conv2(s + ".conv2", co, co, 3)
        if ci != co:
            conv2(s + ".conv_shortcut", ci, co, 1)
        t = n + ".temporal_res_block"
        for k in ("1", "2"):
            norm(t + ".norm" + k, co)
            p[t + ".conv" + k + ".weight"] = (co, co, 3, 1, 1)
            p[t + ".conv" + k + ".bias"] = (co,)
        p[n + ".time_mixer.mix_factor"] = (1,)

    d = "decoder."
    conv2(d + "conv_in", 4, 512, 3)
    st_res(d + "mid_block.resnets.0", 512, 512)
    a = d + "mid_block.attentions.0."
    norm(a + "group_norm", 512)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        p[a + n + ".weight"] = (512, 512)
        p[a + n + ".bias"] = (512,)
    st_res(d + "mid_block.resnets.1", 512, 512)
    prev = 512
    for i, co in enumerate(reversed(BLOCK_OUT)):
        for j in range(3):
            st_res(f"{d}up_blocks.{i}.resnets.{j}", prev, co)
            prev = co
        if i < 3:
            conv2(f"{d}up_blocks.{i}.upsamplers.0.conv", co, co, 3)
    norm(d + "conv_norm_out", 128)
    conv2(d + "conv_out", 128, 3, 3)
    p[d + "time_conv_out.weight"] = (3, 3, 3, 1, 1)
    p[d + "time_conv_out.bias"] = (3,)
    return p


def synth_state_dict(seed: int = 0) -> Dict[str, Tensor]:
    """Seeded random weights with the checkpoint's names and shapes (bf16-representable fp32)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in param_shapes().items():
        is_norm = ".norm" in k or "group_norm" in k or "conv_norm_out" in k
        if k.endswith("mix_factor"):
            v = torch.randn(shp, generator=g) * 0.5
        elif k.endswith(".weight") and len(shp) >= 2:
            fan_in = 1
            for s in shp[1:]:
                fan_in *= s
            v = torch.randn(shp, generator=g) / math.sqrt(fan_in)
        elif k.endswith(".weight") and is_norm:
            v = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif is_norm:
            v = 0.1 * torch.randn(shp, generator=g)
        else:
            v = 0.02 * torch.randn(shp, generator=g)
        sd[k] = v.to(torch.bfloat16).float()
    return sd
