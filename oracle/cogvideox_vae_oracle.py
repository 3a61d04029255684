"""TEST INFRASTRUCTURE ONLY — CPU restatement (plain PyTorch) of the CogVideoX causal 3-D VAE decode path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; nothing under videosys_amd/ does.

Restates from a flat state dict (keys of the reference module, ``decoder.*``), citing
/root/reference/videosys/models/autoencoders/autoencoder_kl_cogvideox.py:
  * AutoencoderKLCogVideoX.decode / _decode (frame batches of 2 latent frames, the remainder joins the first batch)  :1094-1143
  * AutoencoderKLCogVideoX.tiled_decode, blend_v, blend_h                                                            :1145-1239
  * CogVideoXDecoder3D.forward                                                                                      :835-869
  * CogVideoXMidBlock3D / CogVideoXUpBlock3D.forward                                                                 :468-489, :568-594
  * CogVideoXResnetBlock3D.forward                                                                                   :267-299
  * CogVideoXSpatialNorm3D.forward (first-frame split when the frame count is odd)                                   :165-178
  * CogVideoXCausalConv3d.forward with the per-layer conv_cache ("fake context parallel")                             :112-135
  * CogVideoXUpsample3D.forward (videosys/models/modules/upsampling.py:40-67)
All of it is reference-owned code (nothing third-party on this path), so the oracle is PINNED: tests/test_cogvideox_vae_cpu.py
checks it against the reference class (imported by oracle/ref_loader.build_reference_cogvideox_vae) when /root/reference is
present and against tests/golden/cogvideox_vae_small.pt (minted by oracle/make_golden_cogvideox_vae.py) everywhere.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


class _Cache(dict):
    """conv_cache of every CogVideoXCausalConv3d, keyed by the layer's weight name; cleared between tiles / decodes."""


def causal_conv(sd, k, x, cache):
    """CogVideoXCausalConv3d.forward :112-135: the (kt - 1) frames in front are the cached tail of the previous frame batch, or
    copies of the first frame; spatial zero padding."""
    w, b = sd[k + ".conv.weight"], sd.get(k + ".conv.bias")
    kt, kh, kw = w.shape[2:]
    if kt > 1:
        front = [cache[k]] if k in cache else [x[:, :, :1]] * (kt - 1)
        x = torch.cat(front + [x], dim=2)
        cache[k] = x[:, :, -(kt - 1):].clone()
    x = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2), mode="constant", value=0)
    return F.conv3d(x, w, b)


def spatial_norm(sd, k, f, zq, cache):
    """CogVideoXSpatialNorm3D.forward :165-178."""
    if f.shape[2] > 1 and f.shape[2] % 2 == 1:
        z_first = F.interpolate(zq[:, :, :1], size=(1,) + tuple(f.shape[-2:]))
        z_rest = F.interpolate(zq[:, :, 1:], size=(f.shape[2] - 1,) + tuple(f.shape[-2:]))
        zq = torch.cat([z_first, z_rest], dim=2)
    else:
        zq = F.interpolate(zq, size=f.shape[-3:])
    norm_f = F.group_norm(f, 32, sd[k + ".norm_layer.weight"], sd[k + ".norm_layer.bias"], 1e-6)
    return norm_f * causal_conv(sd, k + ".conv_y", zq, cache) + causal_conv(sd, k + ".conv_b", zq, cache)


def resnet(sd, k, x, zq, cache):
    """CogVideoXResnetBlock3D.forward :267-299 (temb None, dropout 0)."""
    h = causal_conv(sd, k + ".conv1", F.silu(spatial_norm(sd, k + ".norm1", x, zq, cache)), cache)
    h = causal_conv(sd, k + ".conv2", F.silu(spatial_norm(sd, k + ".norm2", h, zq, cache)), cache)
    if (k + ".conv_shortcut.weight") in sd:
        x = F.conv3d(x, sd[k + ".conv_shortcut.weight"], sd[k + ".conv_shortcut.bias"])
    return h + x


def upsample(sd, k, x, compress_time):
    """CogVideoXUpsample3D.forward (modules/upsampling.py:40-67)."""
    if compress_time:
        if x.shape[2] > 1 and x.shape[2] % 2 == 1:
            x_first = F.interpolate(x[:, :, 0], scale_factor=2.0)[:, :, None]
            x = torch.cat([x_first, F.interpolate(x[:, :, 1:], scale_factor=2.0)], dim=2)
        elif x.shape[2] > 1:
            x = F.interpolate(x, scale_factor=2.0)
        else:
            x = F.interpolate(x.squeeze(2), scale_factor=2.0)[:, :, None]
    else:
        b, c, t, h, w = x.shape
        x = F.interpolate(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w), scale_factor=2.0)
        x = x.reshape(b, t, c, *x.shape[2:]).permute(0, 2, 1, 3, 4)
    b, c, t, h, w = x.shape
    y = F.conv2d(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w), sd[k + ".conv.weight"], sd[k + ".conv.bias"], padding=1)
    return y.reshape(b, t, *y.shape[1:]).permute(0, 2, 1, 3, 4)


def decoder(sd, z, cache, temporal_compress_level=2):
    """CogVideoXDecoder3D.forward :835-869 (4 up blocks of layers_per_block + 1 = 4 resnets; the first two also double time)."""
    d = "decoder."
    x = causal_conv(sd, d + "conv_in", z, cache)
    for i in range(2):
        x = resnet(sd, f"{d}mid_block.resnets.{i}", x, z, cache)
    for i in range(4):
        j = 0
        while f"{d}up_blocks.{i}.resnets.{j}.conv1.conv.weight" in sd:
            x = resnet(sd, f"{d}up_blocks.{i}.resnets.{j}", x, z, cache)
            j += 1
        if f"{d}up_blocks.{i}.upsamplers.0.conv.weight" in sd:
            x = upsample(sd, f"{d}up_blocks.{i}.upsamplers.0", x, i < temporal_compress_level)
    x = F.silu(spatial_norm(sd, d + "norm_out", x, z, cache))
    return causal_conv(sd, d + "conv_out", x, cache)


def _frame_batches(num_frames, fb=2):
    """_decode :1100-1105: num_frames // fb batches, the remainder joins the first one."""
    rem = num_frames % fb
    return [(fb * i + (0 if i == 0 else rem), fb * (i + 1) + rem) for i in range(num_frames // fb)]


def decode_plain(sd, z):
    cache = _Cache()
    return torch.cat([decoder(sd, z[:, :, a:b], cache) for a, b in _frame_batches(z.shape[2])], dim=2)


def blend_v(a, b, extent):
    extent = min(a.shape[3], b.shape[3], extent)
    for y in range(extent):
        b[:, :, :, y, :] = a[:, :, :, -extent + y, :] * (1 - y / extent) + b[:, :, :, y, :] * (y / extent)
    return b


def blend_h(a, b, extent):
    extent = min(a.shape[4], b.shape[4], extent)
    for x in range(extent):
        b[:, :, :, :, x] = a[:, :, :, :, -extent + x] * (1 - x / extent) + b[:, :, :, :, x] * (x / extent)
    return b


def tile_geometry(sample_height, sample_width, n_blocks=4, fh=1 / 6, fw=1 / 5):
    """__init__ :983-995 + tiled_decode :1183-1189."""
    ts_h, ts_w = sample_height // 2, sample_width // 2
    tl_h, tl_w = int(ts_h / 2 ** (n_blocks - 1)), int(ts_w / 2 ** (n_blocks - 1))
    return dict(tl_h=tl_h, tl_w=tl_w, ov_h=int(tl_h * (1 - fh)), ov_w=int(tl_w * (1 - fw)), be_h=int(ts_h * fh), be_w=int(ts_w * fw),
                lim_h=ts_h - int(ts_h * fh), lim_w=ts_w - int(ts_w * fw))


@torch.no_grad()
def decode(sd, z, sample_height=480, sample_width=720, tiling=True):
    """AutoencoderKLCogVideoX.decode :1121-1143 (use_slicing False) -> _decode / tiled_decode."""
    g = tile_geometry(sample_height, sample_width)
    H, W = z.shape[-2:]
    if not (tiling and (W > g["tl_w"] or H > g["tl_h"])):
        return decode_plain(sd, z)
    rows = []
    for i in range(0, H, g["ov_h"]):
        row = []
        for j in range(0, W, g["ov_w"]):
            row.append(decode_plain(sd, z[:, :, :, i:i + g["tl_h"], j:j + g["tl_w"]]))
        rows.append(row)
    out_rows = []
    for i, row in enumerate(rows):
        res = []
        for j, tile in enumerate(row):
            if i > 0:
                tile = blend_v(rows[i - 1][j], tile, g["be_h"])
            if j > 0:
                tile = blend_h(row[j - 1], tile, g["be_w"])
            res.append(tile[:, :, :, :g["lim_h"], :g["lim_w"]])
        out_rows.append(torch.cat(res, dim=4))
    return torch.cat(out_rows, dim=3)
