"""CogVideoX causal 3-D VAE decode on MI355X (``AutoencoderKLCogVideoX.tiled_decode``).

Mirrors the decode side of the reference ``AutoencoderKLCogVideoX`` (videosys/models/autoencoders/autoencoder_kl_cogvideox.py):
``decode`` (:1121-1143) -> ``_decode`` (:1094-1119, frame batches of 2 latent frames whose causal convolutions hand their last
two input frames to the next batch, ``conv_cache`` :112-135) or ``tiled_decode`` (:1161-1239, overlapping latent tiles decoded
separately and cross-faded by ``blend_v`` / ``blend_h``), ``CogVideoXDecoder3D`` (:835-869), ``CogVideoXResnetBlock3D`` (:267-299),
``CogVideoXSpatialNorm3D`` (:165-178) and ``CogVideoXUpsample3D`` (modules/upsampling.py:40-67).  State-dict keys are the
reference's (``decoder.*``), so the published ``vae/diffusion_pytorch_model.safetensors`` drops in.

Every convolution is the tap-shifted implicit-GEMM kernel of csrc/conv_bf16.hip (causal padding = two extra frames in front of
the conv-input buffer, filled from the layer's cache or with copies of the first frame); the spatially conditioned norm computes
conv_y(zq) | conv_b(zq) ONCE per norm at latent resolution (a 32-wide GEMM) and applies them inside the GroupNorm/SiLU kernel with
the nearest-neighbour index map of F.interpolate.  No CPU path.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops
from .ops import VaeGrid
from .vae_open_sora import _conv_w, _vec


class _SNorm:
    def __init__(self, sd, prefix, dev):
        self.g = sd[prefix + ".norm_layer.weight"].to(dev).to(torch.bfloat16).contiguous()
        self.b = sd[prefix + ".norm_layer.bias"].to(dev).to(torch.bfloat16).contiguous()
        wy, wb = sd[prefix + ".conv_y.conv.weight"], sd[prefix + ".conv_b.conv.weight"]
        C, zc = wy.shape[0], wy.shape[1]
        w = torch.zeros(2 * C, 32, dtype=torch.bfloat16, device=dev)          # K padded 16 -> 32 (one k-tile)
        w[:C, :zc] = wy.reshape(C, zc).to(dev).to(torch.bfloat16)
        w[C:, :zc] = wb.reshape(C, zc).to(dev).to(torch.bfloat16)
        self.w_yb = w.contiguous()
        self.b_yb = torch.cat([sd[prefix + ".conv_y.conv.bias"], sd[prefix + ".conv_b.conv.bias"]]).to(dev).to(torch.bfloat16).contiguous()
        self.C = C


class _CConv:
    """CogVideoXCausalConv3d 3x3x3 (or the 1x1x1 shortcut / the per-frame 3x3 upsampler conv)."""

    def __init__(self, sd, key, dev, cin_pad=None, n_pad=None):
        w = sd[key + ".weight"].to(dev)
        if cin_pad is not None and w.shape[1] < cin_pad:
            w = torch.cat([w, torch.zeros(w.shape[0], cin_pad - w.shape[1], *w.shape[2:], dtype=w.dtype, device=dev)], 1)
        self.key = key
        self.kt = w.shape[2] if w.dim() == 5 else 1
        self.ks = w.shape[-1]
        self.cin, self.cout = w.shape[1], w.shape[0]
        self.w = _conv_w(w, n_pad)
        b = sd.get(key + ".bias")
        self.b = _vec(b.to(dev), n_pad) if b is not None else None


class _CRes:
    def __init__(self, sd, p, dev):
        self.n1, self.n2 = _SNorm(sd, p + ".norm1", dev), _SNorm(sd, p + ".norm2", dev)
        self.c1, self.c2 = _CConv(sd, p + ".conv1.conv", dev), _CConv(sd, p + ".conv2.conv", dev)
        self.sc = _CConv(sd, p + ".conv_shortcut", dev) if (p + ".conv_shortcut.weight") in sd else None


class CogVideoXVAE:
    """Decode side of AutoencoderKLCogVideoX.  ``decode(z)``: z [B, 16, T, H, W] -> [B, 3, T_out, 8H, 8W] bf16."""

    num_latent_frames_batch_size = 2
    tile_overlap_factor_height = 1 / 6
    tile_overlap_factor_width = 1 / 5

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda", sample_height: int = 480, sample_width: int = 720,
                 scaling_factor: float = 1.15258426, temporal_compression_ratio: int = 4, use_tiling: bool = True):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("videosys_amd.CogVideoXVAE needs a HIP device (no CPU path)")
        self.device = dev
        self.config = type("Cfg", (), dict(scaling_factor=scaling_factor, temporal_compression_ratio=temporal_compression_ratio,
                                           block_out_channels=(128, 256, 256, 512), sample_height=sample_height,
                                           sample_width=sample_width, latent_channels=16))()
        self.use_tiling = use_tiling
        self.tile_sample_min_height = sample_height // 2          # __init__ :983-995
        self.tile_sample_min_width = sample_width // 2
        self.tile_latent_min_height = int(self.tile_sample_min_height / 8)
        self.tile_latent_min_width = int(self.tile_sample_min_width / 8)
        sd, d = state_dict, "decoder."
        self.conv_in = _CConv(sd, d + "conv_in.conv", dev, cin_pad=32)
        self.mid = [_CRes(sd, f"{d}mid_block.resnets.{i}", dev) for i in range(2)]
        self.ups = []
        level = 0
        t = temporal_compression_ratio
        while t > 1:
            level, t = level + 1, t // 2
        for i in range(4):
            res = []
            while f"{d}up_blocks.{i}.resnets.{len(res)}.conv1.conv.weight" in sd:
                res.append(_CRes(sd, f"{d}up_blocks.{i}.resnets.{len(res)}", dev))
            uk = f"{d}up_blocks.{i}.upsamplers.0.conv"
            self.ups.append((res, _CConv(sd, uk, dev) if (uk + ".weight") in sd else None, i < level))
        self.norm_out = _SNorm(sd, d + "norm_out", dev)
        self.conv_out = _CConv(sd, d + "conv_out.conv", dev, n_pad=128)
        self._padded: Dict[tuple, tuple] = {}

    def enable_tiling(self):
        self.use_tiling = True

    def disable_tiling(self):
        self.use_tiling = False

    # ------------------------------------------------------------------------------------------------ building blocks
    def clear_cache(self):
        """Drop the zero-bordered staging buffers (they are kept per geometry; call after changing resolution to free HBM)."""
        self._padded.clear()

    def _padded_buf(self, g: VaeGrid, C: int):
        key = (g.n, g.T, g.H, g.W, g.tf, C)
        hit = self._padded.get(key)
        if hit is None:
            hit = g.alloc(C, self.device, zero=True)
            self._padded[key] = hit
        return hit[1]

    def _front(self, rows, g: VaeGrid, key, cache):
        """The two frames in front of a causal conv's input (CogVideoXCausalConv3d.fake_context_parallel_forward :112-117): the
        cached tail of the previous frame batch, or copies of this batch's first frame; then remember this batch's tail."""
        v = rows.view(g.T + g.tf, g.Hp, g.Wp, rows.shape[1])
        if key in cache:
            v[0:2].copy_(cache[key])
        else:
            v[0].copy_(v[2])
            v[1].copy_(v[2])
        cache[key] = v[g.T:g.T + 2].clone()

    def _snorm_silu(self, x, gx: VaeGrid, norm: _SNorm, zrows, zdims, tf):
        yb = ops.gemm128(zrows, norm.w_yb, norm.b_yb)
        gd = VaeGrid(1, gx.T, gx.H, gx.W, 1, tf)
        y = self._padded_buf(gd, norm.C)
        ops.spatial_norm_silu(x, gx, y, gd, norm.C, norm.g, norm.b, yb, zdims)
        return y, gd

    def _cconv(self, h, gh: VaeGrid, cv: _CConv, cache, res=None):
        self._front(h, gh, cv.key, cache)
        return ops.conv(h, gh, cv.w, cv.b, cv.cin, 3, 3, res=res)

    def _resnet(self, x, gx: VaeGrid, r: _CRes, zrows, zdims, cache):
        h, gh = self._snorm_silu(x, gx, r.n1, zrows, zdims, 2)
        y = self._cconv(h, gh, r.c1, cache)
        h2, gh2 = self._snorm_silu(y, gx, r.n2, zrows, zdims, 2)
        res = x if r.sc is None else ops.gemm128(x, r.sc.w, r.sc.b)
        return self._cconv(h2, gh2, r.c2, cache, res=res)

    def _decode_batch(self, zc: torch.Tensor, cache, out: torch.Tensor, f0: int) -> int:
        """One frame batch of one tile: zc planar bf16 [16, Tc, h, w] -> out[3, f0:f0+T_out, 8h, 8w]; returns T_out.
        CogVideoXDecoder3D.forward :835-869."""
        _, Tc, h, w = zc.shape
        zdims = (Tc, h, w)
        zrows = torch.zeros(Tc * h * w, 32, dtype=torch.bfloat16, device=self.device)
        zrows[:, :16] = zc.permute(1, 2, 3, 0).reshape(-1, 16)
        gz = VaeGrid(1, Tc, h, w, 0, 0)
        gp = VaeGrid(1, Tc, h, w, 1, 2)
        zin = self._padded_buf(gp, 32)
        ops.regrid(zrows, gz, zin, gp, 32)
        x = self._cconv(zin, gp, self.conv_in, cache)
        g = gp.conv_out()
        for r in self.mid:
            x = self._resnet(x, g, r, zrows, zdims, cache)
        for res, up, compress_time in self.ups:
            for r in res:
                x = self._resnet(x, g, r, zrows, zdims, cache)
            if up is not None:
                tmode = 0
                if compress_time and g.T > 1:
                    tmode = 2 if g.T % 2 == 1 else 1
                T2 = g.T if tmode == 0 else (2 * g.T if tmode == 1 else 2 * g.T - 1)
                gu = VaeGrid(1, T2, 2 * g.H, 2 * g.W, 1, 0)
                xu = self._padded_buf(gu, up.cin)
                ops.regrid(x, g, xu, gu, up.cin, up=1, tmode=tmode)
                x = ops.conv(xu, gu, up.w, up.b, up.cin, 1, 3)
                g = gu.conv_out()
        hN, gN = self._snorm_silu(x, g, self.norm_out, zrows, zdims, 2)
        y = self._cconv(hN, gN, self.conv_out, cache)
        ops.extract_planar(y, gN.conv_out(), 3, 0, out, f0)
        return g.T

    def _out_frames(self, T: int) -> int:
        fb = self.num_latent_frames_batch_size
        total = 0
        for a, b in self._batches(T):
            t = b - a
            for _ in range(2):  # two time-doubling up blocks (temporal_compression_ratio 4)
                t = 2 * t - 1 if (t > 1 and t % 2 == 1) else (2 * t if t > 1 else t)
            total += t
        return total

    def _batches(self, T: int):
        fb = self.num_latent_frames_batch_size
        rem = T % fb
        return [(fb * i + (0 if i == 0 else rem), fb * (i + 1) + rem) for i in range(T // fb)]   # _decode :1100-1105

    def _decode_tile(self, zt: torch.Tensor) -> torch.Tensor:
        """zt planar bf16 [16, T, h, w] -> planar [3, T_out, 8h, 8w]; a fresh conv cache per tile (:1209 / :1113)."""
        _, T, h, w = zt.shape
        out = torch.empty(3, self._out_frames(T), 8 * h, 8 * w, dtype=torch.bfloat16, device=self.device)
        cache, f0 = {}, 0
        for a, b in self._batches(T):
            f0 += self._decode_batch(zt[:, a:b].contiguous(), cache, out, f0)
        return out

    # ------------------------------------------------------------------------------------------------ public
    @torch.no_grad()
    def decode(self, z: torch.Tensor, group=None) -> torch.Tensor:
        """``group`` (dsp.py group protocol; every rank holds the same latents): the TILES of a tiled decode are decoded by the ranks
        in turn and gathered once — the reference decodes all of them on every rank (pipeline_cogvideox.py:359-364 under
        engine.py:85-95); an untiled decode is not sharded."""
        if not z.is_cuda:
            raise RuntimeError("CogVideoXVAE.decode needs a HIP device tensor (no CPU path)")
        B, C, T, H, W = z.shape
        assert C == 16
        outs = []
        for b in range(B):
            zb = z[b].to(torch.bfloat16).contiguous()
            if self.use_tiling and (W > self.tile_latent_min_width or H > self.tile_latent_min_height):
                outs.append(self._tiled(zb, group))
            else:
                outs.append(self._decode_tile(zb))
        return torch.stack(outs, 0)

    @staticmethod
    def _deal_tiles(areas, P):
        """Which rank decodes which tile: largest tile first, each to the rank with the least latent area so far (the ragged last row
        and column of tiles are a fraction of a full tile: dealt r, r + P, ... the 3 x 3 tiles of config 5 give one of four ranks 2 880
        of 7 560 latent pixels; this way the busiest has 1 980).  Deterministic: every rank computes the same table.
        Returns (owner[t], slot[t], tiles per rank)."""
        load, count = [0] * P, [0] * P
        owner, slot = [0] * len(areas), [0] * len(areas)
        for t in sorted(range(len(areas)), key=lambda k: (-areas[k], k)):
            r = min(range(P), key=lambda q: (load[q], q))
            owner[t], slot[t] = r, count[r]
            load[r] += areas[t]
            count[r] += 1
        return owner, slot, max(count) if count else 0

    def _tiles_over_ranks(self, zb, coords, tl_h, tl_w, group):
        """The decoded tiles of ``coords`` dealt over the ranks of the group (_deal_tiles): one all-gather of the ranks' tiles (padded
        to the full tile's pixel shape); every tile is independent (fresh caches), so the bits are the unsharded decode's."""
        from . import dsp

        P, r = dsp.group_size(group), dsp.group_rank(group)
        n = len(coords)
        Hh, Ww = zb.shape[-2:]
        owner, slot, per = self._deal_tiles([min(tl_h, Hh - i) * min(tl_w, Ww - j) for i, j in coords], P)
        mine = sorted((t for t in range(n) if owner[t] == r), key=lambda t: slot[t])
        dec = {t: self._decode_tile(zb[:, :, coords[t][0]:coords[t][0] + tl_h, coords[t][1]:coords[t][1] + tl_w].contiguous()) for t in mine}
        # (the frame count of a decoded tile depends on the latent frame count alone: every rank can derive the padded shape only
        #  once somebody has decoded — rank 0 always owns tile 0, and the shape travels with the data: a fixed-size header gather)
        shp = torch.zeros(4, dtype=torch.int64, device=zb.device)
        if dec:
            any_tile = next(iter(dec.values()))
            shp[:] = torch.tensor([any_tile.shape[0], any_tile.shape[1], 8 * tl_h, 8 * tl_w])
        allshp = torch.empty(P * 4, dtype=torch.int64, device=zb.device)
        dsp.all_gather_into_tensor(allshp, shp, group)
        Cc, F, TH, TW = (int(v) for v in allshp.view(P, 4)[0].tolist())
        buf = torch.zeros(per, Cc, F, TH, TW, dtype=torch.bfloat16, device=zb.device)
        for t in mine:
            tile = dec[t]
            buf[slot[t], :, :, :tile.shape[-2], :tile.shape[-1]] = tile
        allb = torch.empty(P * per, Cc, F, TH, TW, dtype=torch.bfloat16, device=zb.device)
        dsp.all_gather_into_tensor(allb, buf, group)
        H, W = zb.shape[-2:]
        out = []
        for t, (i, j) in enumerate(coords):
            h, w = 8 * min(tl_h, H - i), 8 * min(tl_w, W - j)
            out.append(allb[owner[t] * per + slot[t]][:, :, :h, :w].contiguous())
        return out

    def _tiled(self, zb: torch.Tensor, group=None) -> torch.Tensor:
        """tiled_decode :1161-1239."""
        _, T, H, W = zb.shape
        tl_h, tl_w = self.tile_latent_min_height, self.tile_latent_min_width
        ov_h, ov_w = int(tl_h * (1 - self.tile_overlap_factor_height)), int(tl_w * (1 - self.tile_overlap_factor_width))
        be_h, be_w = int(self.tile_sample_min_height * self.tile_overlap_factor_height), int(self.tile_sample_min_width * self.tile_overlap_factor_width)
        lim_h, lim_w = self.tile_sample_min_height - be_h, self.tile_sample_min_width - be_w
        if group is not None:
            coords = [(i, j) for i in range(0, H, ov_h) for j in range(0, W, ov_w)]
            flat = self._tiles_over_ranks(zb, coords, tl_h, tl_w, group)
            ncol = len(range(0, W, ov_w))
            rows = [flat[k * ncol:(k + 1) * ncol] for k in range(len(flat) // ncol)]
        else:
            rows = [[self._decode_tile(zb[:, :, i:i + tl_h, j:j + tl_w].contiguous()) for j in range(0, W, ov_w)] for i in range(0, H, ov_h)]
        out_rows = []
        for i, row in enumerate(rows):
            res = []
            for j, tile in enumerate(row):
                if i > 0:   # blend_v / blend_h modify the tile in place, so later tiles see the blended neighbours (:1227-1231)
                    a = rows[i - 1][j]
                    ops.blend_edge(a, tile, min(a.shape[-2], tile.shape[-2], be_h), 0)
                if j > 0:
                    a = row[j - 1]
                    ops.blend_edge(a, tile, min(a.shape[-1], tile.shape[-1], be_w), 1)
                res.append(tile[:, :, :lim_h, :lim_w])
            out_rows.append(torch.cat(res, dim=3))
        return torch.cat(out_rows, dim=2).contiguous()

    def decode_latents(self, latents: torch.Tensor, group=None) -> torch.Tensor:
        """pipeline_cogvideox.py:359-364: latents [B, T, 16, H, W] -> frames [B, 3, T_out, 8H, 8W] (``group``: see decode)."""
        z = latents.permute(0, 2, 1, 3, 4)
        z = 1 / self.config.scaling_factor * z
        return self.decode(z, group)

    __call__ = decode_latents


# ---------------------------------------------------------------------------------------------------- synthetic weights
AutoencoderKLCogVideoX = CogVideoXVAE   # the reference's class name (autoencoder_kl_cogvideox.py); decode side only


def decoder_param_shapes() -> Dict[str, tuple]:
    """Decode-side parameters of the reference AutoencoderKLCogVideoX (checked against its state_dict in the CPU tests)."""
    p: Dict[str, tuple] = {}

    def cconv(name, ci, co, k):
        p[name + ".conv.weight"] = (co, ci, k, k, k)
        p[name + ".conv.bias"] = (co,)

    def snorm(name, c):
        p[name + ".norm_layer.weight"] = (c,)
        p[name + ".norm_layer.bias"] = (c,)
        cconv(name + ".conv_y", 16, c, 1)
        cconv(name + ".conv_b", 16, c, 1)

    def res(name, ci, co):
        snorm(name + ".norm1", ci)
        snorm(name + ".norm2", co)
        cconv(name + ".conv1", ci, co, 3)
        cconv(name + ".conv2", co, co, 3)
        if ci != co:
            p[name + ".conv_shortcut.weight"] = (co, ci, 1, 1, 1)
            p[name + ".conv_shortcut.bias"] = (co,)

    d = "decoder."
    cconv(d + "conv_in", 16, 512, 3)
    for i in range(2):
        res(f"{d}mid_block.resnets.{i}", 512, 512)
    prev = 512
    for i, co in enumerate((512, 256, 256, 128)):
        for j in range(4):
            res(f"{d}up_blocks.{i}.resnets.{j}", prev, co)
            prev = co
        if i < 3:
            p[f"{d}up_blocks.{i}.upsamplers.0.conv.weight"] = (co, co, 3, 3)
            p[f"{d}up_blocks.{i}.upsamplers.0.conv.bias"] = (co,)
    snorm(d + "norm_out", 128)
    cconv(d + "conv_out", 128, 3, 3)
    return p


def synth_state_dict(seed: int = 0) -> Dict[str, torch.Tensor]:
    """Deterministic random decode-side weights (bf16-representable fp32).  conv_y starts around 1 and conv_b around 0 so the
    spatially conditioned norm behaves like a norm at initialisation."""
    import math

    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in decoder_param_shapes().items():
        if k.endswith(".weight") and len(shp) >= 2:
            fan_in = 1
            for s in shp[1:]:
                fan_in *= s
            v = torch.randn(shp, generator=g) / math.sqrt(fan_in)
            if ".conv_y." in k or ".conv_b." in k:
                v = v * 0.3
        elif k.endswith("norm_layer.weight"):
            v = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("norm_layer.bias"):
            v = 0.1 * torch.randn(shp, generator=g)
        elif ".conv_y.conv.bias" in k:
            v = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:
            v = 0.02 * torch.randn(shp, generator=g)
        sd[k] = v.to(torch.bfloat16).float()
    return sd
