"""Latte's default VAE on MI355X: the decode side of ``AutoencoderKLTemporalDecoder`` (Stable Video Diffusion's VAE; diffusers,
third-party) as the reference's Latte pipeline uses it (/root/reference/videosys/pipelines/latte/pipeline_latte.py:24, 211-217:
``enable_vae_temporal_decoder=True`` loads ``<model_path>/vae_temporal_decoder``; :929-948 ``decode_latents_with_temporal_decoder``:
latents / scaling_factor, chunks of 14 frames through ``vae.decode(x, num_frames=n)``, ``(x / 2 + 0.5).clamp(0, 1) * 255`` as uint8
``b f h w c``).  SURVEY.md §8a row a15; state-dict keys are the checkpoint's (``decoder.mid_block.resnets.N.spatial_res_block.*``,
``...temporal_res_block.*``, ``...time_mixer.mix_factor``, ``decoder.time_conv_out.*``), so the real weights drop in.

Everything runs on the kernels of the Open-Sora VAE (csrc/conv_bf16.hip, csrc/vae_ops.hip); what this decoder adds is TIME:
  * ``TemporalResnetBlock``: GroupNorm over (channels of a group, all frames of the chunk, H, W), SiLU, Conv3d (3, 1, 1) with
    SYMMETRIC zero padding, twice, plus the input.  Frames are consecutive planes of the channels-last row matrix, so a frame
    shift is a constant row shift: the (3, 1, 1) convolution is the tap-shifted implicit GEMM with three taps one plane apart,
    reading a buffer that holds one zero plane in front of and one behind the chunk.
  * ``AlphaBlender`` (learned, switch_spatial_to_temporal_mix): out = (1 - s) x_spatial + s x_temporal with s = sigmoid(mix) and
    x_temporal = x_spatial + h  ==>  out = x_spatial + s h: the blend is folded into the second temporal convolution (weights and
    bias scaled by s once at load, residual = x_spatial in the conv epilogue) — no blend pass, no extra rounding.
  * ``time_conv_out``: Conv3d(3, 3, (3, 1, 1)) on the RGB frames, run on the 128-column padded output of ``conv_out``.

No CPU path: without the HIP library / a GPU every call raises.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

from . import ops
from .ops import VaeGrid
from .vae_open_sora import OpenSoraVAE, _Conv, _Norm, _Res, _conv_w, _vec

BLOCK_OUT = (128, 256, 512, 512)


class _TemporalRes:
    """TemporalResnetBlock + the AlphaBlender factor folded into conv2."""

    def __init__(self, sd, prefix, mix_key, dev):
        t = prefix
        self.n1 = _Norm(sd, t + ".norm1", dev, 1e-5)
        self.c1 = _Conv(sd, t + ".conv1", dev)
        self.n2 = _Norm(sd, t + ".norm2", dev, 1e-5)
        s = float(torch.sigmoid(sd[mix_key].float().reshape(-1)[0]))
        self.sigma = s
        w2 = sd[t + ".conv2.weight"].to(dev).float() * s
        self.c2w = _conv_w(w2)
        self.c2b = _vec((sd[t + ".conv2.bias"].to(dev).float() * s))
        self.C = self.c1.cin
        assert self.c1.kt == 3 and self.c1.ks == 1 and self.c1.cout == self.C


class AutoencoderKLTemporalDecoder(OpenSoraVAE):
    """``decode(latents [B, 4, F, H, W]) -> [B, 3, F, 8H, 8W]`` bf16 and ``decode_latents`` -> uint8 [b, f, h, w, c] on the CPU."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda", scaling_factor: float = 0.18215, decode_chunk_size: int = 14):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("videosys_amd.AutoencoderKLTemporalDecoder needs a HIP device (no CPU path)")
        self.device = dev
        self.scaling_factor = scaling_factor
        self.decode_chunk_size = decode_chunk_size
        self._padded = {}
        sd, d = state_dict, "decoder."
        self.s_conv_in_w = _conv_w(sd[d + "conv_in.weight"].to(dev), None, 64)   # K = 9 * 4 = 36 -> 64
        self.s_conv_in_b = _vec(sd[d + "conv_in.bias"].to(dev))

        def st(prefix):
            return (_Res(sd, prefix + ".spatial_res_block", dev, False),
                    _TemporalRes(sd, prefix + ".temporal_res_block", prefix + ".time_mixer.mix_factor", dev))

        self.mid = [st(f"{d}mid_block.resnets.{i}") for i in range(2)]
        a = d + "mid_block.attentions.0."
        self.a_norm = _Norm(sd, a + "group_norm", dev, 1e-6)
        bf = lambda k: sd[k].to(dev).to(torch.bfloat16).contiguous()
        self.a_wq, self.a_bq = bf(a + "to_q.weight"), bf(a + "to_q.bias")
        self.a_wk, self.a_bk = bf(a + "to_k.weight"), bf(a + "to_k.bias")
        self.a_wv = bf(a + "to_v.weight")
        self.a_wo = bf(a + "to_out.0.weight")
        bo = sd[a + "to_out.0.bias"].to(dev).float() + self.a_wo.float() @ bf(a + "to_v.bias").float()   # P (V + 1 b^T) = P V + b
        self.a_bo = bo.to(torch.bfloat16)
        self.up = []
        for i in range(4):
            res = [st(f"{d}up_blocks.{i}.resnets.{j}") for j in range(3)]
            upk = f"{d}up_blocks.{i}.upsamplers.0.conv"
            self.up.append((res, _Conv(sd, upk, dev) if (upk + ".weight") in sd else None))
        self.s_norm = _Norm(sd, d + "conv_norm_out", dev, 1e-6)
        self.s_out = _Conv(sd, d + "conv_out", dev, n_pad=128)
        # time_conv_out on the 128-column rows conv_out produces (columns 3.. are zero): W[co, tap * 128 + ci]
        wt = sd[d + "time_conv_out.weight"].to(dev).float()          # [3, 3, 3, 1, 1]
        m = torch.zeros(128, 3 * 128, dtype=torch.bfloat16, device=dev)
        for k in range(3):
            m[:3, k * 128:k * 128 + 3] = wt[:, :, k, 0, 0].to(torch.bfloat16)
        self.t_out_w = m.contiguous()
        self.t_out_b = _vec(sd[d + "time_conv_out.bias"].to(dev), 128)

    # ------------------------------------------------------------------------------------------------ time
    def _time_buf(self, Fr, H, W, C):
        """Zeroed [(Fr + 2) planes, C] buffer: plane 0 and plane Fr + 1 stay zero (the symmetric time padding), kernels only write
        planes 1..Fr.  Returned with its two descriptors over the SAME memory: ``gw`` (tf = 1) to write the frames, ``gc``
        (tf = 2, what a 3-tap convolution asks for) under which output frame t reads planes t, t+1, t+2 = frames t-1, t, t+1."""
        key = ("time", Fr, H, W, C)
        hit = self._padded.get(key)
        if hit is None:
            rows = (Fr + 2) * H * W
            gw = VaeGrid(1, Fr, H, W, 0, 1, sample_rows=rows)
            gc = VaeGrid(1, Fr, H, W, 0, 2, sample_rows=rows)
            hit = (torch.zeros(rows, C, dtype=torch.bfloat16, device=self.device), gw, gc)
            self._padded[key] = hit
        return hit

    def _temporal_res(self, x, g: VaeGrid, tr: _TemporalRes):
        """x rows over g = (F frames, T = 1) in any spatial padding -> dense rows (F, 1, H, W): x + sigma * TemporalResnetBlock-h."""
        Fr, H, W, C = g.n, g.H, g.W, tr.C
        assert g.T == 1 and g.tf == 0 and g.sample_rows == g.plane, "frames must be consecutive planes"
        dense = VaeGrid(1, Fr, H, W, 0, 0)
        if g.pad:
            xd = torch.empty(dense.rows, C, dtype=torch.bfloat16, device=self.device)
            ops.regrid(x, VaeGrid(1, Fr, H, W, g.pad, 0), xd, dense, C)   # the frames of the chunk ARE the time axis: same memory
        else:
            xd = x
        buf, gw, gc = self._time_buf(Fr, H, W, C)
        ops.group_norm(xd, dense, buf, gw, C, tr.n1.g, tr.n1.b, tr.n1.eps, True)       # statistics over all frames of the chunk
        y = ops.conv(buf, gc, tr.c1.w, tr.c1.b, C, 3, 1)
        ops.group_norm(y, dense, buf, gw, C, tr.n2.g, tr.n2.b, tr.n2.eps, True)
        out = ops.conv(buf, gc, tr.c2w, tr.c2b, C, 3, 1, res=xd)
        return out, VaeGrid(Fr, 1, H, W, 0, 0)

    def _st_res(self, x, g, pair):
        x, g = self._resblock(x, g, pair[0])
        return self._temporal_res(x, g, pair[1])

    # ------------------------------------------------------------------------------------------------ decode
    def _decode_chunk(self, xz: torch.Tensor, out: torch.Tensor, f0: int):
        """xz planar bf16 [4, F, H, W] (one chunk, F <= decode_chunk_size) -> out[3, f0:f0+F, 8H, 8W].  TemporalDecoder.forward."""
        _, Fr, H, W = xz.shape
        inv = 1.0 / self.scaling_factor
        params = [inv] * 4 + [0.0] * 4 + [1.0 if i % 5 == 0 else 0.0 for i in range(16)] + [0.0] * 4   # no post_quant_conv here
        a = ops.vae_first_im2col(xz, 1, 64, params)
        g = VaeGrid(Fr, 1, H, W, 0, 0)
        x = ops.gemm128(a, self.s_conv_in_w, self.s_conv_in_b)
        x, g = self._st_res(x, g, self.mid[0])
        x, g = self._attention(x, g)
        x, g = self._st_res(x, g, self.mid[1])
        for res, up in self.up:
            for pair in res:
                x, g = self._st_res(x, g, pair)
            if up is not None:
                gp = VaeGrid(Fr, 1, 2 * g.H, 2 * g.W, 1, 0)
                xp = self._padded_buf(gp, up.cin)
                ops.regrid(x, g, xp, gp, up.cin, up=1)
                x = ops.conv(xp, gp, up.w, up.b, up.cin, 1, 3)
                g = gp.conv_out()
        h, gh = self._norm_act(x, g, self.s_norm, 128, 0)
        y = ops.conv(h, gh, self.s_out.w, self.s_out.b, 128, 1, 3)        # rows over (F, 1, 8H, 8W, pad 1), 128 columns (3 real)
        Ho, Wo = gh.H, gh.W
        buf, gw, gc = self._time_buf(Fr, Ho, Wo, 128)
        ops.regrid(y, VaeGrid(1, Fr, Ho, Wo, 1, 0), buf, gw, 128)
        y2 = ops.conv(buf, gc, self.t_out_w, self.t_out_b, 128, 3, 1)
        ops.extract_planar(y2, VaeGrid(Fr, 1, Ho, Wo, 0, 0), 3, 0, out, f0)

    @torch.no_grad()
    def decode(self, latents: torch.Tensor) -> torch.Tensor:
        """latents [B, 4, F, H, W] -> sample [B, 3, F, 8H, 8W] bf16 (before the pipeline's / 2 + 0.5).  As in the reference the chunks
        of ``decode_chunk_size`` frames run over the flattened (b f) axis (pipeline_latte.py:932-942)."""
        if not latents.is_cuda:
            raise RuntimeError("AutoencoderKLTemporalDecoder.decode needs a HIP device tensor (no CPU path)")
        B, C, Fr, H, W = latents.shape
        assert C == 4
        flat = latents.to(torch.bfloat16).permute(1, 0, 2, 3, 4).reshape(4, B * Fr, H, W).contiguous()   # planar, frames = (b f)
        vid = torch.empty(3, B * Fr, 8 * H, 8 * W, dtype=torch.bfloat16, device=self.device)
        for f in range(0, B * Fr, self.decode_chunk_size):
            m = min(self.decode_chunk_size, B * Fr - f)
            self._decode_chunk(flat[:, f:f + m].contiguous(), vid, f)
        return vid.view(3, B, Fr, 8 * H, 8 * W).permute(1, 0, 2, 3, 4).contiguous()

    def decode_latents(self, latents: torch.Tensor) -> torch.Tensor:
        """pipeline_latte.py:929-948 -> uint8 [b, f, h, w, c] on the CPU."""
        v = self.decode(latents).float()
        return ((v / 2.0 + 0.5).clamp(0, 1) * 255).permute(0, 2, 3, 4, 1).to(dtype=torch.uint8).cpu().contiguous()

    __call__ = decode_latents


# ---------------------------------------------------------------------------------------------------- synthetic weights
def temporal_decoder_param_shapes() -> Dict[str, tuple]:
    """Names and shapes of the decode-side parameters of the ``vae_temporal_decoder`` checkpoint (block_out_channels
    (128, 256, 512, 512), layers_per_block 2, latent_channels 4)."""
    p: Dict[str, tuple] = {}

    def norm(n, c):
        p[n + ".weight"] = (c,)
        p[n + ".bias"] = (c,)

    def conv2(n, ci, co, k):
        p[n + ".weight"] = (co, ci, k, k)
        p[n + ".bias"] = (co,)

    def st_res(n, ci, co):
        s = n + ".spatial_res_block"
        norm(s + ".norm1", ci)
        conv2(s + ".conv1", ci, co, 3)
        norm(s + ".norm2", co)
        conv2(s + ".conv2", co, co, 3)
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


def synth_state_dict(seed: int = 0) -> Dict[str, torch.Tensor]:
    """Deterministic random decode-side weights (bf16-representable fp32); no checkpoint can be fetched here (no network)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in temporal_decoder_param_shapes().items():
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
