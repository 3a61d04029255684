"""Open-Sora v1.2 VAE on MI355X: decode and, for image / video conditioning, encode.

Mirrors the decode side of the reference ``VideoAutoencoderPipeline`` (videosys/models/autoencoders/
autoencoder_kl_open_sora.py:620-735, factory ``OpenSoraVAE_V1_2`` :738-761):

    z * scale + shift                                     (:676-677)
    -> per micro-batch of 17 frames (5 latent frames): VAE_Temporal.decode (:453-462: post_quant_conv, Decoder :275-376,
       drop the leading time padding)                     (:683-689)
    -> VideoAutoencoderKL.decode (:522-538): every frame through the 2-D SDXL decoder (diffusers 0.30.0 AutoencoderKL,
       third-party: post_quant_conv + vae.Decoder), x / 0.18215 first.

Every convolution runs on the tap-shifted implicit-GEMM MFMA kernel (csrc/conv_bf16.hip) over channels-last activations;
GroupNorm/SiLU, upsampling, depth-to-space and the first/last small-channel layers are the kernels of csrc/vae_ops.hip; the
single-head d = 512 mid-block attention is three batched GEMMs + a row softmax.  State-dict keys are the reference's
(``temporal_vae.decoder.*``, ``spatial_vae.module.decoder.*``) so the real checkpoints drop in.

``encode`` (reference :653-670, VAE_Temporal.encode :441-451, VideoAutoencoderKL.encode :503-520) is what the pipeline's
conditioning needs (reference frames, ``loop`` > 1): the 2-D SDXL encoder per frame, then the temporal encoder per 17-frame
micro batch, each followed by a draw from its diagonal Gaussian.  It runs on the same kernels; the three stride-2 convolutions of
the 2-D encoder and the two stride-2-in-time causal convolutions of the temporal encoder are computed as the stride-1
convolution sampled at the odd positions (vsys_subsample) — 4x / 2x the arithmetic of a strided kernel on those five layers of
a path that runs once per conditioning clip, in exchange for no new convolution kernel.  Encoder weights are optional: a
checkpoint without ``*.encoder.*`` keys gives a decode-only object.

There is no CPU path: without the HIP library / a GPU every call raises.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Callable, Dict, List, Optional

import torch

from . import ops
from .ops import VaeGrid

_SD_SCALE = 0.18215  # autoencoder_kl_open_sora.py:527,533


def _conv_w(w: torch.Tensor, n_pad: Optional[int] = None, k_pad: Optional[int] = None) -> torch.Tensor:
    """[Cout, Cin, (kt,) kh, kw] -> bf16 [Cout(+pad), taps*Cin(+pad)] with k = tap*Cin + c."""
    co = w.shape[0]
    if w.dim() == 5:
        m = w.permute(0, 2, 3, 4, 1).reshape(co, -1)
    else:
        m = w.permute(0, 2, 3, 1).reshape(co, -1)
    n_pad = n_pad or co
    k_pad = k_pad or m.shape[1]
    out = torch.zeros(n_pad, k_pad, dtype=torch.bfloat16, device=w.device)
    out[:co, :m.shape[1]] = m.to(torch.bfloat16)
    return out.contiguous()


def _vec(v: Optional[torch.Tensor], n_pad: Optional[int] = None) -> Optional[torch.Tensor]:
    if v is None:
        return None
    n_pad = n_pad or v.numel()
    out = torch.zeros(n_pad, dtype=torch.bfloat16, device=v.device)
    out[:v.numel()] = v.to(torch.bfloat16)
    return out


class _Conv:
    """One convolution: weight matrix, bias, geometry (cin, temporal taps, spatial kernel)."""

    def __init__(self, sd, prefix, dev, n_pad=None):
        w = sd[prefix + ".weight"].to(dev)
        self.kt = w.shape[2] if w.dim() == 5 else 1
        self.ks = w.shape[-1]
        self.cin, self.cout = w.shape[1], w.shape[0]
        self.w = _conv_w(w, n_pad)
        b = sd.get(prefix + ".bias")
        self.b = _vec(b.to(dev), n_pad) if b is not None else None


class _Norm:
    def __init__(self, sd, prefix, dev, eps):
        self.g = sd[prefix + ".weight"].to(dev).to(torch.bfloat16).contiguous()
        self.b = sd[prefix + ".bias"].to(dev).to(torch.bfloat16).contiguous()
        self.eps = eps


class _Res:
    """norm1 -> SiLU -> conv1 -> norm2 -> SiLU -> conv2 (+ shortcut): ResBlock (autoencoder_kl_open_sora.py:127-164, GroupNorm
    eps 1e-5, bias-free causal 3x3x3 convs, ``conv3``) and diffusers ResnetBlock2D (eps 1e-6, ``conv_shortcut``)."""

    def __init__(self, sd, prefix, dev, three_d):
        sub = ".conv" if three_d else ""
        eps = 1e-5 if three_d else 1e-6
        self.n1 = _Norm(sd, prefix + ".norm1", dev, eps)
        self.c1 = _Conv(sd, prefix + ".conv1" + sub, dev)
        self.n2 = _Norm(sd, prefix + ".norm2", dev, eps)
        self.c2 = _Conv(sd, prefix + ".conv2" + sub, dev)
        sk = prefix + (".conv3.conv" if three_d else ".conv_shortcut")
        self.sc = _Conv(sd, sk, dev) if (sk + ".weight") in sd else None


class OpenSoraVAE:
    """Decode side of VideoAutoencoderPipeline; ``decode(z, num_frames)`` as autoencoder_kl_open_sora.py:672-695."""

    micro_frame_size = 17
    micro_batch_size = 4
    has_encoder = False
    out_channels = 4
    shift = (-0.10, 0.34, 0.27, 0.98)   # OpenSoraVAE_V1_2, autoencoder_kl_open_sora.py:756-757
    scale = (3.85, 2.32, 2.33, 3.06)
    time_downsample_factor = 4

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda", micro_frame_size: int = 17, micro_batch_size: int = 4,
                 frames_per_launch: int = 16):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("videosys_amd.OpenSoraVAE needs a HIP device (no CPU path)")
        self.device = dev
        self.micro_frame_size = micro_frame_size
        self.micro_batch_size = micro_batch_size   # kept for API parity; frames are independent in the 2-D decoder
        self.frames_per_launch = frames_per_launch
        self.micro_z_frame_size = self.get_temporal_latent_size(micro_frame_size)
        self._padded: Dict[tuple, tuple] = {}
        self._init_temporal(state_dict, dev)
        self._init_spatial(state_dict, dev, "spatial_vae.module.")
        self.has_encoder = "temporal_vae.encoder.conv_in.conv.weight" in state_dict and \
            "spatial_vae.module.encoder.conv_in.weight" in state_dict
        if self.has_encoder:
            self._init_encoders(state_dict, dev)

    def _init_temporal(self, sd, dev):
        # ---- temporal VAE (VAE_Temporal_SD: filters 128, multipliers (1,2,2,4), 4 res blocks, temporal up (F,T,T))
        t = "temporal_vae."
        self.t_pq = (sd[t + "post_quant_conv.conv.weight"].float().reshape(4, 4).cpu(), sd[t + "post_quant_conv.conv.bias"].float().cpu())
        d = t + "decoder."
        w1 = sd[d + "conv1.conv.weight"].to(dev)
        self.t_conv1_w = _conv_w(w1, None, 128)                      # K = 27 taps * 4 channels = 108 -> 128
        self.t_conv1_b = _vec(sd[d + "conv1.conv.bias"].to(dev))
        self.t_res = [_Res(sd, f"{d}res_blocks.{i}", dev, True) for i in range(4)]
        self.t_blocks = [[_Res(sd, f"{d}block_res_blocks.{i}.{j}", dev, True) for j in range(4)] for i in range(4)]
        self.t_up = {i: _Conv(sd, f"{d}conv_blocks.{i}.conv", dev) for i in range(3) if f"{d}conv_blocks.{i}.conv.weight" in sd}
        self.t_norm = _Norm(sd, d + "norm1", dev, 1e-5)
        self.t_out = _Conv(sd, d + "conv_out.conv", dev, n_pad=128)

    def _init_spatial(self, sd, dev, s):
        """2-D SD / SDXL decoder (diffusers AutoencoderKL decode side) under key prefix ``s``."""
        self.s_pq = (sd[s + "post_quant_conv.weight"].float().reshape(4, 4).cpu(), sd[s + "post_quant_conv.bias"].float().cpu())
        d = s + "decoder."
        self.s_conv_in_w = _conv_w(sd[d + "conv_in.weight"].to(dev), None, 64)   # K = 9 * 4 = 36 -> 64
        self.s_conv_in_b = _vec(sd[d + "conv_in.bias"].to(dev))
        self.s_mid = [_Res(sd, f"{d}mid_block.resnets.{i}", dev, False) for i in range(2)]
        a = d + "mid_block.attentions.0."
        self.a_norm = _Norm(sd, a + "group_norm", dev, 1e-6)
        bf = lambda k: sd[k].to(dev).to(torch.bfloat16).contiguous()
        self.a_wq, self.a_bq = bf(a + "to_q.weight"), bf(a + "to_q.bias")
        self.a_wk, self.a_bk = bf(a + "to_k.weight"), bf(a + "to_k.bias")
        self.a_wv = bf(a + "to_v.weight")
        self.a_wo = bf(a + "to_out.0.weight")
        # softmax rows sum to 1, so the value bias passes through the attention: P (V + 1 b_v^T) = P V + b_v; fold it into the
        # output projection's bias (the transposed V^T = W_v X^T product has no per-column bias slot)
        bo = sd[a + "to_out.0.bias"].to(dev).float() + self.a_wo.float() @ bf(a + "to_v.bias").float()
        self.a_bo = bo.to(torch.bfloat16)
        self.s_up = []
        for i in range(4):
            res = [_Res(sd, f"{d}up_blocks.{i}.resnets.{j}", dev, False) for j in range(3)]
            upk = f"{d}up_blocks.{i}.upsamplers.0.conv"
            self.s_up.append((res, _Conv(sd, upk, dev) if (upk + ".weight") in sd else None))
        self.s_norm = _Norm(sd, d + "conv_norm_out", dev, 1e-6)
        self.s_out = _Conv(sd, d + "conv_out", dev, n_pad=128)

    def _init_encoders(self, sd, dev):
        """Encode side: diffusers vae.Encoder + quant_conv under ``spatial_vae.module.`` and Encoder (:177-272) + quant_conv under
        ``temporal_vae.``.  The two 8-channel heads (conv_out / conv2, then the 1x1 quant convs) are padded to the 128-column tile."""
        e = "spatial_vae.module.encoder."
        w_in = sd[e + "conv_in.weight"].to(dev)                                     # [128, 3, 3, 3]: RGB + one zero channel
        w_in = torch.cat([w_in, torch.zeros_like(w_in[:, :1])], 1)
        self.e_conv_in_w = _conv_w(w_in, None, 64)                                  # K = 9 * 4 = 36 -> 64
        self.e_conv_in_b = _vec(sd[e + "conv_in.bias"].to(dev))
        self.e_down = []
        for i in range(4):
            res = [_Res(sd, f"{e}down_blocks.{i}.resnets.{j}", dev, False) for j in range(2)]
            dk = f"{e}down_blocks.{i}.downsamplers.0.conv"
            self.e_down.append((res, _Conv(sd, dk, dev) if (dk + ".weight") in sd else None))
        self.e_mid = [_Res(sd, f"{e}mid_block.resnets.{i}", dev, False) for i in range(2)]
        self.e_attn = self._attn_weights(sd, e + "mid_block.attentions.0.", dev)
        self.e_norm = _Norm(sd, e + "conv_norm_out", dev, 1e-6)
        self.e_out = _Conv(sd, e + "conv_out", dev, n_pad=128)
        q = "spatial_vae.module.quant_conv"
        self.e_quant_w = _conv_w(sd[q + ".weight"].to(dev), 128, 128)
        self.e_quant_b = _vec(sd[q + ".bias"].to(dev), 128)
        t = "temporal_vae.encoder."
        self.te_conv_in_w = _conv_w(sd[t + "conv_in.conv.weight"].to(dev), None, 128)   # K = 27 * 4 = 108 -> 128, no bias
        self.te_blocks = [[_Res(sd, f"{t}block_res_blocks.{i}.{j}", dev, True) for j in range(4)] for i in range(4)]
        self.te_down = {i: _Conv(sd, f"{t}conv_blocks.{i}.conv", dev) for i in range(3) if f"{t}conv_blocks.{i}.conv.weight" in sd}
        self.te_res = [_Res(sd, f"{t}res_blocks.{i}", dev, True) for i in range(4)]
        self.te_norm = _Norm(sd, t + "norm1", dev, 1e-5)
        self.te_out = _Conv(sd, t + "conv2.conv", dev, n_pad=128)                       # 1x1x1, 512 -> 8
        q = "temporal_vae.quant_conv.conv"
        self.te_quant_w = _conv_w(sd[q + ".weight"].to(dev), 128, 128)
        self.te_quant_b = _vec(sd[q + ".bias"].to(dev), 128)

    @staticmethod
    def _attn_weights(sd, a, dev):
        """The single-head mid-block attention of the 2-D VAE: (norm, wq, bq, wk, bk, wv, wo, bo') with the value bias folded
        into the output bias (softmax rows sum to 1: P (V + 1 b_v^T) = P V + b_v)."""
        bf = lambda k: sd[k].to(dev).to(torch.bfloat16).contiguous()
        wo = bf(a + "to_out.0.weight")
        bo = sd[a + "to_out.0.bias"].to(dev).float() + wo.float() @ bf(a + "to_v.bias").float()
        return SimpleNamespace(a_norm=_Norm(sd, a + "group_norm", dev, 1e-6), a_wq=bf(a + "to_q.weight"), a_bq=bf(a + "to_q.bias"),
                               a_wk=bf(a + "to_k.weight"), a_bk=bf(a + "to_k.bias"), a_wv=bf(a + "to_v.weight"), a_wo=wo,
                               a_bo=bo.to(torch.bfloat16))

    # ------------------------------------------------------------------------------------------------ sizes
    def get_temporal_latent_size(self, t: int) -> int:
        pad = 0 if t % self.time_downsample_factor == 0 else self.time_downsample_factor - t % self.time_downsample_factor
        return (t + pad) // self.time_downsample_factor

    def get_latent_size(self, input_size):
        """VideoAutoencoderPipeline.get_latent_size (autoencoder_kl_open_sora.py:704-716) for (T, H, W)."""
        T, H, W = input_size
        h, w = (H // 8 if H is not None else None), (W // 8 if W is not None else None)
        if self.micro_frame_size is None or T is None:
            return [None if T is None else self.get_temporal_latent_size(T), h, w]
        t = self.get_temporal_latent_size(self.micro_frame_size) * (T // self.micro_frame_size)
        if T % self.micro_frame_size > 0:
            t += self.get_temporal_latent_size(T % self.micro_frame_size)
        return [t, h, w]

    # ------------------------------------------------------------------------------------------------ building blocks
    def clear_cache(self):
        """Drop the zero-bordered staging buffers (they are kept per geometry; call after changing resolution to free HBM)."""
        self._padded.clear()

    def _padded_buf(self, g: VaeGrid, C: int):
        """Zero-bordered conv-input buffer for grid g, allocated once per geometry (kernels only ever write its interior)."""
        key = (g.n, g.T, g.H, g.W, g.tf, C)
        hit = self._padded.get(key)
        if hit is None:
            hit = g.alloc(C, self.device, zero=True)
            self._padded[key] = hit
        return hit[1]

    def _norm_act(self, x, gx: VaeGrid, norm: _Norm, C: int, tf: int, silu=True, dense=False):
        gd = VaeGrid(gx.n, gx.T, gx.H, gx.W, 0 if dense else 1, 0 if dense else tf)
        y = torch.empty(gd.rows, C, dtype=torch.bfloat16, device=self.device) if dense else self._padded_buf(gd, C)
        ops.group_norm(x, gx, y, gd, C, norm.g, norm.b, norm.eps, silu)
        return y, gd

    def _resblock(self, x, gx: VaeGrid, r: _Res):
        """x rows over gx (any layout) -> rows over the conv-output grid (pad 1, tf 0)."""
        tf = r.c1.kt - 1
        h, gh = self._norm_act(x, gx, r.n1, r.c1.cin, tf)
        y = ops.conv(h, gh, r.c1.w, r.c1.b, r.c1.cin, r.c1.kt, r.c1.ks)
        go = gh.conv_out()
        h2, gh2 = self._norm_act(y, go, r.n2, r.c2.cin, tf)
        if gx.pad != 1 or gx.tf != 0:   # the residual is added row-for-row in the conv-output layout
            xr = torch.empty(go.rows, x.shape[1], dtype=torch.bfloat16, device=self.device)
            ops.regrid(x, gx, xr, go, x.shape[1])
            x = xr
        res = x
        if r.sc is not None:
            res = ops.gemm128(x, r.sc.w, r.sc.b)
        out = ops.conv(h2, gh2, r.c2.w, r.c2.b, r.c2.cin, r.c2.kt, r.c2.ks, res=res)
        return out, go

    # ------------------------------------------------------------------------------------------------ temporal VAE
    def _temporal_decode(self, z4: torch.Tensor, num_frames: int, out: torch.Tensor, f0: int) -> int:
        """z4: planar bf16 [4, Tz, H, W] (already z*scale + shift'ed inside the first kernel) -> writes the decoded latent
        frames into out[4, :, H, W] from frame f0; returns the number of frames written.  VAE_Temporal.decode :453-462."""
        _, Tz, H, W = z4.shape
        tpad = 0 if num_frames % 4 == 0 else 4 - num_frames % 4
        params = list(self.scale) + list(self.shift) + self.t_pq[0].flatten().tolist() + self.t_pq[1].tolist()
        a = ops.vae_first_im2col(z4, 3, 128, params)
        g = VaeGrid(1, Tz, H, W, 0, 0)
        x = ops.gemm128(a, self.t_conv1_w, self.t_conv1_b)
        for r in self.t_res:
            x, g = self._resblock(x, g, r)
        for i in (3, 2, 1, 0):
            for r in self.t_blocks[i]:
                x, g = self._resblock(x, g, r)
            if i > 0 and (i - 1) in self.t_up:
                cv = self.t_up[i - 1]
                gp = VaeGrid(1, g.T, g.H, g.W, 1, 2)
                xp = self._padded_buf(gp, cv.cin)
                ops.regrid(x, g, xp, gp, cv.cin)
                y = ops.conv(xp, gp, cv.w, cv.b, cv.cin, 3, 3)
                g2 = VaeGrid(1, 2 * g.T, g.H, g.W, 1, 0)
                x = torch.empty(g2.rows, cv.cout // 2, dtype=torch.bfloat16, device=self.device)
                ops.d2s_time(y, gp.conv_out(), x, g2, cv.cout // 2)
                g = g2
        h, gh = self._norm_act(x, g, self.t_norm, 128, 2)
        y = ops.conv(h, gh, self.t_out.w, self.t_out.b, 128, 3, 3)
        ops.extract_planar(y, gh.conv_out(), 4, tpad, out, f0)
        return g.T - tpad

    # ------------------------------------------------------------------------------------------------ 2-D decoder
    def _attention(self, x, g: VaeGrid, aw=None):
        """diffusers Attention (1 head, d = C = 512) with residual (``aw``: another mid block's weights, default the decoder's); x rows over g -> dense rows whose per-frame stride is the
        token count rounded up to the 128-column GEMM tile (pad rows are zero on input, finite junk on output)."""
        C, L, n = 512, g.H * g.W, g.n
        Lp = (L + 127) // 128 * 128
        gdn = VaeGrid(n, 1, g.H, g.W, 0, 0, sample_rows=Lp)
        key = ("attn", n, g.H, g.W)
        bufs = self._padded.get(key)
        if bufs is None:   # pad rows must be zero (and stay zero: kernels write the L interior rows only)
            bufs = (torch.zeros(gdn.rows, C, dtype=torch.bfloat16, device=self.device),
                    torch.zeros(gdn.rows, C, dtype=torch.bfloat16, device=self.device))
            self._padded[key] = bufs
        hn, xd = bufs
        A = aw if aw is not None else self
        ops.group_norm(x, g, hn, gdn, C, A.a_norm.g, A.a_norm.b, A.a_norm.eps, False)
        ops.regrid(x, g, xd, gdn, C)
        q = ops.gemm128(hn, A.a_wq, A.a_bq)
        k = ops.gemm128(hn, A.a_wk, A.a_bk)
        vt = torch.empty(n, C, Lp, dtype=torch.bfloat16, device=self.device)   # V^T per frame = W_v X^T (pad columns = 0)
        ops.gemm128(A.a_wv, hn.view(n, Lp, C), out=vt, batch=n, batch_a=0, batch_w=Lp * C, batch_o=C * Lp, M=C)
        o = torch.empty(n * Lp, C, dtype=torch.bfloat16, device=self.device)
        step = max(1, min(n, (1 << 28) // (Lp * Lp)))          # <= 1 GiB of fp32 scores at a time
        for f in range(0, n, step):
            m = min(step, n - f)
            s = torch.empty(m, Lp, Lp, dtype=torch.float32, device=self.device)
            ops.gemm128(q[f * Lp:(f + m) * Lp].view(m, Lp, C), k[f * Lp:(f + m) * Lp].view(m, Lp, C), out_f32=s,
                        out_scale=1.0 / math.sqrt(C), batch=m, batch_a=Lp * C, batch_w=Lp * C, batch_o=Lp * Lp, M=Lp)
            p = ops.softmax_rows(s, n=L)
            ops.gemm128(p, vt[f:f + m], out=o[f * Lp:(f + m) * Lp].view(m, Lp, C), batch=m, batch_a=Lp * Lp, batch_w=C * Lp,
                        batch_o=Lp * C, M=Lp)
        return ops.gemm128(o, A.a_wo, A.a_bo, res=xd), gdn

    def _spatial_decode(self, xz: torch.Tensor, out: torch.Tensor, f0: int, in_scale: float = 1.0 / _SD_SCALE):
        """xz planar bf16 [4, F, H, W] -> out[3, f0:f0+F, 8H, 8W].  VideoAutoencoderKL.decode :522-538 + diffusers Decoder."""
        _, F, H, W = xz.shape
        params = [in_scale] * 4 + [0.0] * 4 + self.s_pq[0].flatten().tolist() + self.s_pq[1].tolist()
        a = ops.vae_first_im2col(xz, 1, 64, params)
        g = VaeGrid(F, 1, H, W, 0, 0)
        x = ops.gemm128(a, self.s_conv_in_w, self.s_conv_in_b)
        x, g = self._resblock(x, g, self.s_mid[0])
        x, g = self._attention(x, g)
        x, g = self._resblock(x, g, self.s_mid[1])
        for res, up in self.s_up:
            for r in res:
                x, g = self._resblock(x, g, r)
            if up is not None:
                gp = VaeGrid(F, 1, 2 * g.H, 2 * g.W, 1, 0)
                xp = self._padded_buf(gp, up.cin)
                ops.regrid(x, g, xp, gp, up.cin, up=1)
                x = ops.conv(xp, gp, up.w, up.b, up.cin, 1, 3)
                g = gp.conv_out()
        h, gh = self._norm_act(x, g, self.s_norm, 128, 0)
        y = ops.conv(h, gh, self.s_out.w, self.s_out.b, 128, 1, 3)
        ops.extract_planar(y, gh.conv_out(), 3, 0, out, f0)

    # ------------------------------------------------------------------------------------------------ public
    @torch.no_grad()
    def decode(self, z: torch.Tensor, num_frames: int, frames: Optional[tuple] = None) -> torch.Tensor:
        """z [B, 4, Tz, H, W] -> video [B, 3, num_frames, 8H, 8W] bf16 (autoencoder_kl_open_sora.py:672-695).
        ``frames = (f0, f1)``: only the output frames f0 <= f < f1 -> [B, 3, f1 - f0, 8H, 8W] — the temporal VAE runs for the
        micro-frame chunks that hold them (the chunks are independent, :680-690), the 2-D decoder for those frames alone (it is per
        frame, :522-538): the same values as the slice of the full decode, bit for bit."""
        if z.device.type != self.device.type:      # (self.device is a HIP device: the constructor refuses anything else)
            raise RuntimeError("OpenSoraVAE.decode needs a HIP device tensor (no CPU path)")
        B, C, Tz, H, W = z.shape
        assert C == 4
        f_lo, f_hi = (0, num_frames) if frames is None else (int(frames[0]), int(frames[1]))
        if not (0 <= f_lo <= f_hi <= num_frames):
            raise ValueError(f"frames {frames} outside [0, {num_frames}]")
        outs = []
        for b in range(B):
            vid = torch.empty(3, f_hi - f_lo, 8 * H, 8 * W, dtype=torch.bfloat16, device=self.device)
            if f_hi == f_lo:
                outs.append(vid)
                continue
            zb = z[b].to(torch.bfloat16).contiguous()
            if self.micro_frame_size is None:
                chunks = [(0, Tz, 0, num_frames)]
            else:       # chunk c: latent frames [c * mz, (c + 1) * mz) -> output frames [c * mf, min((c + 1) * mf, num_frames))
                mf, mz = self.micro_frame_size, self.micro_z_frame_size
                chunks = [(i, min(i + mz, Tz), (i // mz) * mf, min((i // mz + 1) * mf, num_frames)) for i in range(0, Tz, mz)]
            chunks = [c for c in chunks if c[3] > f_lo and c[2] < f_hi]          # the chunks that hold a wanted frame
            base = chunks[0][2]
            xz = torch.empty(4, chunks[-1][3] - base, H, W, dtype=torch.bfloat16, device=self.device)
            for z0, z1, o0, o1 in chunks:
                wrote = self._temporal_decode(zb[:, z0:z1].contiguous(), o1 - o0, xz, o0 - base)
                assert wrote == o1 - o0, (wrote, o0, o1)
            if frames is None:
                assert chunks[-1][3] == num_frames, (chunks[-1][3], num_frames)
            for f in range(f_lo, f_hi, self.frames_per_launch):
                m = min(self.frames_per_launch, f_hi - f)
                self._spatial_decode(xz[:, f - base:f - base + m].contiguous(), vid, f - f_lo)
            outs.append(vid)
        return torch.stack(outs, 0)

    def frame_shards(self, num_frames: int, P: int) -> list:
        """The block of output frames every rank of ``P`` decodes: [(f0, f1)] * P, contiguous, in rank order.  A rank pays one temporal
        decode (the whole 17-frame micro chunk, ~4 frames' worth of 2-D decoding at 512 x 512) per chunk its block touches, so with at
        least as many ranks as chunks the blocks are cut INSIDE chunks — ranks are dealt to the chunks by frame count, then a chunk's
        frames are split evenly over its ranks (64 frames over 8 ranks: 9, 8 | 9, 8 | 9, 8 | 7, 6 — no rank decodes two chunks);
        with fewer ranks than chunks: equal contiguous blocks."""
        mf = self.micro_frame_size
        per = -(-num_frames // P)
        even = [(min(r * per, num_frames), min((r + 1) * per, num_frames)) for r in range(P)]
        if mf is None or num_frames <= mf:
            return even
        chunks = [(c0, min(c0 + mf, num_frames)) for c0 in range(0, num_frames, mf)]
        if P < len(chunks):
            return even
        ranks = [1] * len(chunks)
        for _ in range(P - len(chunks)):          # the next rank goes to the chunk whose ranks hold the most frames each
            k = max(range(len(chunks)), key=lambda i: (chunks[i][1] - chunks[i][0]) / ranks[i])
            ranks[k] += 1
        out = []
        for (c0, c1), n in zip(chunks, ranks):
            q = -(-(c1 - c0) // n)
            out += [(min(c0 + i * q, c1), min(c0 + (i + 1) * q, c1)) for i in range(n)]
        return out

    def frame_shard(self, num_frames: int, P: int, rank: int) -> tuple:
        return self.frame_shards(num_frames, P)[rank]

    @torch.no_grad()
    def decode_sharded(self, z: torch.Tensor, num_frames: int, group, to_uint8: bool = True) -> torch.Tensor:
        """decode() with the output frames sharded over the ranks of ``group`` (dsp.py group protocol: RCCL on a GPU node) and ONE
        all-gather of the finished frames.  The reference decodes the whole video redundantly on every rank
        (pipeline_open_sora.py:648-656 after autoencoder_kl_open_sora.py:672-695); once the denoising loop is sequence parallel the
        unsharded decode is the largest serial term of a video.  ``to_uint8``: gather the frames as uint8 [B, F, 8H, 8W, 3] in the
        pipeline's output convention ((x.clamp(-1, 1) / 2 + 0.5) * 255 rounded, :648-656) — 1 byte per value on the wire; else
        bf16 [B, 3, F, 8H, 8W].  Every rank returns the whole video; frames are bit-identical to decode()'s."""
        from . import dsp

        P, r = dsp.group_size(group), dsp.group_rank(group)
        shards = self.frame_shards(num_frames, P)
        f0, f1 = shards[r]
        per = max(b - a for a, b in shards)                                 # every rank's piece of the gather has this many frames
        part = self.decode(z, num_frames, frames=(f0, f1))                 # [B, 3, n, 8H, 8W]
        B, _, n, Hh, Ww = part.shape
        if to_uint8:
            mine = torch.zeros(B, per, Hh, Ww, 3, dtype=torch.uint8, device=self.device)
            mine[:, :n] = pixels_to_uint8(part)
        else:
            mine = torch.zeros(B, 3, per, Hh, Ww, dtype=torch.bfloat16, device=self.device)
            mine[:, :, :n] = part
        # (the gathered tensor is the ranks' pieces stacked along dim 0 — the shape every backend's all_gather_into_tensor accepts)
        flat = torch.empty((P * mine.shape[0],) + tuple(mine.shape[1:]), dtype=mine.dtype, device=self.device)
        dsp.all_gather_into_tensor(flat, mine, group)
        allp = flat.view((P,) + tuple(mine.shape))
        if to_uint8:                                                        # [P, B, per, H, W, 3] -> [B, F, H, W, 3]
            return torch.cat([allp[q, :, :b - a] for q, (a, b) in enumerate(shards)], dim=1)
        return torch.cat([allp[q, :, :, :b - a] for q, (a, b) in enumerate(shards)], dim=2)


    # ------------------------------------------------------------------------------------------------ encode
    _IDENT = [1.0] * 4 + [0.0] * 4 + [1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0] + [0.0] * 4   # first-layer params: x as is

    def _strided_conv(self, x, g: VaeGrid, cv: "_Conv", t_stride: int, s_stride: int):
        """3x3(x3) convolution with stride (t_stride, s_stride, s_stride) in the encoders' padding conventions: the stride-1
        convolution over the zero-bordered grid, sampled at the odd positions (module docstring)."""
        tf = cv.kt - 1
        gp = VaeGrid(g.n, g.T, g.H, g.W, 1, tf)
        xp = self._padded_buf(gp, cv.cin)
        ops.regrid(x, g, xp, gp, cv.cin)
        y = ops.conv(xp, gp, cv.w, cv.b, cv.cin, cv.kt, cv.ks)
        g2 = VaeGrid(g.n, g.T // t_stride, g.H // s_stride, g.W // s_stride, 0, 0)
        out = torch.empty(g2.rows, cv.cout, dtype=torch.bfloat16, device=self.device)
        ops.subsample(y, gp.conv_out(), out, g2, cv.cout, t_stride, s_stride, t_stride - 1, s_stride - 1)
        return out, g2

    def _spatial_encode(self, pix: torch.Tensor) -> torch.Tensor:
        """pix planar bf16 [3, F, H, W] in [-1, 1] -> moments planar bf16 [8, F, H/8, W/8] (mean | logvar) of the 2-D VAE:
        diffusers AutoencoderKL.encode up to the distribution, VideoAutoencoderKL.encode :503-520."""
        _, F, H, W = pix.shape
        z4 = torch.cat([pix, torch.zeros_like(pix[:1])], 0).contiguous()
        a = ops.vae_first_im2col(z4, 1, 64, self._IDENT)
        g = VaeGrid(F, 1, H, W, 0, 0)
        x = ops.gemm128(a, self.e_conv_in_w, self.e_conv_in_b)
        for res, down in self.e_down:
            for r in res:
                x, g = self._resblock(x, g, r)
            if down is not None:
                x, g = self._strided_conv(x, g, down, 1, 2)
        x, g = self._resblock(x, g, self.e_mid[0])
        x, g = self._attention(x, g, self.e_attn)
        x, g = self._resblock(x, g, self.e_mid[1])
        h, gh = self._norm_act(x, g, self.e_norm, 512, 0)
        y = ops.conv(h, gh, self.e_out.w, self.e_out.b, 512, 1, 3)           # 8 of 128 columns carry the head
        m = ops.gemm128(y, self.e_quant_w, self.e_quant_b)                  # quant_conv (1x1) on those 8
        out = torch.empty(8, F, H // 8, W // 8, dtype=torch.bfloat16, device=self.device)
        ops.extract_planar(m, gh.conv_out(), 4, 0, out[:4], 0)               # mean   (the kernel moves <= 4 channels a call)
        ops.extract_planar(m[:, 4:], gh.conv_out(), 4, 0, out[4:], 0)        # logvar
        return out

    def _temporal_encode(self, xz: torch.Tensor) -> torch.Tensor:
        """xz planar bf16 [4, T, h, w] (T <= micro_frame_size frames of 2-D latents) -> moments planar bf16 [8, ceil(T/4), h, w]:
        VAE_Temporal.encode :441-451 up to the distribution (zero frames in FRONT up to a multiple of 4, Encoder :258-272)."""
        _, T, H, W = xz.shape
        tpad = 0 if T % 4 == 0 else 4 - T % 4
        if tpad:
            xz = torch.cat([torch.zeros(4, tpad, H, W, dtype=xz.dtype, device=xz.device), xz], 1)
        Tp = T + tpad
        a = ops.vae_first_im2col(xz.contiguous(), 3, 128, self._IDENT)
        g = VaeGrid(1, Tp, H, W, 0, 0)
        x = ops.gemm128(a, self.te_conv_in_w, None)
        for i in range(4):
            for r in self.te_blocks[i]:
                x, g = self._resblock(x, g, r)
            if i in self.te_down:
                x, g = self._strided_conv(x, g, self.te_down[i], 2, 1)
        for r in self.te_res:
            x, g = self._resblock(x, g, r)
        h, gh = self._norm_act(x, g, self.te_norm, 512, 0, dense=True)
        y = ops.gemm128(h, self.te_out.w, self.te_out.b)
        m = ops.gemm128(y, self.te_quant_w, self.te_quant_b)
        out = torch.empty(8, g.T, H, W, dtype=torch.bfloat16, device=self.device)
        ops.extract_planar(m, gh, 4, 0, out[:4], 0)
        ops.extract_planar(m[:, 4:], gh, 4, 0, out[4:], 0)
        return out

    @staticmethod
    def _sample(moments: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
        """DiagonalGaussianDistribution.sample (:21-40) on planar moments [8, ...] with the noise handed in -> fp32 [4, ...]."""
        mean, logvar = moments[:4].float(), moments[4:].float()
        return mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * noise

    @torch.no_grad()
    def encode(self, x: torch.Tensor, noise_fn: Optional[Callable] = None, frames_per_launch: Optional[int] = None) -> torch.Tensor:
        """video [B, 3, T, H, W] in [-1, 1] -> normalised latents [B, 4, Tz, H/8, W/8] fp32 (autoencoder_kl_open_sora.py:653-670,
        cal_loss False).  ``noise_fn(shape)`` supplies the standard-normal draws of the two posteriors ([B', 4, h, w] per 2-D
        micro batch of ``micro_batch_size`` frames, then [B, 4, tz, h, w] per temporal micro batch — the reference's order);
        default torch.randn on the CPU generator."""
        if not self.has_encoder:
            raise RuntimeError("this OpenSoraVAE was built from a checkpoint without encoder weights (decode only)")
        if not x.is_cuda:
            raise RuntimeError("OpenSoraVAE.encode needs a HIP device tensor (no CPU path)")
        return self._encode(x, noise_fn)

    def _encode(self, x, noise_fn):
        B, C, T, H, W = x.shape
        assert C == 3 and H % 8 == 0 and W % 8 == 0, (C, H, W)
        noise_fn = noise_fn or (lambda shape: torch.randn(shape, dtype=torch.float32))
        dev = self.device
        h, w = H // 8, W // 8
        # 2-D VAE over the (B T) frames in micro batches (:509-518)
        fr = x.to(torch.bfloat16).permute(0, 2, 1, 3, 4).reshape(B * T, 3, H, W)
        mb = self.micro_batch_size or B * T
        lat = torch.empty(B * T, 4, h, w, dtype=torch.float32, device=dev)
        for i in range(0, B * T, mb):
            n = min(mb, B * T - i)
            m = self._spatial_encode(fr[i:i + n].permute(1, 0, 2, 3).contiguous())             # [8, n, h, w]
            nz = noise_fn((n, 4, h, w)).to(dev, torch.float32).permute(1, 0, 2, 3)
            lat[i:i + n] = (self._sample(m, nz) * _SD_SCALE).permute(1, 0, 2, 3)
        x_z = lat.view(B, T, 4, h, w).permute(0, 2, 1, 3, 4).to(torch.bfloat16)                  # [B, 4, T, h, w]
        # temporal VAE per micro batch of frames (:659-665)
        step = self.micro_frame_size or T
        zs = []
        for i in range(0, T, step):
            ms = torch.stack([self._temporal_encode(x_z[b, :, i:i + step].contiguous()) for b in range(B)], 0)   # [B, 8, tz, h, w]
            nz = noise_fn((B, 4, ms.shape[2], h, w)).to(dev, torch.float32)
            zs.append(torch.stack([self._sample(ms[b], nz[b]) for b in range(B)], 0))
        z = torch.cat(zs, dim=2)
        scale = torch.tensor(self.scale, device=dev)[None, :, None, None, None]
        shift = torch.tensor(self.shift, device=dev)[None, :, None, None, None]
        return (z - shift) / scale

    __call__ = decode


def pixels_to_uint8(video: torch.Tensor) -> torch.Tensor:
    """[B, 3, F, H, W] in [-1, 1] -> uint8 [B, F, H, W, 3]: the output convention of OpenSoraPipeline.generate
    (pipeline_open_sora.py:648-656: clamp, scale to 0..255, add 0.5, clamp, cast)."""
    return (video.clamp(-1, 1) * 0.5 + 0.5).mul(255).add_(0.5).clamp_(0, 255).permute(0, 2, 3, 4, 1).to(torch.uint8)


def OpenSoraVAE_V1_2(micro_batch_size=4, micro_frame_size=17, from_pretrained=None, freeze_vae_2d=False, cal_loss=False, device="cuda"):
    """autoencoder_kl_open_sora.py:731-761: the Open-Sora 1.2 video VAE (SDXL 2-D VAE + VAE_Temporal_SD, the published latent shift /
    scale).  ``from_pretrained``: a LOCAL checkpoint directory holding ``model.safetensors`` with the reference's keys (the hub
    download of the reference is not available offline), ``"synthetic:<seed>"`` (decoder only) or ``"synthetic-full:<seed>"``
    (with the encoders, for image / video conditioning).  ``freeze_vae_2d`` / ``cal_loss`` are training switches with nothing to
    act on in an inference build."""
    from .utils import read_component

    name = from_pretrained
    if isinstance(name, str) and name.startswith(("synthetic:", "synthetic-full:")):
        sd = synth_state_dict(int(name.split(":", 1)[1]), encoder=name.startswith("synthetic-full:"))
    else:
        sd = read_component(name)[1]
    if sd is None:
        raise FileNotFoundError(f"OpenSoraVAE_V1_2(from_pretrained={name!r}): not a local checkpoint directory (model.safetensors) and "
                                "not 'synthetic[-full]:<seed>' — hub ids cannot be fetched on this box")
    return OpenSoraVAE(sd, device=device, micro_frame_size=micro_frame_size, micro_batch_size=micro_batch_size)


class AutoencoderKLDecoder(OpenSoraVAE):
    """Per-frame decode with a diffusers ``AutoencoderKL`` (SD / SDXL VAE geometry, state-dict keys ``decoder.*``,
    ``post_quant_conv.*``) — what LattePipeline.decode_latents does with ``enable_vae_temporal_decoder=False``
    (pipeline_latte.py:916-927: latents / scaling_factor, every frame through ``vae.decode``, ``(x / 2 + 0.5).clamp(0, 1) * 255``
    as uint8 [b, f, h, w, c]).  The 2-D decoder kernels are the ones of OpenSoraVAE."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda", scaling_factor: float = 0.18215, frames_per_launch: int = 16):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("videosys_amd.AutoencoderKLDecoder needs a HIP device (no CPU path)")
        self.device = dev
        self.frames_per_launch = frames_per_launch
        self.scaling_factor = scaling_factor
        self._padded = {}
        self._init_spatial(state_dict, dev, "")

    @torch.no_grad()
    def decode(self, latents: torch.Tensor) -> torch.Tensor:
        """latents [B, 4, F, H, W] -> sample [B, 3, F, 8H, 8W] bf16 (before the pipeline's /2 + 0.5)."""
        if not latents.is_cuda:
            raise RuntimeError("AutoencoderKLDecoder.decode needs a HIP device tensor (no CPU path)")
        B, C, Fr, H, W = latents.shape
        outs = []
        for b in range(B):
            xz = latents[b].to(torch.bfloat16).contiguous()
            vid = torch.empty(3, Fr, 8 * H, 8 * W, dtype=torch.bfloat16, device=self.device)
            for f in range(0, Fr, self.frames_per_launch):
                m = min(self.frames_per_launch, Fr - f)
                self._spatial_decode(xz[:, f:f + m].contiguous(), vid, f, 1.0 / self.scaling_factor)
            outs.append(vid)
        return torch.stack(outs, 0)

    def decode_latents(self, latents: torch.Tensor) -> torch.Tensor:
        """pipeline_latte.py:916-927 -> uint8 [b, f, h, w, c] on the CPU."""
        v = self.decode(latents).float()
        return ((v / 2.0 + 0.5).clamp(0, 1) * 255).permute(0, 2, 3, 4, 1).to(dtype=torch.uint8).cpu().contiguous()

    __call__ = decode_latents


# ---------------------------------------------------------------------------------------------------- synthetic weights
def decoder_param_shapes() -> Dict[str, tuple]:
    """Names and shapes of every decode-side parameter of the reference VideoAutoencoderPipeline (checked against the reference's
    own state_dict in tests/test_vae_cpu.py)."""
    p: Dict[str, tuple] = {}

    def conv3(name, ci, co, k, bias):
        p[name + ".conv.weight"] = (co, ci, k, k, k)
        if bias:
            p[name + ".conv.bias"] = (co,)

    def norm(name, c):
        p[name + ".weight"] = (c,)
        p[name + ".bias"] = (c,)

    def res3(name, ci, co):
        norm(name + ".norm1", ci); conv3(name + ".conv1", ci, co, 3, False)
        norm(name + ".norm2", co); conv3(name + ".conv2", co, co, 3, False)
        if ci != co:
            conv3(name + ".conv3", ci, co, 1, False)

    t = "temporal_vae."
    conv3(t + "post_quant_conv", 4, 4, 1, True)
    d = t + "decoder."
    conv3(d + "conv1", 4, 512, 3, True)
    for i in range(4):
        res3(f"{d}res_blocks.{i}", 512, 512)
    prev = 512
    for i in (3, 2, 1, 0):
        f = 128 * (1, 2, 2, 4)[i]
        for j in range(4):
            res3(f"{d}block_res_blocks.{i}.{j}", prev, f)
            prev = f
        if i > 0 and (False, True, True)[i - 1]:
            conv3(f"{d}conv_blocks.{i - 1}", prev, prev * 2, 3, True)
    norm(d + "norm1", 128)
    conv3(d + "conv_out", 128, 4, 3, True)

    def conv2(name, ci, co, k):
        p[name + ".weight"] = (co, ci, k, k)
        p[name + ".bias"] = (co,)

    def res2(name, ci, co):
        norm(name + ".norm1", ci); conv2(name + ".conv1", ci, co, 3)
        norm(name + ".norm2", co); conv2(name + ".conv2", co, co, 3)
        if ci != co:
            conv2(name + ".conv_shortcut", ci, co, 1)

    s = "spatial_vae.module."
    conv2(s + "post_quant_conv", 4, 4, 1)
    d = s + "decoder."
    conv2(d + "conv_in", 4, 512, 3)
    res2(d + "mid_block.resnets.0", 512, 512)
    a = d + "mid_block.attentions.0."
    norm(a + "group_norm", 512)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        p[a + n + ".weight"] = (512, 512)
        p[a + n + ".bias"] = (512,)
    res2(d + "mid_block.resnets.1", 512, 512)
    prev = 512
    for i, co in enumerate((512, 512, 256, 128)):
        for j in range(3):
            res2(f"{d}up_blocks.{i}.resnets.{j}", prev, co)
            prev = co
        if i < 3:
            conv2(f"{d}up_blocks.{i}.upsamplers.0.conv", co, co, 3)
    norm(d + "conv_norm_out", 128)
    conv2(d + "conv_out", 128, 3, 3)
    return p


def encoder_param_shapes() -> Dict[str, tuple]:
    """Names and shapes of every encode-side parameter of the reference VideoAutoencoderPipeline (checked against the reference's
    own state_dict in tests/test_vae_cpu.py): the diffusers vae.Encoder + quant_conv and the temporal Encoder (:177-256) +
    quant_conv."""
    p: Dict[str, tuple] = {}

    def norm(name, c):
        p[name + ".weight"] = (c,)
        p[name + ".bias"] = (c,)

    def conv2(name, ci, co, k):
        p[name + ".weight"] = (co, ci, k, k)
        p[name + ".bias"] = (co,)

    def res2(name, ci, co):
        norm(name + ".norm1", ci); conv2(name + ".conv1", ci, co, 3)
        norm(name + ".norm2", co); conv2(name + ".conv2", co, co, 3)
        if ci != co:
            conv2(name + ".conv_shortcut", ci, co, 1)

    s = "spatial_vae.module."
    e = s + "encoder."
    conv2(e + "conv_in", 3, 128, 3)
    prev = 128
    for i, co in enumerate((128, 256, 512, 512)):
        for j in range(2):
            res2(f"{e}down_blocks.{i}.resnets.{j}", prev, co)
            prev = co
        if i < 3:
            conv2(f"{e}down_blocks.{i}.downsamplers.0.conv", co, co, 3)
    res2(e + "mid_block.resnets.0", 512, 512)
    a = e + "mid_block.attentions.0."
    norm(a + "group_norm", 512)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        p[a + n + ".weight"] = (512, 512)
        p[a + n + ".bias"] = (512,)
    res2(e + "mid_block.resnets.1", 512, 512)
    norm(e + "conv_norm_out", 512)
    conv2(e + "conv_out", 512, 8, 3)
    conv2(s + "quant_conv", 8, 8, 1)

    def conv3(name, ci, co, k, bias):
        p[name + ".conv.weight"] = (co, ci, k, k, k)
        if bias:
            p[name + ".conv.bias"] = (co,)

    def res3(name, ci, co):
        norm(name + ".norm1", ci); conv3(name + ".conv1", ci, co, 3, False)
        norm(name + ".norm2", co); conv3(name + ".conv2", co, co, 3, False)
        if ci != co:
            conv3(name + ".conv3", ci, co, 1, False)

    t = "temporal_vae."
    e = t + "encoder."
    conv3(e + "conv_in", 4, 128, 3, False)
    prev = 128
    for i in range(4):
        f = 128 * (1, 2, 2, 4)[i]
        for j in range(4):
            res3(f"{e}block_res_blocks.{i}.{j}", prev, f)
            prev = f
        if i < 3 and (False, True, True)[i]:
            conv3(f"{e}conv_blocks.{i}", prev, f, 3, True)
    for i in range(4):
        res3(f"{e}res_blocks.{i}", 512, 512)
    norm(e + "norm1", 512)
    conv3(e + "conv2", 512, 8, 1, True)
    conv3(t + "quant_conv", 8, 8, 1, True)
    return p


def synth_state_dict(seed: int = 0, encoder: bool = False) -> Dict[str, torch.Tensor]:
    """Deterministic random weights (bf16-representable fp32): conv / linear weights N(0, 1/fan_in), biases N(0, 0.02), norm
    scales 1 + N(0, 0.1), norm shifts N(0, 0.1).  No checkpoint can be fetched here (no network).  The decode-side values do
    not depend on ``encoder`` (the encode-side ones come from their own generator)."""
    sd = _synth(decoder_param_shapes(), torch.Generator().manual_seed(seed))
    if encoder:
        sd.update(_synth(encoder_param_shapes(), torch.Generator().manual_seed(seed + 7919)))
    return sd


def _synth(shapes, g) -> Dict[str, torch.Tensor]:
    sd = {}
    for k, shp in shapes.items():
        is_norm = ".norm" in k or "group_norm" in k or "conv_norm_out" in k
        if k.endswith(".weight") and len(shp) >= 2:
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
