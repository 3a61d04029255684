"""Open-Sora v1.2 VAE decode on MI355X.

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

There is no CPU path: without the HIP library / a GPU every call raises.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

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
    def _attention(self, x, g: VaeGrid):
        """diffusers Attention (1 head, d = C = 512) with residual; x rows over g -> dense rows whose per-frame stride is the
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
        ops.group_norm(x, g, hn, gdn, C, self.a_norm.g, self.a_norm.b, self.a_norm.eps, False)
        ops.regrid(x, g, xd, gdn, C)
        q = ops.gemm128(hn, self.a_wq, self.a_bq)
        k = ops.gemm128(hn, self.a_wk, self.a_bk)
        vt = torch.empty(n, C, Lp, dtype=torch.bfloat16, device=self.device)   # V^T per frame = W_v X^T (pad columns = 0)
        ops.gemm128(self.a_wv, hn.view(n, Lp, C), out=vt, batch=n, batch_a=0, batch_w=Lp * C, batch_o=C * Lp, M=C)
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
        return ops.gemm128(o, self.a_wo, self.a_bo, res=xd), gdn

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
    def decode(self, z: torch.Tensor, num_frames: int) -> torch.Tensor:
        """z [B, 4, Tz, H, W] -> video [B, 3, num_frames, 8H, 8W] bf16 (autoencoder_kl_open_sora.py:672-695)."""
        if not z.is_cuda:
            raise RuntimeError("OpenSoraVAE.decode needs a HIP device tensor (no CPU path)")
        B, C, Tz, H, W = z.shape
        assert C == 4
        outs = []
        for b in range(B):
            zb = z[b].to(torch.bfloat16).contiguous()
            xz = torch.empty(4, num_frames, H, W, dtype=torch.bfloat16, device=self.device)
            left, f0 = num_frames, 0
            step = self.micro_z_frame_size if self.micro_frame_size is not None else Tz
            for i in range(0, Tz, step):
                nf = min(self.micro_frame_size, left) if self.micro_frame_size is not None else left
                f0 += self._temporal_decode(zb[:, i:i + step].contiguous(), nf, xz, f0)
                left -= self.micro_frame_size if self.micro_frame_size is not None else left
            assert f0 == num_frames, (f0, num_frames)
            vid = torch.empty(3, num_frames, 8 * H, 8 * W, dtype=torch.bfloat16, device=self.device)
            for f in range(0, num_frames, self.frames_per_launch):
                m = min(self.frames_per_launch, num_frames - f)
                self._spatial_decode(xz[:, f:f + m].contiguous(), vid, f)
            outs.append(vid)
        return torch.stack(outs, 0)

    __call__ = decode


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


def synth_state_dict(seed: int = 0) -> Dict[str, torch.Tensor]:
    """Deterministic random decode-side weights (bf16-representable fp32): conv / linear weights N(0, 1/fan_in), biases
    N(0, 0.02), norm scales 1 + N(0, 0.1), norm shifts N(0, 0.1).  No checkpoint can be fetched here (no network)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in decoder_param_shapes().items():
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
