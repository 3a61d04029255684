"""CogVideoX transformer denoise step on MI355X — host mirror of
videosys/models/transformers/cogvideox_transformer_3d.py (CogVideoXTransformer3DModel :315-589, CogVideoXBlock :179-312,
CogVideoXAttnProcessor2_0 :35-175) with modules/normalization.py:36-114 and modules/embeddings.py:14-51,283-412.

Same constructor kwargs/defaults, same ``forward(hidden_states [B, F, C, H, W], encoder_hidden_states, timestep,
timestep_cond=None, image_rotary_emb=None, return_dict=True)`` and ``[B, F, C_out, H, W]`` result (fp32 here; the
pipeline casts to fp32 right after, pipeline_cogvideox.py:699), same state-dict key names as THUDM/CogVideoX-2b / -5b
(transformer/).  Every tensor op of the per-step path is a call into libvideosys_amd.so:

  * the joint [text | video] sequence lives in ONE buffer [B, Lt + Lv, C]; CogVideoXLayerNormZero's two modulation sets
    and the two gates are applied by row segment inside the LayerNorm-modulate kernel and the GEMM epilogue, so the
    reference's per-block torch.cat / split (:96,:171-174,:298) never happens;
  * to_q / to_k / to_v are one [3C, C] GEMM; LayerNorm qk-norm and the rotary embedding run inside the attention kernels
    (K side in vsys_attn_prep_kv64, Q side in the flash prologue);
  * all 2 L + 1 ``linear(silu(temb))`` modulation rows of a step come from one launch;
  * CogVideoXPatchEmbed's Conv2d is an im2col + MFMA GEMM writing straight into the video rows of the joint buffer.
PAB (spatial only, CogVideoXPABConfig) caches the un-gated attention output and re-gates it on broadcast steps like the
reference (:276-289).  Not built this round (raise): Ulysses sequence parallelism (:112-165), cp batch split.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, Optional

import numpy as np
import torch

from . import ops, pab
from .utils import same_tensor


def _sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    omega = np.arange(embed_dim // 2, dtype=np.float64) / (embed_dim / 2.0)
    omega = 1.0 / 10000**omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def cogvideox_pos_embed_3d(embed_dim, w, h, t, spatial_scale, temporal_scale) -> torch.Tensor:
    """diffusers get_3d_sincos_pos_embed as the reference builds its pos_embedding buffer (:430-439): constant table
    [t*h*w, embed_dim] = [temporal D/4 | spatial 3D/4], built on the host once and uploaded."""
    ds, dt = 3 * embed_dim // 4, embed_dim // 4
    gh = np.arange(h, dtype=np.float32) / spatial_scale
    gw = np.arange(w, dtype=np.float32) / spatial_scale
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape([2, 1, h, w])
    pos_s = np.concatenate([_sincos_1d(ds // 2, grid[0]), _sincos_1d(ds // 2, grid[1])], axis=1)
    pos_t = _sincos_1d(dt, np.arange(t, dtype=np.float32) / temporal_scale)
    pos_s = np.repeat(pos_s[np.newaxis], t, axis=0)
    pos_t = np.repeat(pos_t[:, np.newaxis], w * h, axis=1)
    return torch.from_numpy(np.concatenate([pos_t, pos_s], axis=-1)).float().flatten(0, 1)


class CogVideoXTransformer3DModel:
    def __init__(self, num_attention_heads=30, attention_head_dim=64, in_channels=16, out_channels=16, flip_sin_to_cos=True,
                 freq_shift=0, time_embed_dim=512, text_embed_dim=4096, num_layers=30, dropout=0.0, attention_bias=True,
                 sample_width=90, sample_height=60, sample_frames=49, patch_size=2, temporal_compression_ratio=4,
                 max_text_seq_length=226, activation_fn="gelu-approximate", timestep_activation_fn="silu",
                 norm_elementwise_affine=True, norm_eps=1e-5, spatial_interpolation_scale=1.875,
                 temporal_interpolation_scale=1.0, use_rotary_positional_embeddings=False, device="cuda",
                 dtype=torch.bfloat16):
        from . import _lib

        _lib.load()  # fail loudly if the HIP library is missing
        if attention_head_dim != 64:
            raise ValueError("the CogVideoX attention kernels are built for head_dim 64")
        if (activation_fn != "gelu-approximate" or not norm_elementwise_affine or not attention_bias or not flip_sin_to_cos
                or freq_shift != 0 or timestep_activation_fn != "silu"):
            raise NotImplementedError("only the THUDM/CogVideoX-2b / -5b transformer configuration")
        if dtype != torch.bfloat16:
            raise ValueError("the MI355X path computes in bf16 (fp32 accumulate); CogVideoX-2b's fp16 checkpoints load as bf16")
        C = num_attention_heads * attention_head_dim
        if C % 192 or C % 64:
            raise ValueError("hidden size must be a multiple of 192 (GEMM tile); 1920 and 3072 are")
        self.config = SimpleNamespace(num_attention_heads=num_attention_heads, attention_head_dim=attention_head_dim,
                                      in_channels=in_channels, out_channels=out_channels, time_embed_dim=time_embed_dim,
                                      text_embed_dim=text_embed_dim, num_layers=num_layers, sample_width=sample_width,
                                      sample_height=sample_height, sample_frames=sample_frames, patch_size=patch_size,
                                      temporal_compression_ratio=temporal_compression_ratio,
                                      max_text_seq_length=max_text_seq_length, norm_eps=norm_eps,
                                      use_rotary_positional_embeddings=use_rotary_positional_embeddings)
        self.H, self.C, self.L = num_attention_heads, C, num_layers
        self.device, self.dtype = torch.device(device), dtype
        self.w: Dict[str, torch.Tensor] = {}
        self.parallel_manager = SimpleNamespace(sp_size=1, cp_size=1, dp_size=1, dp_rank=0, sp_group=None, cp_group=None)
        self._ws = {}
        self._kbounds = {}   # per block: the promise about its key norms (ops.ln_key_bound), None = none
        self._hidden_tap = None
        pf = (sample_frames - 1) // temporal_compression_ratio + 1
        self._pos3d = None
        if not use_rotary_positional_embeddings:
            self._pos3d = cogvideox_pos_embed_3d(C, sample_width // patch_size, sample_height // patch_size, pf,
                                                 spatial_interpolation_scale, temporal_interpolation_scale).to(
                device=self.device, dtype=dtype).contiguous()
        self.attn_count = [0] * num_layers
        self.last_attn = [None] * num_layers
        self._rope_cache = None

    # ------------------------------------------------------------------ weights
    def expected_keys(self):
        keys = []
        for l in ("patch_embed.proj", "patch_embed.text_proj", "time_embedding.linear_1", "time_embedding.linear_2",
                  "norm_final", "norm_out.linear", "norm_out.norm", "proj_out"):
            keys += [l + ".weight", l + ".bias"]
        for i in range(self.L):
            p = f"transformer_blocks.{i}"
            for l in ("norm1.linear", "norm1.norm", "norm2.linear", "norm2.norm", "attn1.norm_q", "attn1.norm_k", "attn1.to_q",
                      "attn1.to_k", "attn1.to_v", "attn1.to_out.0", "ff.net.0.proj", "ff.net.2"):
                keys += [f"{p}.{l}.weight", f"{p}.{l}.bias"]
        return keys

    def _kbound(self, pre):
        """What lets the long-sequence attention kernel drop the running max (vsys_flash_attn_d64_kb): from the block's norm_q /
        norm_k weights, once.  VSYS_FLASH_STATIC=0 never promises."""
        import os

        if pre not in self._kbounds:
            kb = None
            if os.environ.get("VSYS_FLASH_STATIC", "1") != "0":
                g = self.w.get
                kb = ops.ln_key_bound(g(pre + ".attn1.norm_q.weight"), g(pre + ".attn1.norm_q.bias"), g(pre + ".attn1.norm_k.weight"),
                                      g(pre + ".attn1.norm_k.bias"))
            self._kbounds[pre] = kb
        return self._kbounds[pre]

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        self._kbounds = {}
        missing = [k for k in self.expected_keys() if k not in sd]
        if strict and missing:
            raise KeyError(f"missing keys: {missing[:8]}{'...' if len(missing) > 8 else ''}")
        dev = lambda t: t.detach().to(device=self.device, dtype=self.dtype).contiguous()
        C = self.C
        for k in self.expected_keys():
            if k in sd:
                t = sd[k]
                if k == "patch_embed.proj.weight":
                    t = t.reshape(t.shape[0], -1)
                self.w[k] = dev(t)
        mods_w, mods_b = [], []
        for i in range(self.L):
            p = f"transformer_blocks.{i}"
            self.w[p + ".attn1.qkv.weight"] = dev(torch.cat([sd[f"{p}.attn1.{l}.weight"] for l in ("to_q", "to_k", "to_v")], 0))
            self.w[p + ".attn1.qkv.bias"] = dev(torch.cat([sd[f"{p}.attn1.{l}.bias"] for l in ("to_q", "to_k", "to_v")], 0))
            for n in ("norm1", "norm2"):
                mods_w.append(sd[f"{p}.{n}.linear.weight"])
                mods_b.append(sd[f"{p}.{n}.linear.bias"])
        # every linear(silu(temb)) of a step in one matrix: [2L blocks x 6C | norm_out 2C (padded to 6C)] x time_embed_dim
        pad_w = torch.zeros(6 * C, sd["norm_out.linear.weight"].shape[1])
        pad_w[: 2 * C] = sd["norm_out.linear.weight"]
        pad_b = torch.zeros(6 * C)
        pad_b[: 2 * C] = sd["norm_out.linear.bias"]
        self.w["_mod.weight"] = dev(torch.cat(mods_w + [pad_w], 0))
        self.w["_mod.bias"] = dev(torch.cat(mods_b + [pad_b], 0))
        # proj_out is [p*p*Cout, C] = 64 rows: pad to one 192-wide MFMA column tile (extra rows are zero, never read back)
        po = torch.zeros(192, C)
        po[: sd["proj_out.weight"].shape[0]] = sd["proj_out.weight"]
        pb = torch.zeros(192)
        pb[: sd["proj_out.bias"].shape[0]] = sd["proj_out.bias"]
        self.w["_proj_out.weight"], self.w["_proj_out.bias"] = dev(po), dev(pb)
        return self

    def enable_parallel(self, dp_size=None, sp_size=None, enable_cp=None, parallel_mgr=None, copy_executor=None):
        """cogvideox_transformer_3d.py:466-477.  Video tokens are sharded at rest (text rows replicated, :530-532); attention
        exchanges sequence for heads (Ulysses, :112-123,160-165).  cp (CFG batch split) is not built: accepted and ignored."""
        from . import dsp

        self.parallel_manager = parallel_mgr if parallel_mgr is not None else dsp.ParallelManager(dp_size or 1, 1, sp_size or 1)
        P = self.parallel_manager.sp_size
        if P > 1:
            if self.H % P:
                raise ValueError(f"Number of heads {self.H} must be divisible by sequence parallel size {P}")
            kw = {} if copy_executor is None else {"copy_executor": copy_executor}
            self._sp = dsp.UlyssesParallel(self.parallel_manager.sp_group, **kw)
        else:
            self._sp = None

    def reset_pab_state(self):
        self.attn_count = [0] * self.L

    def _buf(self, name, shape, dtype=None):
        n = int(np.prod(shape))
        b = self._ws.get(name)
        if b is None or b.numel() < n:
            b = torch.empty(n, dtype=dtype or self.dtype, device=self.device)
            self._ws[name] = b
        return b[:n].view(*shape)

    def _rope(self, image_rotary_emb):
        if image_rotary_emb is None:
            return None, None
        cos, sin = image_rotary_emb
        # identity + version of the table tensors (strong references held): two resolutions can share a shape and, once the first
        # table is freed, an address
        c = self._rope_cache
        if c is None or not same_tensor(c[0], cos) or not same_tensor(c[1], sin) or c[2] != (cos._version, sin._version):
            c = (cos, sin, (cos._version, sin._version), cos.to(device=self.device, dtype=torch.float32).contiguous(),
                 sin.to(device=self.device, dtype=torch.float32).contiguous())
            self._rope_cache = c
        return c[3], c[4]

    def reset_text_cache(self):
        self._rope_cache = None

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, hidden_states, encoder_hidden_states, timestep, timestep_cond=None, image_rotary_emb=None,
                return_dict: bool = True):
        if timestep_cond is not None:
            raise NotImplementedError("timestep_cond is not used by the CogVideoX pipelines")
        w, C, H, cfgm = self.w, self.C, self.H, self.config
        p = cfgm.patch_size
        Bz, Fr, cin, Hh, Ww = hidden_states.shape
        B, Lt, _ = encoder_hidden_states.shape  # CFG batch; a latent batch of B/2 is read twice by the im2col kernel
        if B % Bz:
            raise ValueError("batch sizes of latents and encoder_hidden_states do not agree")
        Hp, Wp = Hh // p, Ww // p
        Lv = Fr * Hp * Wp
        L = Lt + Lv          # rows of a sample in the attention
        sp = getattr(self, "_sp", None)
        Lvl = Lv if sp is None else sp.shard_len(Lv)   # video rows of a sample held by this rank
        v0 = 0 if sp is None else sp.rank * Lvl
        nv = max(0, min(Lvl, Lv - v0))                # ... of which real (the rest is set_pad("pad") zero padding, :531)
        Ll = Lt + Lvl        # rows of a sample at rest
        Hl = H if sp is None else H // sp.P
        dev = self.device
        # 1. time embedding and every modulation row of the step
        ts_host = torch.as_tensor(timestep).detach().to("cpu").float().reshape(-1)
        if ts_host.numel() != B:
            ts_host = ts_host.repeat(B // ts_host.numel())
        f = ops.timestep_embedding(ts_host.to(dev).contiguous(), C)
        e1 = ops.linear_small(f, w["time_embedding.linear_1.weight"], w["time_embedding.linear_1.bias"], act_out=ops.ACT_SILU)
        emb = ops.linear_small(e1, w["time_embedding.linear_2.weight"], w["time_embedding.linear_2.bias"])
        mod = ops.linear_small(emb, w["_mod.weight"], w["_mod.bias"], act_in=ops.ACT_SILU)  # [B, (2L + 1) * 6C]
        ms = mod.shape[1]
        # 2. patch embedding straight into the joint buffer [B, Lt + Lv, C]
        x = self._buf("x", (B * Ll, C))
        txt = encoder_hidden_states.to(device=dev, dtype=self.dtype).reshape(B * Lt, -1).contiguous()
        cols = ops.im2col_patch(hidden_states.to(device=dev, dtype=torch.float32).contiguous(), B, p)
        if nv < Lvl:
            x.view(B, Ll, C)[:, Lt + nv:].zero_()
        for b in range(B):
            ops.gemm(txt[b * Lt:(b + 1) * Lt], w["patch_embed.text_proj.weight"], w["patch_embed.text_proj.bias"],
                     out=x[b * Ll: b * Ll + Lt])
            if nv:
                ops.gemm(cols[b * Lv + v0: b * Lv + v0 + nv], w["patch_embed.proj.weight"], w["patch_embed.proj.bias"],
                         out=x[b * Ll + Lt: b * Ll + Lt + nv])
                if not cfgm.use_rotary_positional_embeddings:
                    ops.add_bcast_rows(x[b * Ll + Lt: b * Ll + Lt + nv], self._pos3d[v0:v0 + nv], 1, nv)
        cos, sin = self._rope(image_rotary_emb if cfgm.use_rotary_positional_embeddings else None)
        use_pab = pab.enable_pab()
        timestep_int = int(ts_host[0]) if use_pab else None
        kp, vt = self._kv(B, L, Hl)
        C6 = 6 * C
        hw = Hl * 64
        for i in range(self.L):
            pre = f"transformer_blocks.{i}"
            m1 = mod[:, (2 * i) * C6:(2 * i + 1) * C6]  # shift, scale, gate, enc_shift, enc_scale, enc_gate
            m2 = mod[:, (2 * i + 1) * C6:(2 * i + 2) * C6]
            bc = False
            if use_pab:
                bc, self.attn_count[i] = pab.if_broadcast_spatial(timestep_int, self.attn_count[i])
            if not bc:
                xm = ops.ln_modulate(x, w[pre + ".norm1.norm.weight"], w[pre + ".norm1.norm.bias"], m1[0, 0:C], m1[0, C:2 * C], Ll,
                                     mod_stride=ms, seg_split=Lt, mod_alt=3 * C, eps=cfgm.norm_eps, out=self._buf("xm", (B * Ll, C)))
                qkv = ops.gemm(xm, w[pre + ".attn1.qkv.weight"], w[pre + ".attn1.qkv.bias"], out=self._buf("qkv", (B * Ll, 3 * C)))
                if sp is not None:  # sequence -> heads: this rank's H/P heads over the whole [text | video] sequence
                    qkv = sp.scatter_heads(qkv, B, Lt, Lv, C, out=self._buf("qkv_h", (B, L, 3 * hw)))
                ops.attn_prep_kv64(qkv[:, hw:2 * hw], qkv[:, 2 * hw:], w[pre + ".attn1.norm_k.weight"], w[pre + ".attn1.norm_k.bias"],
                                   cos, sin, Lt, kp, vt, B, Hl, L)
                ao = self._buf("attn_out_h" if sp is not None else "attn_out", (B * L, hw))
                ops.flash_attn64(qkv[:, :hw], w[pre + ".attn1.norm_q.weight"], w[pre + ".attn1.norm_q.bias"], cos, sin, Lt, kp, vt,
                                 ao, B, Hl, L, L, k_norm_bound=self._kbound(pre))
                if sp is not None:  # heads -> sequence
                    ao = sp.gather_heads(ao, B, Lt, Lv, C, out=self._buf("attn_out", (B, Ll, C)))
            if use_pab:
                # the cache holds the UN-gated attention output (:284-286); it is re-gated with this step's gate (:288-289)
                if not bc:
                    if self.last_attn[i] is None or self.last_attn[i].shape != x.shape:
                        self.last_attn[i] = torch.empty_like(x)
                    ops.gemm(ao, w[pre + ".attn1.to_out.0.weight"], w[pre + ".attn1.to_out.0.bias"], out=self.last_attn[i])
                ops.gate_add_rows(x, self.last_attn[i], m1[0, 2 * C:3 * C], Ll, ms, Lt, 3 * C)
            else:
                ops.gemm_gate2(ao, w[pre + ".attn1.to_out.0.weight"], w[pre + ".attn1.to_out.0.bias"], m1[0, 2 * C:3 * C], ms, Ll, Lt,
                               3 * C, res=x, out=x)
            xm = ops.ln_modulate(x, w[pre + ".norm2.norm.weight"], w[pre + ".norm2.norm.bias"], m2[0, 0:C], m2[0, C:2 * C], Ll,
                                 mod_stride=ms, seg_split=Lt, mod_alt=3 * C, eps=cfgm.norm_eps, out=self._buf("xm", (B * Ll, C)))
            hb = ops.gemm(xm, w[pre + ".ff.net.0.proj.weight"], w[pre + ".ff.net.0.proj.bias"], epilogue=ops.EPI_BIAS_GELU,
                          out=self._buf("mlp_h", (B * Ll, w[pre + ".ff.net.0.proj.weight"].shape[0])))
            ops.gemm_gate2(hb, w[pre + ".ff.net.2.weight"], w[pre + ".ff.net.2.bias"], m2[0, 2 * C:3 * C], ms, Ll, Lt, 3 * C, res=x,
                           out=x)
            if self._hidden_tap is not None:   # test hook (tests/fulldepth_util.py): joint hidden state after block i
                self._hidden_tap(i, x.view(B, Ll, C))
        # 3. norm_final -> norm_out (AdaLayerNorm, chunk_dim=1: shift, scale) -> proj_out -> unpatchify, video rows only
        mo = mod[:, 2 * self.L * C6:]
        xv = self._buf("xm", (B * Lvl, C))
        for b in range(B):
            ops.ln_modulate(x[b * Ll + Lt:(b + 1) * Ll], w["norm_final.weight"], w["norm_final.bias"], None, None, Lvl,
                            eps=cfgm.norm_eps, out=xv[b * Lvl:(b + 1) * Lvl])
        xo = ops.ln_modulate(xv, w["norm_out.norm.weight"], w["norm_out.norm.bias"], mo[0, 0:C], mo[0, C:2 * C], Lvl, mod_stride=ms,
                             eps=cfgm.norm_eps, out=self._buf("attn_out", (B * Lvl, C)))
        po = ops.gemm(xo, w["_proj_out.weight"], w["_proj_out.bias"], out=self._buf("proj", (B * Lvl, 192)))
        if sp is not None:  # gather_sequence (:569-570) on the projected rows (192 columns instead of C), padding dropped
            from . import dsp

            parts = torch.empty(sp.P, B, Lvl, 192, dtype=po.dtype, device=dev)
            dsp.all_gather_into_tensor(parts.view(sp.P * B, Lvl, 192), po.view(B, Lvl, 192).contiguous(), sp.group)   # (group protocol)
            po = parts.permute(1, 0, 2, 3).reshape(B, sp.P * Lvl, 192)[:, :Lv].reshape(B * Lv, 192).contiguous()
        out = ops.unpatchify_cvx(po, B, Fr, Hp, Wp, cfgm.out_channels, p)
        if not return_dict:
            return (out,)
        return SimpleNamespace(sample=out)

    __call__ = forward

    def _kv(self, batch, kv_len, heads=None):
        heads = heads or self.H
        key = ("kv", batch, kv_len, heads)
        if key not in self._ws:
            self._ws[key] = ops.alloc_kv_buffers64(batch, heads, kv_len, self.device)
        return self._ws[key]


def synth_state_dict(num_layers=30, num_heads=30, head_dim=64, text_embed_dim=4096, in_channels=16, out_channels=16,
                     time_embed_dim=512, patch_size=2, seed: int = 777) -> Dict[str, torch.Tensor]:
    """Seeded random weights with the THUDM/CogVideoX transformer key names (no pretrained weights offline)."""
    g = torch.Generator().manual_seed(seed)
    C = num_heads * head_dim
    sd: Dict[str, torch.Tensor] = {}

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
