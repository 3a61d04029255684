"""Latte (LatteT2V) denoise step on MI355X — host mirror of videosys/models/transformers/latte_transformer_3d.py
(LatteT2V :893-1470, BasicTransformerBlock :178-517, BasicTransformerBlock_ :520-843, AdaLayerNormSingle :846-878).

Same constructor kwargs (the subset the Latte-1 checkpoint uses: ``norm_type="ada_norm_single"``, patch input,
``activation_fn="gelu-approximate"``, ``attention_bias=True``), same ``forward(hidden_states, timestep, all_timesteps,
encoder_hidden_states, added_cond_kwargs, ..., encoder_attention_mask, ..., enable_temporal_attentions, return_dict)``
signature and ``[B, out_channels, F, H, W]`` result, same state-dict key names as ``maxin-cn/Latte-1`` (transformer/) —
every tensor op of the per-step path is a call into libvideosys_amd.so.  head_dim is 72 (16 x 72 = 1152), so the
STDiT3 kernels are reused as they are: flash_attn_d72 for the spatial and the cross attention (no qk-norm), the
frame-strided temporal kernel without norm / RoPE, the AdaLN-modulate kernel and the fused-epilogue GEMMs.

Differences that do not change results:
  * attn1's to_q/to_k/to_v are one [3C, C] GEMM; attn2's to_k/to_v run once per prompt (the text is constant over the
    steps) and their attention layouts are cached per block;
  * the ``(b f) t d <-> (b t) f d`` rearranges around every temporal block (:1391,:1425) are never materialised;
  * the cross-attention mask (additive -10000 on padded text tokens, :1245-1248) is applied as a per-sample key length:
    exp(-10000) underflows to exactly 0 in fp32, so excluding those keys is the same softmax.  Masks that are not a
    prefix of ones raise NotImplementedError.
Unsupported (raise): norm types other than ada_norm_single, image joint training (use_image_num), cp batch split.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, Optional

import numpy as np
import torch

from . import dsp, ops, pab
from .utils import same_tensor


def _sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    """diffusers get_1d_sincos_pos_embed_from_grid ([sin | cos], float64 frequencies) — constant table."""
    omega = np.arange(embed_dim // 2, dtype=np.float64) / (embed_dim / 2.0)
    omega = 1.0 / 10000**omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def latte_pos_embed_2d(embed_dim: int, gh: int, gw: int, base_size: int, interpolation_scale: float = 1.0) -> torch.Tensor:
    """diffusers get_2d_sincos_pos_embed as PatchEmbed calls it (pos_embed, latte_transformer_3d.py:1032-1039): constant
    per resolution, built on the host like the reference does and uploaded once."""
    grid_h = np.arange(gh, dtype=np.float32) / (gh / base_size) / interpolation_scale
    grid_w = np.arange(gw, dtype=np.float32) / (gw / base_size) / interpolation_scale
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape([2, 1, gw, gh])
    emb = np.concatenate([_sincos_1d(embed_dim // 2, grid[0]), _sincos_1d(embed_dim // 2, grid[1])], axis=1)
    return torch.from_numpy(emb).float()


class _BlockState:
    def __init__(self, block_idx, temporal):
        self.block_idx, self.temporal = block_idx, temporal
        self.attn_count = self.cross_count = self.mlp_count = 0
        self.last_attn = self.last_cross = None


class LatteT2V:
    """Drop-in for the reference LatteT2V at the operator boundary ``transformer(latent_model_input, ...)[0]``."""

    def __init__(self, num_attention_heads=16, attention_head_dim=72, in_channels=4, out_channels=8, num_layers=28,
                 cross_attention_dim=1152, attention_bias=True, sample_size=64, patch_size=2,
                 activation_fn="gelu-approximate", norm_type="ada_norm_single", norm_elementwise_affine=False,
                 norm_eps=1e-6, caption_channels=4096, video_length=16, device="cuda", dtype=torch.bfloat16, **unused):
        from . import _lib

        _lib.load()  # fail loudly if the HIP library is missing
        if norm_type != "ada_norm_single" or activation_fn != "gelu-approximate" or norm_elementwise_affine or not attention_bias:
            raise NotImplementedError("only the Latte-1 configuration (ada_norm_single, gelu-approximate, biased attention)")
        if attention_head_dim != ops.HEAD_DIM:
            raise ValueError("attention kernels are built for head_dim 72")
        if dtype != torch.bfloat16:
            raise ValueError("the MI355X path computes in bf16 (fp32 accumulate)")
        self.config = SimpleNamespace(num_attention_heads=num_attention_heads, attention_head_dim=attention_head_dim,
                                      in_channels=in_channels, out_channels=out_channels, num_layers=num_layers,
                                      cross_attention_dim=cross_attention_dim, sample_size=sample_size, patch_size=patch_size,
                                      norm_type=norm_type, norm_eps=norm_eps, caption_channels=caption_channels,
                                      video_length=video_length)
        self.H, self.C = num_attention_heads, num_attention_heads * attention_head_dim
        self.L = num_layers
        self.patch_size = patch_size
        self.in_channels, self.out_channels = in_channels, out_channels
        self.device, self.dtype = torch.device(device), dtype
        self.w: Dict[str, torch.Tensor] = {}
        self.parallel_manager = SimpleNamespace(sp_size=1, cp_size=1, dp_size=1, dp_rank=0, sp_group=None, cp_group=None)
        self._sp = None
        self.states = [_BlockState(i // 2, bool(i % 2)) for i in range(2 * num_layers)]
        self._pos_cache, self._ws = {}, {}
        self._text_cache = None
        tpe = _sincos_1d(self.C, np.arange(0, video_length)[:, None].astype(np.float64))  # :1129-1130
        self.temp_pos_embed = torch.from_numpy(tpe).float().to(device=self.device, dtype=dtype).contiguous()

    # ------------------------------------------------------------------ weights
    def block_prefix(self, i):
        return f"{'temporal_transformer_blocks' if i % 2 else 'transformer_blocks'}.{i // 2}"

    def expected_keys(self):
        keys = ["pos_embed.proj.weight", "pos_embed.proj.bias", "scale_shift_table", "proj_out.weight", "proj_out.bias"]
        for l in ("adaln_single.emb.timestep_embedder.linear_1", "adaln_single.emb.timestep_embedder.linear_2",
                  "adaln_single.linear", "caption_projection.linear_1", "caption_projection.linear_2"):
            keys += [l + ".weight", l + ".bias"]
        for i in range(2 * self.L):
            p = self.block_prefix(i)
            keys.append(p + ".scale_shift_table")
            for a in (("attn1",) if i % 2 else ("attn1", "attn2")):
                for l in ("to_q", "to_k", "to_v", "to_out.0"):
                    keys += [f"{p}.{a}.{l}.weight", f"{p}.{a}.{l}.bias"]
            for l in ("ff.net.0.proj", "ff.net.2"):
                keys += [f"{p}.{l}.weight", f"{p}.{l}.bias"]
        return keys

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        missing = [k for k in self.expected_keys() if k not in sd]
        if strict and missing:
            raise KeyError(f"missing keys: {missing[:8]}{'...' if len(missing) > 8 else ''}")
        dev = lambda t: t.detach().to(device=self.device, dtype=self.dtype).contiguous()
        for k in self.expected_keys():
            if k in sd:
                t = sd[k]
                if k == "pos_embed.proj.weight":
                    t = t.reshape(t.shape[0], -1)
                self.w[k] = dev(t)
        for i in range(2 * self.L):
            p = self.block_prefix(i)
            self.w[p + ".attn1.qkv.weight"] = dev(torch.cat([sd[f"{p}.attn1.{l}.weight"] for l in ("to_q", "to_k", "to_v")], 0))
            self.w[p + ".attn1.qkv.bias"] = dev(torch.cat([sd[f"{p}.attn1.{l}.bias"] for l in ("to_q", "to_k", "to_v")], 0))
            if not i % 2:
                self.w[p + ".attn2.kv.weight"] = dev(torch.cat([sd[f"{p}.attn2.{l}.weight"] for l in ("to_k", "to_v")], 0))
                self.w[p + ".attn2.kv.bias"] = dev(torch.cat([sd[f"{p}.attn2.{l}.bias"] for l in ("to_k", "to_v")], 0))
        tabs = [self.w[self.block_prefix(i) + ".scale_shift_table"].reshape(-1) for i in range(2 * self.L)]
        self.w["_all_tables"] = torch.stack(tabs).contiguous()
        self._text_cache = None
        return self

    def enable_parallel(self, dp_size=None, sp_size=None, enable_cp=None, parallel_mgr=None, copy_executor=None):
        """latte_transformer_3d.py:1127-1140.  Frames are sharded at rest (split_from_second_dim :1469-1475): spatial blocks
        and cross-attention need no exchange; each temporal block switches to the pixel shard around its attention
        (dynamic_switch :826-843).  cp (CFG batch split) is not built: enable_cp is accepted and ignored."""
        self.parallel_manager = parallel_mgr if parallel_mgr is not None else dsp.ParallelManager(dp_size or 1, 1, sp_size or 1)
        if self.parallel_manager.sp_size > 1:
            kw = {} if copy_executor is None else {"copy_executor": copy_executor}
            self._sp = dsp.SequenceParallel(self.parallel_manager.sp_group, **kw)
        else:
            self._sp = None

    # ------------------------------------------------------------------ helpers
    def _buf(self, name, shape):
        n = int(np.prod(shape))
        b = self._ws.get(name)
        if b is None or b.numel() < n:
            b = torch.empty(n, dtype=self.dtype, device=self.device)
            self._ws[name] = b
        return b[:n].view(*shape)

    def _gemm(self, x, wname, **kw):
        w, b = self.w[wname + ".weight"], self.w[wname + ".bias"]
        if w.shape[0] % 192 == 0 and w.shape[1] % 64 == 0:
            return ops.gemm(x, w, b, **kw)
        assert not kw or set(kw) <= {"out"}, "small shapes take the plain linear"
        return ops.linear_small(x, w, b, out=kw.get("out"))

    def _encode_text(self, y, mask, frames):
        # identity + version of the prompt tensors (strong references held): a recycled storage address is not the same prompt
        c = self._text_cache
        if (c is not None and same_tensor(c["y"], y) and c["y_version"] == y._version and same_tensor(c["mask"], mask)
                and (mask is None or c["mask_version"] == mask._version)):
            return c
        B, Lk, Cc = y.shape
        C, H = self.C, self.H
        if mask is None:
            lens = [Lk] * B
        else:
            m = mask.reshape(B, Lk).to("cpu") != 0
            lens = [int(v) for v in m.sum(dim=1).tolist()]
            for b in range(B):
                if lens[b] == 0 or not bool(m[b, : lens[b]].all()):
                    raise NotImplementedError("encoder_attention_mask must be a non-empty prefix of ones per sample")
        yb = y.to(device=self.device, dtype=self.dtype).reshape(B * Lk, Cc).contiguous()
        h = ops.linear_small(yb, self.w["caption_projection.linear_1.weight"], self.w["caption_projection.linear_1.bias"],
                             act_out=ops.ACT_GELU_TANH)
        ye = ops.linear_small(h, self.w["caption_projection.linear_2.weight"], self.w["caption_projection.linear_2.bias"])
        kv_pad = ops.kv_pad_len(Lk)
        kps = torch.zeros(self.L, B, H, kv_pad, ops.HEAD_DIM, dtype=self.dtype, device=self.device)
        vts = torch.zeros(self.L, B, H, ops.VT_ROWS, kv_pad, dtype=self.dtype, device=self.device)
        kv = torch.empty(B * Lk, 2 * C, dtype=self.dtype, device=self.device)
        for d in range(self.L):
            self._gemm(ye, f"transformer_blocks.{d}.attn2.kv", out=kv)
            ops.attn_prep_kv(kv[:, :C], kv[:, C:], None, kps[d], vts[d], B, H, Lk)
        self._text_cache = dict(y=y, y_version=y._version, mask=mask, mask_version=None if mask is None else mask._version,
                                lens=lens, kp=kps, vt=vts, Lk=Lk)
        return self._text_cache

    def reset_text_cache(self):
        self._text_cache = None

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, hidden_states, timestep=None, all_timesteps=None, encoder_hidden_states=None, added_cond_kwargs=None,
                class_labels=None, cross_attention_kwargs=None, attention_mask=None, encoder_attention_mask=None,
                use_image_num: int = 0, enable_temporal_attentions: bool = True, return_dict: bool = True):
        if use_image_num or attention_mask is not None or class_labels is not None:
            raise NotImplementedError("image joint training / self-attention masks / class labels are outside the hot path")
        w, C, H, p = self.w, self.C, self.H, self.patch_size
        Bz, cin, Fr, Hh, Ww = hidden_states.shape
        # B = the CFG batch; a latent batch of B/2 is read twice by the patch-embed kernel instead of torch.cat([z] * 2)
        B = encoder_hidden_states.shape[0]
        if B % Bz or timestep.numel() not in (B, Bz, 1):
            raise ValueError("batch sizes of latents / timestep / encoder_hidden_states do not agree")
        gh, gw = Hh // p, Ww // p
        S = gh * gw
        dev = self.device
        # AdaLayerNormSingle: embedded_timestep and the 6C modulation row; per-block tables added in one kernel
        ts_host = timestep.detach().to("cpu").float().reshape(-1)
        if ts_host.numel() != B:
            ts_host = ts_host.repeat(B // ts_host.numel())
        f = ops.timestep_embedding(ts_host.to(dev).contiguous(), 256)
        e1 = ops.linear_small(f, w["adaln_single.emb.timestep_embedder.linear_1.weight"],
                              w["adaln_single.emb.timestep_embedder.linear_1.bias"], act_out=ops.ACT_SILU)
        emb = ops.linear_small(e1, w["adaln_single.emb.timestep_embedder.linear_2.weight"],
                               w["adaln_single.emb.timestep_embedder.linear_2.bias"])  # embedded_timestep [B, C]
        t6 = ops.linear_small(emb, w["adaln_single.linear.weight"], w["adaln_single.linear.bias"], act_in=ops.ACT_SILU)
        mod = ops.mod_table(w["_all_tables"], t6)  # [2L, B, 6C]
        txt = self._encode_text(encoder_hidden_states, encoder_attention_mask, Fr)

        pkey = (gh, gw)
        if pkey not in self._pos_cache:
            interp = max(self.config.sample_size // 64, 1)
            self._pos_cache[pkey] = latte_pos_embed_2d(C, gh, gw, self.config.sample_size // p, interp).to(
                device=dev, dtype=self.dtype).contiguous()
        # (b f) frames of one sample are consecutive, exactly the [B, T, S, C] layout of the patch-embed kernel
        xz = hidden_states.to(device=dev, dtype=torch.float32)
        sp = self._sp
        Fa, f0 = Fr, 0          # frames in the (temporal) attention / first frame of this rank
        tpe = self.temp_pos_embed
        if sp is not None:
            # frame shard at rest: rank r owns frames [r*Tl, (r+1)*Tl), zero latents past the end (set_pad("temporal") :1301)
            Tl = -(-Fr // sp.P)
            f0 = sp.rank * Tl
            loc = torch.zeros(Bz, cin, Tl, Hh, Ww, dtype=torch.float32, device=dev)
            nv = max(0, min(Tl, Fr - f0))
            if nv:
                loc[:, :, :nv] = xz[:, :, f0:f0 + nv]
                tl = torch.zeros(Tl, C, dtype=tpe.dtype, device=dev)
                tl[:nv] = tpe[f0:f0 + nv]
            else:
                tl = torch.zeros(Tl, C, dtype=tpe.dtype, device=dev)
            xz, tpe, Fr = loc, tl, Tl
        x = ops.patch_embed(xz.contiguous(), w["pos_embed.proj.weight"], w["pos_embed.proj.bias"], self._pos_cache[pkey], B, (1, p, p), C)
        x = x.view(B * Fr * S, C)

        timestep_int = int(ts_host[0]) if pab.enable_pab() else None
        ats = None if all_timesteps is None else [int(v) for v in torch.as_tensor(all_timesteps).tolist()]
        for d in range(self.L):
            x = self._spatial_block(2 * d, x, mod[2 * d], txt, B, Fr, S, timestep_int, ats)
            if enable_temporal_attentions:
                if d == 0 and Fa > 1:
                    ops.add_bcast_rows(x, tpe, S, Fr)  # hidden + temp_pos_embed (:1410-1411)
                x = self._temporal_block(2 * d + 1, x, mod[2 * d + 1], B, Fr, S, timestep_int, ats, Fa)
            if getattr(self, "_hidden_tap", None) is not None:   # test hook: error growth per block pair
                self._hidden_tap(d, x)
        out = ops.final_layer(x, w["scale_shift_table"], emb, w["proj_out.weight"], w["proj_out.bias"], B, Fr, gh, gw, Hh, Ww,
                              (1, p, p), self.out_channels)
        if sp is not None:  # gather_from_second_dim (:1477-1482): frames of all ranks, time padding dropped
            import torch.distributed as dist

            parts = torch.empty(sp.P, *out.shape, dtype=out.dtype, device=dev)
            dist.all_gather_into_tensor(parts.view(sp.P * out.shape[0], *out.shape[1:]), out.contiguous(), group=sp.group)
            out = parts.permute(1, 2, 0, 3, 4, 5).reshape(B, self.out_channels, sp.P * Fr, Hh, Ww)[:, :, :Fa].contiguous()
        if not return_dict:
            return (out,)
        return SimpleNamespace(sample=out)

    __call__ = forward

    def _self_attn_out(self, i, x, mod_i, B, Fr, S, st, use_pab, temporal, Fa=None):
        """norm1 + modulate, attn1, gate, residual (+ PAB slab).  x: [B*Fr*S, C] rows ordered (b, f, s).  Under sequence
        parallelism Fr is the local frame count and Fa the global one: the temporal attention runs on the pixel shard."""
        w, C, H = self.w, self.C, self.H
        p = self.block_prefix(i)
        N, C6 = B * Fr * S, 6 * C
        shift, scale, gate = mod_i[0, 0:C], mod_i[0, C:2 * C], mod_i[0, 2 * C:3 * C]
        xm = ops.adaln_modulate(x, shift, scale, Fr * S, C6, eps=self.config.norm_eps, out=self._buf("xm", (N, C)))
        if temporal and self._sp is not None:
            # [B, Tl, S, C] -> all-to-all -> [B, Fa, Sl, C]: every frame of 1/P of the pixels (the modulated activations travel,
            # one C-wide tensor instead of q, k and v), attention, and back
            sp = self._sp
            Sl = -(-S // sp.P)
            xs = sp.to_spatial_shard(xm.view(B, Fr, S, C), Fa, Sl, out=self._buf("sp_x", (B, Fa, Sl, C)))
            Ns = B * Fa * Sl
            qkv = self._gemm(xs.view(Ns, C), p + ".attn1.qkv", out=self._buf("qkv", (Ns, 3 * C)))
            aos = self._buf("sp_ao", (Ns, C))
            ops.attn_temporal(qkv, C, None, None, None, None, aos, B, Fa, Sl, H)
            ao = sp.to_temporal_shard(aos.view(B, Fa, Sl, C), S, out=self._buf("attn_out", (B, Fr, S, C))).view(N, C)
            qkv = None
        else:
            qkv = self._gemm(xm, p + ".attn1.qkv", out=self._buf("qkv", (N, 3 * C)))
            ao = self._buf("attn_out", (N, C))
        if temporal and qkv is None:
            pass
        elif temporal:
            ops.attn_temporal(qkv, C, None, None, None, None, ao, B, Fr, S, H)
        else:
            key = ("kv_spatial", B * Fr, S)
            if key not in self._ws:
                self._ws[key] = ops.alloc_kv_buffers(B * Fr, H, S, self.device)
            kp, vt = self._ws[key]
            ops.attn_prep_kv(qkv[:, C:2 * C], qkv[:, 2 * C:], None, kp, vt, B * Fr, H, S)
            ops.flash_attn(qkv[:, :C], None, kp, vt, ao, B * Fr, H, S, S)
        aux = None
        if use_pab:
            if st.last_attn is None or st.last_attn.shape != x.shape:
                st.last_attn = torch.empty_like(x)
            aux = st.last_attn
        ops.gemm(ao, w[p + ".attn1.to_out.0.weight"], w[p + ".attn1.to_out.0.bias"], epilogue=ops.EPI_GATE_RES, gate=gate,
                 gate_stride=C6, rows_per_sample=Fr * S, res=x, aux=aux, out=x)

    def _ff(self, i, x, mod_i, B, Fr, S, st, timestep_int, ats, temporal):
        w, C = self.w, self.C
        p = self.block_prefix(i)
        N, C6 = B * Fr * S, 6 * C
        use_pab = pab.enable_pab()
        broadcast_mlp, broadcast_next, rng = False, False, None
        if use_pab:
            broadcast_mlp, st.mlp_count, broadcast_next, rng = pab.if_broadcast_mlp(timestep_int, st.mlp_count, st.block_idx,
                                                                                    ats, is_temporal=temporal)
        if broadcast_mlp:
            slab = pab.get_mlp_output(rng, timestep=timestep_int, block_idx=st.block_idx, is_temporal=temporal)
            ops.add_rows(x, slab)
            if timestep_int == rng[-1]:   # window closed: hand the slab back (stream order keeps the add above ahead of a reuse)
                self._ws.setdefault("mlp_slab_pool", []).append(slab)
            return
        shift, scale, gate = mod_i[0, 3 * C:4 * C], mod_i[0, 4 * C:5 * C], mod_i[0, 5 * C:6 * C]
        xm = ops.adaln_modulate(x, shift, scale, Fr * S, C6, eps=self.config.norm_eps, out=self._buf("xm", (N, C)))
        hb = ops.gemm(xm, w[p + ".ff.net.0.proj.weight"], w[p + ".ff.net.0.proj.bias"], epilogue=ops.EPI_BIAS_GELU,
                      out=self._buf("mlp_h", (N, 4 * C)))
        aux = self._mlp_slab(x) if broadcast_next else None
        ops.gemm(hb, w[p + ".ff.net.2.weight"], w[p + ".ff.net.2.bias"], epilogue=ops.EPI_GATE_RES, gate=gate, gate_stride=C6,
                 rows_per_sample=Fr * S, res=x, aux=aux, out=x)
        if broadcast_next:
            pab.save_mlp_output(timestep=timestep_int, block_idx=st.block_idx, ff_output=aux, is_temporal=temporal)

    def _spatial_block(self, i, x, mod_i, txt, B, Fr, S, timestep_int, ats):
        """BasicTransformerBlock.forward (latte_transformer_3d.py:357-517)."""
        w, C, H = self.w, self.C, self.H
        p = self.block_prefix(i)
        st = self.states[i]
        N = B * Fr * S
        use_pab = pab.enable_pab()
        bc = False
        if use_pab:
            bc, st.attn_count = pab.if_broadcast_spatial(timestep_int, st.attn_count)
        if bc:
            ops.add_rows(x, st.last_attn)
        else:
            self._self_attn_out(i, x, mod_i, B, Fr, S, st, use_pab, temporal=False)
        # cross attention: no norm, no modulation, no gate (:440-469); every frame of a sample sees the same text
        bc = False
        if use_pab:
            bc, st.cross_count = pab.if_broadcast_cross(timestep_int, st.cross_count)
        if bc:
            ops.add_rows(x, st.last_cross)
        else:
            q = self._gemm(x, p + ".attn2.to_q", out=self._buf("xm", (N, C)))
            ao = self._buf("attn_out", (N, C))
            d = i // 2
            for b in range(B):  # per-sample text length (cond / uncond prompts differ)
                rows = slice(b * Fr * S, (b + 1) * Fr * S)
                ops.flash_attn(q[rows], None, txt["kp"][d, b:b + 1], txt["vt"][d, b:b + 1], ao[rows], 1, H, Fr * S, txt["lens"][b])
            aux = None
            if use_pab:
                if st.last_cross is None or st.last_cross.shape != x.shape:
                    st.last_cross = torch.empty_like(x)
                aux = st.last_cross
            ops.gemm(ao, w[p + ".attn2.to_out.0.weight"], w[p + ".attn2.to_out.0.bias"], epilogue=ops.EPI_GATE_RES, res=x,
                     aux=aux, out=x)
        self._ff(i, x, mod_i, B, Fr, S, st, timestep_int, ats, temporal=False)
        return x

    def _temporal_block(self, i, x, mod_i, B, Fr, S, timestep_int, ats, Fa=None):
        """BasicTransformerBlock_.forward (latte_transformer_3d.py:680-824) on the (b, f, s)-ordered rows."""
        st = self.states[i]
        use_pab = pab.enable_pab()
        bc = False
        if use_pab:
            bc, st.attn_count = pab.if_broadcast_temporal(timestep_int, st.attn_count)
        if bc:
            ops.add_rows(x, st.last_attn)
        else:
            self._self_attn_out(i, x, mod_i, B, Fr, S, st, use_pab, temporal=True, Fa=Fa)
        self._ff(i, x, mod_i, B, Fr, S, st, timestep_int, ats, temporal=True)
        return x

    def _mlp_slab(self, like):
        """A slab for a PAB MLP-broadcast window: taken from the pool of slabs that closed windows handed back (a window's stored
        output lives until its last timestep, pab_mgr.py:148-174), so a generate() allocates at most as many 90 MB slabs as
        windows are open at once instead of one per window opening."""
        pool = self._ws.setdefault("mlp_slab_pool", [])
        for k, b in enumerate(pool):
            if b.shape == like.shape:
                return pool.pop(k)
        return torch.empty_like(like)

    def reset_pab_state(self):
        for st in self.states:
            st.attn_count = st.cross_count = st.mlp_count = 0
        if pab.PAB_MANAGER is not None:
            pab.PAB_MANAGER.config.mlp_spatial_outputs.clear()
            pab.PAB_MANAGER.config.mlp_temporal_outputs.clear()


def synth_state_dict(num_layers=28, num_heads=16, head_dim=72, caption_channels=4096, in_channels=4, out_channels=8,
                     patch_size=2, seed: int = 4321) -> Dict[str, torch.Tensor]:
    """Seeded random weights with the maxin-cn/Latte-1 transformer key names (no pretrained weights offline)."""
    g = torch.Generator().manual_seed(seed)
    C = num_heads * head_dim
    sd: Dict[str, torch.Tensor] = {}

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
