"""TEST INFRASTRUCTURE ONLY — CPU restatement (oracle) of the reference hot path.

Plain fp32 torch-on-CPU restatement of the Open-Sora STDiT3 denoise step, the
RFLOW sampler, the PAB schedule and the DSP layout switch, each function citing
the reference file:line it follows (paths relative to /root/reference).  It is
the checker for the HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  Nothing under
``videosys_amd/`` imports it and the product never falls back to it.

Pinning status: the reference's own tests hold NO golden vectors / KATs for
this path, so the restatement is pinned against outputs of
the reference itself run in the build container: ``oracle/make_golden.py``
imports the real reference (``oracle/ref_loader.py``) and writes
``tests/golden/*.pt``; ``tests/test_oracle_vs_golden.py`` checks this file
against those fixtures (and live against the reference when /root/reference
exists).  Third-party arithmetic restated from its published algorithm:
rotary-embedding-torch (unpinned in requirements.txt) and timm ``Mlp``.

Everything operates on a flat ``state_dict`` with the HF checkpoint key names
(``spatial_blocks.N.attn.qkv.weight`` ...), so the same weights drive the
reference, this oracle and the HIP path.

Device / dtype: the oracle is plain PyTorch, so the full-depth parity tests run it
AS THE CHECKER on the GPU (``STDiT3Oracle(..., device="cuda", dtype=torch.float32)``) where
a config-2 step takes seconds instead of an hour.  ``dtype=torch.bfloat16`` is the
"reference's own bf16 run": the same graph executed with torch's bf16 kernels the way
the reference runs it (``model.to(bf16)``: nn.LayerNorm / F.gelu / F.linear / SDPA, and
``native_attention`` with an fp32 softmax for sequences < 30, attentions.py:58,95-100,111-120)
— its distance from the fp32 run is the bf16 noise floor the HIP path is held against.
The fp32 path is unchanged op for op (the goldens pin it bit-tight).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# bench.py's cpu_baseline leg sets this: the fp32 SDPA call sites then go through F.scaled_dot_product_attention — the call the
# reference itself makes on this path (attentions.py:100,269) — instead of the explicit softmax(QK^T)V the goldens are pinned
# with (same arithmetic, fused CPU kernel: what the reference's CPU run would really cost)
FUSED_SDPA = False


# --------------------------------------------------------------------------
# elementary ops
# --------------------------------------------------------------------------
def layer_norm(x: Tensor, eps: float = 1e-6) -> Tensor:
    """nn.LayerNorm(C, eps=1e-6, elementwise_affine=False) — open_sora_transformer_3d.py:117,129,58."""
    if x.dtype != torch.float32:  # low-precision run: the reference's module is nn.LayerNorm (fp32 statistics inside)
        return F.layer_norm(x, (x.shape[-1],), None, None, eps)
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps)


def t2i_modulate(x: Tensor, shift: Tensor, scale: Tensor) -> Tensor:
    """open_sora_transformer_3d.py:47-48."""
    return x * (1 + scale) + shift


def rms_norm(x: Tensor, weight: Tensor, eps: float = 1e-6) -> Tensor:
    """LlamaRMSNorm.forward — modules/normalization.py:28-33 (fp32 inside, weight applied after)."""
    xf = x.to(torch.float32)
    var = xf.pow(2).mean(-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return weight * xf.to(x.dtype)


def gelu_tanh(x: Tensor) -> Tensor:
    """nn.GELU(approximate='tanh') — modules/activations.py:3."""
    if x.dtype != torch.float32:
        return F.gelu(x, approximate="tanh")
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x**3)))


def linear(x: Tensor, sd: Dict[str, Tensor], prefix: str) -> Tensor:
    w = sd[prefix + ".weight"]
    b = sd.get(prefix + ".bias")
    if x.dtype != torch.float32:  # nn.Linear: bias added to the fp32 accumulator, one rounding
        return F.linear(x, w, b)
    y = x @ w.t()
    return y + b if b is not None else y


def rope_table(freqs: Tensor, seq_len: int, pos_dtype=torch.float32):
    """cos/sin table of rotary_embedding_torch.RotaryEmbedding.forward (third-party, restated):
    angle[p, 2i] = angle[p, 2i+1] = p * freqs[i]."""
    seq = torch.arange(seq_len, dtype=pos_dtype, device=freqs.device)
    ang = torch.einsum("p,f->pf", seq.type(freqs.dtype), freqs)
    ang = ang.repeat_interleave(2, dim=-1)
    return ang.cos(), ang.sin()


def rope_rotate(t: Tensor, freqs: Tensor) -> Tensor:
    """RotaryEmbedding.rotate_queries_or_keys(t, seq_dim=-2) (called attentions.py:76-78):
    out = t*cos + rotate_half(t)*sin with rotate_half((a,b)) = (-b,a) on interleaved pairs."""
    cos, sin = rope_table(freqs, t.shape[-2], pos_dtype=t.dtype)
    x = t.reshape(*t.shape[:-1], -1, 2)
    x1, x2 = x.unbind(-1)
    rot = torch.stack((-x2, x1), dim=-1).reshape(t.shape)
    return ((t * cos) + (rot * sin)).type(t.dtype)


def sdpa(q: Tensor, k: Tensor, v: Tensor, key_len: Optional[Sequence[int]] = None, native: bool = False) -> Tensor:
    """softmax(q k^T / sqrt(d)) v in fp32 — what F.scaled_dot_product_attention (attentions.py:100,269)
    and native_attention (attentions.py:111-120) both compute; key_len[i] masks keys >= len for batch i
    (torch_impl mask, attentions.py:264-266).  In a low-precision run the two reference paths are kept apart:
    ``native`` = native_attention (q pre-scaled, logits in the activation dtype, fp32 softmax, cast back), else SDPA."""
    d = q.shape[-1]
    if q.dtype != torch.float32 or (FUSED_SDPA and not native):
        if native:
            attn = ((q * d**-0.5) @ k.transpose(-2, -1)).to(torch.float32).softmax(dim=-1).to(q.dtype)
            return attn @ v
        attn_mask = None
        if key_len is not None and any(m < k.shape[-2] for m in key_len):
            attn_mask = torch.zeros(q.shape[0], 1, q.shape[-2], k.shape[-2], dtype=torch.bool, device=q.device)
            for i, m in enumerate(key_len):
                attn_mask[i, :, :, :m] = True
        return F.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask)
    # a score tensor past ~8 GB (720p x 128f: 76 frames x 16 heads x 3600^2 fp32 = 63 GB) is computed in slices of the leading
    # axis — every op below is per (batch, head, row), so the slices give the same values
    per = q.shape[1] * q.shape[-2] * k.shape[-2] * 4 if q.dim() == 4 else 0
    if q.dim() == 4 and q.shape[0] > 1 and per * q.shape[0] > (8 << 30):
        step = max(1, (8 << 30) // per)
        return torch.cat([sdpa(q[i:i + step], k[i:i + step], v[i:i + step],
                               None if key_len is None else list(key_len[i:i + step]), native) for i in range(0, q.shape[0], step)], 0)
    s = (q @ k.transpose(-2, -1)) * (d**-0.5)
    if key_len is not None:
        L = k.shape[-2]
        for i, m in enumerate(key_len):
            if m < L:
                s[i, ..., m:] = float("-inf")
    return s.softmax(dim=-1) @ v


# --------------------------------------------------------------------------
# attention modules
# --------------------------------------------------------------------------
def self_attention(x: Tensor, sd, prefix: str, num_heads: int, rope_freqs: Optional[Tensor]) -> Tensor:
    """OpenSoraAttention.forward — modules/attentions.py:55-109.  x: [B', N', C]."""
    Bp, Np, C = x.shape
    D = C // num_heads
    qkv = linear(x, sd, prefix + ".qkv").view(Bp, Np, 3, num_heads, D).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)  # [B', H, N', D]
    if Np == 1:
        o = v  # attentions.py:65-66
    else:
        q = rms_norm(q, sd[prefix + ".q_norm.weight"])
        k = rms_norm(k, sd[prefix + ".k_norm.weight"])
        if rope_freqs is not None:  # attentions.py:76-78 (after the norm)
            q = rope_rotate(q, rope_freqs)
            k = rope_rotate(k, rope_freqs)
        o = sdpa(q, k, v, native=Np < 30)  # attentions.py:58 "use_flash_attn = N >= 30"
    o = o.transpose(1, 2).reshape(Bp, Np, C)
    return linear(o, sd, prefix + ".proj")


def cross_attention(x: Tensor, y: Tensor, y_lens: List[int], sd, prefix: str, num_heads: int) -> Tensor:
    """OpenSoraMultiHeadCrossAttention.forward/torch_impl — attentions.py:152-185,259-270.
    x: [B, N, C]; y: packed text [1, sum(y_lens), C] (equal lengths, as the reference requires)."""
    B, N, C = x.shape
    D = C // num_heads
    q = linear(x, sd, prefix + ".q_linear").view(1, -1, num_heads, D)
    kv = linear(y, sd, prefix + ".kv_linear").view(1, -1, 2, num_heads, D)
    k, v = kv.unbind(2)
    q = q.view(B, -1, num_heads, D).transpose(1, 2)
    k = k.view(B, -1, num_heads, D).transpose(1, 2)
    v = v.view(B, -1, num_heads, D).transpose(1, 2)
    o = sdpa(q, k, v, key_len=y_lens)
    o = o.transpose(1, 2).contiguous().view(B, N, C)
    return linear(o, sd, prefix + ".proj")


def mlp(x: Tensor, sd, prefix: str) -> Tensor:
    """timm Mlp(fc1 -> GELU(tanh) -> fc2) — used open_sora_transformer_3d.py:130-132 (third-party, restated)."""
    return linear(gelu_tanh(linear(x, sd, prefix + ".fc1")), sd, prefix + ".fc2")


# --------------------------------------------------------------------------
# PAB schedule — core/pab/pab_mgr.py
# --------------------------------------------------------------------------
class PABSchedule:
    """Restates PABManager.if_broadcast_{spatial,temporal,cross} (pab_mgr.py:54-91):
    flag = enabled and count % range != 0 and lo < t < hi ; count = (count+1) % steps."""

    def __init__(self, steps, spatial=None, temporal=None, cross=None, mlp_spatial=None, mlp_temporal=None):
        # each of spatial/temporal/cross: None (disabled) or (threshold_lo, threshold_hi, range)
        # mlp_spatial / mlp_temporal: None (MLP broadcast off) or {timestep: {"block": [...], "skip_count": n}}
        self.steps = steps
        self.cfg = {"spatial": spatial, "temporal": temporal, "cross": cross}
        self.mlp = {False: mlp_spatial, True: mlp_temporal}
        self.mlp_store = {False: {}, True: {}}

    def enabled(self):
        return any(v is not None for v in self.cfg.values())

    def mlp_enabled(self):
        return self.mlp[False] is not None or self.mlp[True] is not None

    def decide(self, kind: str, timestep: int, count: int):
        c = self.cfg[kind]
        flag = bool(c is not None and timestep is not None and (count % c[2] != 0) and (c[0] < timestep < c[1]))
        return flag, (count + 1) % self.steps

    def decide_mlp(self, timestep: int, block_idx: int, all_timesteps, temporal: bool):
        """PABManager.if_skip_mlp + _is_t_in_skip_config (pab_mgr.py:93-141), counter dropped (nothing reads it):
        returns (reuse stored output?, store this output?, key timestep of the window)."""
        rule = self.mlp[temporal] or {}
        window = None
        for first in rule:                      # first configured window (dict order) that contains the timestep
            if first not in all_timesteps:
                continue
            at = all_timesteps.index(first)
            n = int(rule[first]["skip_count"])
            if timestep in all_timesteps[at:at + 1 + n]:
                window = (all_timesteps[at], all_timesteps[at + n])
                break
        if timestep in rule and block_idx in rule[timestep]["block"]:
            return False, True, window
        if window is not None and block_idx in rule[window[0]]["block"]:
            return True, False, window
        return False, False, window


class _BlockState:
    def __init__(self):
        self.attn_count = 0
        self.cross_count = 0
        self.last_attn = None
        self.last_cross = None


# --------------------------------------------------------------------------
# DSP layout switch (pure tensor emulation over P in-process shards)
# --------------------------------------------------------------------------
def dsp_pad(n: int, P: int) -> int:
    """set_pad — comm.py:271-275."""
    return (P - n % P) % P


def dsp_split_sequence(x: Tensor, P: int, dim: int) -> List[Tensor]:
    """_split_sequence_func for every rank — comm.py:148-167 (zero pad then equal split)."""
    pad = dsp_pad(x.shape[dim], P)
    if pad:
        shp = list(x.shape)
        shp[dim] = pad
        x = torch.cat([x, torch.zeros(shp, dtype=x.dtype)], dim=dim)
    return [c.contiguous() for c in torch.split(x, x.shape[dim] // P, dim=dim)]


def dsp_gather_sequence(shards: List[Tensor], dim: int, pad: int) -> Tensor:
    """_gather_sequence_func — comm.py:170-190."""
    out = torch.cat(shards, dim=dim)
    return out.narrow(dim, 0, out.size(dim) - pad) if pad else out


def dsp_all_to_all(shards: List[Tensor], scatter_dim: int, gather_dim: int, scatter_pad: int, gather_pad: int):
    """all_to_all_with_pad on every rank at once — comm.py:104-108,282-304.
    shards[r] is rank r's [b, t, s, d] tensor; returns the list of per-rank outputs."""
    P = len(shards)
    ins = []
    for x in shards:
        if scatter_pad:
            shp = list(x.shape)
            shp[scatter_dim] = scatter_pad
            x = torch.cat([x, torch.zeros(shp, dtype=x.dtype)], dim=scatter_dim)
        assert x.shape[scatter_dim] % P == 0
        ins.append([t.contiguous() for t in torch.tensor_split(x, P, scatter_dim)])
    outs = []
    for r in range(P):
        o = torch.cat([ins[src][r] for src in range(P)], dim=gather_dim)
        if gather_pad:
            o = o.narrow(gather_dim, 0, o.size(gather_dim) - gather_pad)
        outs.append(o.contiguous())
    return outs


# --------------------------------------------------------------------------
# STDiT3
# --------------------------------------------------------------------------
def timestep_embedding(t: Tensor, dim: int = 256, max_period: int = 10000) -> Tensor:
    """TimestepEmbedder.timestep_embedding — modules/embeddings.py:123-141."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def embed_mlp(t_freq: Tensor, sd, prefix: str) -> Tensor:
    """Linear -> SiLU -> Linear (TimestepEmbedder/SizeEmbedder.mlp) — embeddings.py:114-118,143-146."""
    t_freq = t_freq.to(sd[prefix + ".mlp.0.weight"].dtype)  # "if t_freq.dtype != dtype: t_freq = t_freq.to(dtype)"
    return linear(F.silu(linear(t_freq, sd, prefix + ".mlp.0")), sd, prefix + ".mlp.2")


def pos_embed_2d(dim: int, h: int, w: int, scale: float, base_size: int) -> Tensor:
    """OpenSoraPositionEmbedding2D._get_cached_emb — embeddings.py:231-272."""
    half = dim // 2
    inv_freq = 1.0 / (10000 ** (torch.arange(0, half, 2).float() / half))
    grid_h = torch.arange(h) / scale
    grid_w = torch.arange(w) / scale
    grid_h = grid_h * (base_size / h)
    grid_w = grid_w * (base_size / w)
    gh, gw = torch.meshgrid(grid_w, grid_h, indexing="ij")  # "here w goes first"
    gh = gh.t().reshape(-1)
    gw = gw.t().reshape(-1)

    def sc(t):
        out = torch.einsum("i,d->id", t, inv_freq)
        return torch.cat((torch.sin(out), torch.cos(out)), dim=-1)

    return torch.cat([sc(gh), sc(gw)], dim=-1).unsqueeze(0)


def patch_embed(x: Tensor, sd, patch=(1, 2, 2)) -> Tensor:
    """OpenSoraPatchEmbed3D.forward — embeddings.py:85-104 (pad, Conv3d k=s=patch, flatten)."""
    _, _, D, H, W = x.shape
    if W % patch[2]:
        x = F.pad(x, (0, patch[2] - W % patch[2]))
    if H % patch[1]:
        x = F.pad(x, (0, 0, 0, patch[1] - H % patch[1]))
    if D % patch[0]:
        x = F.pad(x, (0, 0, 0, 0, 0, patch[0] - D % patch[0]))
    x = F.conv3d(x, sd["x_embedder.proj.weight"], sd["x_embedder.proj.bias"], stride=patch)
    return x.flatten(2).transpose(1, 2)


def encode_text(y: Tensor, mask: Optional[Tensor], sd):
    """STDiT3.encode_text — open_sora_transformer_3d.py:526-537 (+ OpenSoraCaptionEmbedder y_proj Mlp)."""
    y = mlp(y, sd, "y_embedder.y_proj")  # [B,1,L,C]
    C = y.shape[-1]
    if mask is not None:
        if mask.shape[0] != y.shape[0]:
            mask = mask.repeat(y.shape[0] // mask.shape[0], 1)
        mask = mask.squeeze(1).squeeze(1)
        y = y.squeeze(1).masked_select(mask.unsqueeze(-1) != 0).view(1, -1, C)
        y_lens = mask.sum(dim=1).tolist()
    else:
        y_lens = [y.shape[2]] * y.shape[0]
        y = y.squeeze(1).view(1, -1, C)
    return y, y_lens


def t_mask_select(x_mask: Tensor, x: Tensor, masked_x: Tensor, T: int, S: int) -> Tensor:
    """t_mask_select — open_sora_transformer_3d.py:65-73,152-160: rows of frames with x_mask True keep x, the others take masked_x."""
    B, _, C = x.shape
    m = x_mask.to(torch.bool)[:, :, None, None]
    return torch.where(m, x.reshape(B, T, S, C), masked_x.reshape(B, T, S, C)).reshape(B, T * S, C)


def stdit3_block(
    x, y, t_mlp, y_lens, T, S, sd, prefix, num_heads, temporal, rope_freqs,
    pab: Optional[PABSchedule] = None, state: Optional[_BlockState] = None, timestep_int: Optional[int] = None,
    sp_shards: int = 1, block_idx: int = 0, all_timesteps=None, x_mask: Optional[Tensor] = None,
    t0_mlp: Optional[Tensor] = None,
):
    """STDiT3Block.forward — open_sora_transformer_3d.py:162-286, incl. the MLP broadcast (:232-280) when the schedule carries
    MLP rules and ``all_timesteps`` is handed down (which the reference's STDiT3.forward forgets to do), and the frame-wise
    choice between the modulation of ``t`` and of timestep 0 when ``x_mask`` [B, T] is given (:181-184,198-200,220-222,
    262-264,271-273: frames whose mask is False are conditioning frames and see the t = 0 shift / scale / gate)."""
    B, N, C = x.shape
    mods = (sd[prefix + ".scale_shift_table"][None] + t_mlp.reshape(B, 6, -1)).chunk(6, dim=1)
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = mods
    if x_mask is not None:
        mods0 = (sd[prefix + ".scale_shift_table"][None] + t0_mlp.reshape(B, 6, -1)).chunk(6, dim=1)
        shift_msa0, scale_msa0, gate_msa0, shift_mlp0, scale_mlp0, gate_mlp0 = mods0

    broadcast_attn = False
    if pab is not None and pab.enabled():
        broadcast_attn, state.attn_count = pab.decide("temporal" if temporal else "spatial", timestep_int, state.attn_count)
    if broadcast_attn:
        x_m_s = state.last_attn
    else:
        normed = layer_norm(x)
        x_m = t2i_modulate(normed, shift_msa, scale_msa)
        if x_mask is not None:
            x_m = t_mask_select(x_mask, x_m, t2i_modulate(normed, shift_msa0, scale_msa0), T, S)
        if temporal:
            x_m = x_m.reshape(B, T, S, C).permute(0, 2, 1, 3).reshape(B * S, T, C)
            x_m = self_attention(x_m, sd, prefix + ".attn", num_heads, rope_freqs)
            x_m = x_m.reshape(B, S, T, C).permute(0, 2, 1, 3).reshape(B, T * S, C)
        else:
            x_m = x_m.reshape(B * T, S, C)
            x_m = self_attention(x_m, sd, prefix + ".attn", num_heads, None)
            x_m = x_m.reshape(B, T * S, C)
        x_m_s = gate_msa * x_m
        if x_mask is not None:
            x_m_s = t_mask_select(x_mask, x_m_s, gate_msa0 * x_m, T, S)
        if pab is not None and pab.enabled():
            state.last_attn = x_m_s
    x = x + x_m_s

    broadcast_cross = False
    if pab is not None and pab.enabled():
        broadcast_cross, state.cross_count = pab.decide("cross", timestep_int, state.cross_count)
    if broadcast_cross:
        x = x + state.last_cross
    else:
        x_cross = cross_attention(x, y, y_lens, sd, prefix + ".cross_attn", num_heads)
        if pab is not None and pab.enabled():
            state.last_cross = x_cross
        x = x + x_cross

    reuse = store = False
    window = None
    if pab is not None and pab.enabled() and pab.mlp_enabled():
        reuse, store, window = pab.decide_mlp(timestep_int, block_idx, all_timesteps, temporal)
    if reuse:   # get_mlp_output (pab_mgr.py:148-174): the entry is dropped at the window's last timestep
        bank = pab.mlp_store[temporal]
        x_m_s = bank[(window[0], block_idx)]
        if timestep_int == window[1]:
            del bank[(window[0], block_idx)]
    else:
        normed = layer_norm(x)
        x_m = t2i_modulate(normed, shift_mlp, scale_mlp)
        if x_mask is not None:
            x_m = t_mask_select(x_mask, x_m, t2i_modulate(normed, shift_mlp0, scale_mlp0), T, S)
        h = mlp(x_m, sd, prefix + ".mlp")
        x_m_s = gate_mlp * h
        if x_mask is not None:
            x_m_s = t_mask_select(x_mask, x_m_s, gate_mlp0 * h, T, S)
        if store:
            pab.mlp_store[temporal][(timestep_int, block_idx)] = x_m_s
    return x + x_m_s


def final_layer(x, t, sd, x_mask=None, t0=None, T=None, S=None):
    """T2IFinalLayer.forward — open_sora_transformer_3d.py:75-87."""
    shift, scale = (sd["final_layer.scale_shift_table"][None] + t[:, None]).chunk(2, dim=1)
    x = t2i_modulate(layer_norm(x), shift, scale)
    if x_mask is not None:
        # as written in the reference (:83-85) the conditioning branch normalises the ALREADY modulated x a second time
        # (``x`` was reassigned two lines earlier): x_zero = mod0(LN(mod_t(LN(x)))).  Restated as is — the fixture pins it.
        shift0, scale0 = (sd["final_layer.scale_shift_table"][None] + t0[:, None]).chunk(2, dim=1)
        x = t_mask_select(x_mask, x, t2i_modulate(layer_norm(x), shift0, scale0), T, S)
    return linear(x, sd, "final_layer.linear")


def unpatchify(x, N_t, N_h, N_w, R_t, R_h, R_w, patch, c_out):
    """STDiT3.unpatchify — open_sora_transformer_3d.py:634-658."""
    B = x.shape[0]
    Tp, Hp, Wp = patch
    x = x.view(B, N_t, N_h, N_w, Tp, Hp, Wp, c_out)
    x = x.permute(0, 7, 1, 4, 2, 5, 3, 6).reshape(B, c_out, N_t * Tp, N_h * Hp, N_w * Wp)
    return x[:, :, :R_t, :R_h, :R_w]


class STDiT3Oracle:
    """STDiT3.forward — open_sora_transformer_3d.py:539-632, sp=cp=1, fp32."""

    def __init__(self, sd: Dict[str, Tensor], depth: int, hidden_size: int, num_heads: int,
                 patch_size=(1, 2, 2), in_channels: int = 4, input_sq_size: int = 512, pred_sigma: bool = True,
                 device=None, dtype=torch.float32):
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.dtype = dtype
        # rope.freqs stays fp32 in every run (the HIP model also keeps the checkpoint dtype for it)
        self.sd = {k: v.to(device=self.device, dtype=torch.float32 if k == "rope.freqs" else dtype) for k, v in sd.items()}
        self.depth, self.C, self.H = depth, hidden_size, num_heads
        self.patch = tuple(patch_size)
        self.in_channels = in_channels
        self.out_channels = in_channels * 2 if pred_sigma else in_channels
        self.input_sq_size = input_sq_size
        self.pab: Optional[PABSchedule] = None
        self.states = {}

    def set_pab(self, pab: Optional[PABSchedule]):
        self.pab = pab
        self.states = {}

    def _state(self, key):
        if key not in self.states:
            self.states[key] = _BlockState()
        return self.states[key]

    def embed(self, x, timestep, y, mask, fps, height, width):
        sd = self.sd
        _, _, Tx, Hx, Wx = x.shape
        p = self.patch
        T, H, W = -(-Tx // p[0]), -(-Hx // p[1]), -(-Wx // p[2])
        S = H * W
        base_size = round(S**0.5)
        scale = ((float(height[0]) * float(width[0])) ** 0.5) / self.input_sq_size
        dev, dt = self.device, self.dtype
        x, timestep, y, fps = x.to(dev), timestep.to(dev), y.to(dev), fps.to(dev)
        mask = None if mask is None else mask.to(dev)
        pos = pos_embed_2d(self.C, H, W, scale, base_size).to(device=dev, dtype=dt)
        B = x.shape[0]
        t = embed_mlp(timestep_embedding(timestep.float()), sd, "t_embedder")
        f = fps.unsqueeze(1) if fps.ndim == 1 else fps
        if f.shape[0] != B:
            f = f.repeat(B // f.shape[0], 1)
        fe = embed_mlp(timestep_embedding(f.reshape(-1).float()), sd, "fps_embedder").view(B, -1)
        t = t + fe
        t_mlp = linear(F.silu(t), sd, "t_block.1")
        # the timestep-0 embedding conditioning frames are modulated with (:578-582); dropped by callers without x_mask
        self._t0 = embed_mlp(timestep_embedding(torch.zeros_like(timestep).float()), sd, "t_embedder") + fe
        self._t0_mlp = linear(F.silu(self._t0), sd, "t_block.1")
        yy, y_lens = encode_text(y.to(dt), mask, sd)
        xe = patch_embed(x.to(dt), sd, p).view(B, T, S, self.C) + pos
        return xe.reshape(B, T * S, self.C), t, t_mlp, yy, y_lens, (T, H, W, Tx, Hx, Wx)

    def forward(self, x, timestep, y, mask=None, fps=None, height=None, width=None, valid_depth=None,
                return_hidden=False, all_timesteps=None, x_mask=None):
        x, t, t_mlp, yy, y_lens, (T, H, W, Tx, Hx, Wx) = self.embed(x, timestep, y, mask, fps, height, width)
        S = H * W
        xm_kw = {}
        t0 = None
        if x_mask is not None:
            x_mask = x_mask.to(self.device).to(torch.bool)
            t0 = self._t0
            xm_kw = dict(x_mask=x_mask, t0_mlp=self._t0_mlp)
        rope_freqs = self.sd["rope.freqs"]
        ts_int = int(timestep[0]) if self.pab is not None else None
        depth = self.depth if valid_depth is None else valid_depth
        hidden = []
        for d in range(depth):
            x = stdit3_block(x, yy, t_mlp, y_lens, T, S, self.sd, f"spatial_blocks.{d}", self.H, False, None,
                             self.pab, self._state(("s", d)), ts_int, block_idx=d, all_timesteps=all_timesteps, **xm_kw)
            x = stdit3_block(x, yy, t_mlp, y_lens, T, S, self.sd, f"temporal_blocks.{d}", self.H, True, rope_freqs,
                             self.pab, self._state(("t", d)), ts_int, block_idx=d, all_timesteps=all_timesteps, **xm_kw)
            if callable(return_hidden):   # full-depth parity tests: per-block-pair error growth without keeping 28 copies
                return_hidden(d, x)
            elif return_hidden:
                hidden.append(x.clone())
        out = final_layer(x, t, self.sd, x_mask, t0, T, S)
        out = unpatchify(out, T, H, W, Tx, Hx, Wx, self.patch, self.out_channels).to(torch.float32)
        return (out, hidden) if (return_hidden and not callable(return_hidden)) else out

    __call__ = forward


# --------------------------------------------------------------------------
# RFLOW sampler — schedulers/scheduling_rflow_open_sora.py
# --------------------------------------------------------------------------
def timestep_transform(t, height, width, num_frames, base_resolution=512 * 512, base_num_frames=1, scale=1.0,
                       num_timesteps=1):
    """scheduling_rflow_open_sora.py:47-70."""
    t = t / num_timesteps
    resolution = height * width
    ratio_space = (resolution / base_resolution).sqrt()
    if num_frames[0] == 1:
        nf = torch.ones_like(num_frames)
    else:
        nf = num_frames // 17 * 5
    ratio_time = (nf / base_num_frames).sqrt()
    ratio = ratio_space * ratio_time * scale
    new_t = ratio * t / (1 + (ratio - 1) * t)
    return new_t * num_timesteps


def rflow_timesteps(num_sampling_steps, batch, height, width, num_frames, num_timesteps=1000,
                    use_timestep_transform=True):
    """scheduling_rflow_open_sora.py:208-213."""
    ts = [(1.0 - i / num_sampling_steps) * num_timesteps for i in range(num_sampling_steps)]
    ts = [torch.tensor([t] * batch) for t in ts]
    if use_timestep_transform:
        ts = [timestep_transform(t, height, width, num_frames, num_timesteps=num_timesteps) for t in ts]
    return ts


def rflow_sample(model, z, y, y_null, mask, fps, height, width, num_frames, num_sampling_steps=30,
                 cfg_scale=7.0, num_timesteps=1000, use_timestep_transform=True, model_dtype=torch.float32,
                 return_all=False, cond_mask=None, noise_fn=None):
    """RFLOW.sample — scheduling_rflow_open_sora.py:188-257.  ``cond_mask`` [B, T] float is the reference's ``mask`` argument
    (apply_mask_strategy, pipeline_open_sora.py:825-854: 1 = generate, an edit ratio in [0, 1) = conditioning frame): a frame is
    denoised only while mask * num_timesteps >= t (:227-236), is noised ONCE, at the step it joins (:229-236), sees the t = 0
    modulation before that (x_mask, :232) and is put back after every update while it is still conditioning (:254-255).
    ``noise_fn(shape)`` draws the per-step noise (default torch.randn: the reference's randn_like on the global generator)."""
    yy = torch.cat([y, y_null], 0)
    noise_fn = noise_fn or (lambda shape: torch.randn(shape))
    noise_added = None if cond_mask is None else (cond_mask == 1)
    timesteps = rflow_timesteps(num_sampling_steps, z.shape[0], height, width, num_frames, num_timesteps,
                                use_timestep_transform)
    all_timesteps = [int(t.to(model_dtype).item()) for t in timesteps]
    zs = []
    for i, t in enumerate(timesteps):
        extra = {"all_timesteps": all_timesteps} if getattr(getattr(model, "pab", None), "mlp_enabled", lambda: False)() else {}
        if cond_mask is not None:
            x0 = z.clone()
            tp = 1 - t.float() / num_timesteps                       # RFlowScheduler.add_noise (:144-161)
            x_noise = tp[:, None, None, None, None] * x0 + (1 - tp)[:, None, None, None, None] * noise_fn(x0.shape)
            upper = (cond_mask * num_timesteps) >= t.unsqueeze(1)
            extra["x_mask"] = upper.repeat(2, 1)
            z = torch.where((upper & ~noise_added)[:, None, :, None, None], x_noise, x0)
            noise_added = upper
        z_in = torch.cat([z, z], 0)
        tt = torch.cat([t, t], 0).to(model_dtype)  # STDiT3.forward casts timestep to model dtype (:562)
        out = model(z_in, tt, yy, mask=mask, fps=torch.cat([fps, fps]), height=torch.cat([height, height]),
                    width=torch.cat([width, width]), **extra)
        pred = out.to(z.device).chunk(2, dim=1)[0]  # the model may run on another device (GPU-as-checker)
        pred_cond, pred_uncond = pred.chunk(2, dim=0)
        v_pred = pred_uncond + cfg_scale * (pred_cond - pred_uncond)
        dt = timesteps[i] - timesteps[i + 1] if i < len(timesteps) - 1 else timesteps[i]
        dt = dt / num_timesteps
        z = z + v_pred * dt[:, None, None, None, None]
        if cond_mask is not None:
            z = torch.where(upper[:, None, :, None, None], z, x0)
        if return_all:
            zs.append(z.clone())
    return (z, zs, all_timesteps) if return_all else z


# --------------------------------------------------------------------------
# synthetic weights: seeded, zero-initialised tensors re-drawn
# --------------------------------------------------------------------------
def synth_state_dict(depth, hidden_size, num_heads, caption_channels=4096, model_max_length=300,
                     in_channels=4, patch_size=(1, 2, 2), pred_sigma=True, mlp_ratio=4.0, seed=1234,
                     freq_dim=256) -> Dict[str, Tensor]:
    """Random-init weights with the HF-checkpoint key names/shapes of STDiT3 (open_sora_transformer_3d.py:364-446).
    Distribution is ours (no pretrained weights offline): N(0, 0.02)-scaled linears (std 1/sqrt(fan_in) capped),
    q/k norm weights 1+N(0,0.1), nothing left at zero so every path is exercised."""
    g = torch.Generator().manual_seed(seed)
    C, D = hidden_size, hidden_size // num_heads
    Hm = int(hidden_size * mlp_ratio)
    out_ch = in_channels * 2 if pred_sigma else in_channels
    sd: Dict[str, Tensor] = {}

    def lin(name, n_out, n_in, bias=True, std=None):
        s = std if std is not None else min(0.02 * 4, 1.0 / math.sqrt(n_in))
        sd[name + ".weight"] = torch.randn(n_out, n_in, generator=g) * s
        if bias:
            sd[name + ".bias"] = torch.randn(n_out, generator=g) * 0.02

    sd["x_embedder.proj.weight"] = torch.randn(C, in_channels, *patch_size, generator=g) * 0.1
    sd["x_embedder.proj.bias"] = torch.randn(C, generator=g) * 0.02
    for e in ("t_embedder", "fps_embedder"):
        lin(e + ".mlp.0", C, freq_dim)
        lin(e + ".mlp.2", C, C)
    lin("t_block.1", 6 * C, C)
    lin("y_embedder.y_proj.fc1", C, caption_channels)
    lin("y_embedder.y_proj.fc2", C, C)
    sd["y_embedder.y_embedding"] = torch.randn(model_max_length, caption_channels, generator=g) / caption_channels**0.5
    sd["rope.freqs"] = 1.0 / (10000 ** (torch.arange(0, D, 2)[: (D // 2)].float() / D))
    for kind in ("spatial_blocks", "temporal_blocks"):
        for i in range(depth):
            p = f"{kind}.{i}"
            sd[p + ".scale_shift_table"] = torch.randn(6, C, generator=g) / C**0.5
            lin(p + ".attn.qkv", 3 * C, C)
            sd[p + ".attn.q_norm.weight"] = 1 + 0.1 * torch.randn(D, generator=g)
            sd[p + ".attn.k_norm.weight"] = 1 + 0.1 * torch.randn(D, generator=g)
            lin(p + ".attn.proj", C, C)
            lin(p + ".cross_attn.q_linear", C, C)
            lin(p + ".cross_attn.kv_linear", 2 * C, C)
            lin(p + ".cross_attn.proj", C, C)
            lin(p + ".mlp.fc1", Hm, C)
            lin(p + ".mlp.fc2", C, Hm)
    sd["final_layer.scale_shift_table"] = torch.randn(2, C, generator=g) / C**0.5
    lin("final_layer.linear", int(math.prod(patch_size)) * out_ch, C)
    return sd
