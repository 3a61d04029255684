"""STDiT3 (Open-Sora v1.2) denoise step on MI355X — host mirror of
videosys/models/transformers/open_sora_transformer_3d.py (reference :318-667).

Same constructor config, same ``forward(x, timestep, y, all_timesteps=None, mask=None, x_mask=None, fps=None,
height=None, width=None, **kwargs)`` signature and return (``[B, 2*C_in, T, H, W]`` fp32), same state-dict key names
as the HF checkpoint ``hpcai-tech/OpenSora-STDiT-v3`` — but every tensor op of the per-step path is a call into
libvideosys_amd.so (hand-written gfx950 kernels behind the C ABI of include/videosys_amd.h).  PyTorch only owns the
HBM buffers, the stream and torch.distributed.  There is no eager fallback: on a machine without the library or
without a HIP device, constructing the model raises.

Differences from the reference that do not change results (SURVEY.md §7 "hard parts"):
  * step-invariant work is hoisted: y_embedder(y), the 2*depth kv_linear(y) projections and their attention layouts
    are computed once per prompt and cached (the reference recomputes them every step, attentions.py:157);
  * the 2*depth x 6 modulation vectors of a step come from one kernel (open_sora_transformer_3d.py:177-179);
  * no torch.utils.checkpoint wrapper around blocks (core/dcp/recompute.py:141-153 is pure overhead under no_grad);
  * PAB decisions use the host-side integer timestep (no ``int(timestep[0])`` device sync per block);
  * ``all_timesteps`` IS forwarded to the blocks (the reference forgets to: SURVEY.md §0.9), so ``mlp_broadcast=True`` — the
    default of OpenSoraPABConfig — works here instead of raising TypeError;
  * x_mask (image / video conditioning, :181-184,198-200,220-222,262-273,578-582): the reference computes both modulations of
    every row and picks per frame with torch.where; here the modulation VECTOR is picked per (sample, frame) up front and the
    kernels run with one frame's rows per modulation row — the same selection without the second pass.  Such steps are issued
    eagerly (no launch program): the mask changes from step to step.
"""
from __future__ import annotations

import math
import os
from types import SimpleNamespace
from typing import Dict, Optional

import torch

from . import dsp, ops, pab, program
from .utils import same_tensor


class STDiT3Config:
    """Same kwargs/defaults as the reference STDiT3Config (open_sora_transformer_3d.py:318-361)."""

    model_type = "STDiT3"

    def __init__(self, input_size=(None, None, None), input_sq_size=512, in_channels=4, patch_size=(1, 2, 2),
                 hidden_size=1152, depth=28, num_heads=16, mlp_ratio=4.0, class_dropout_prob=0.1, pred_sigma=True,
                 drop_path=0.0, caption_channels=4096, model_max_length=300, qk_norm=True, enable_flash_attn=False,
                 only_train_temporal=False, freeze_y_embedder=False, skip_y_embedder=False, **kwargs):
        self.input_size = input_size
        self.input_sq_size = input_sq_size
        self.in_channels = in_channels
        self.patch_size = tuple(patch_size)
        self.hidden_size = hidden_size
        self.depth = depth
        self.num_heads = num_heads
        self.mlp_ratio = mlp_ratio
        self.class_dropout_prob = class_dropout_prob
        self.pred_sigma = pred_sigma
        self.drop_path = drop_path
        self.caption_channels = caption_channels
        self.model_max_length = model_max_length
        self.qk_norm = qk_norm
        self.enable_flash_attn = enable_flash_attn
        self.only_train_temporal = only_train_temporal
        self.freeze_y_embedder = freeze_y_embedder
        self.skip_y_embedder = skip_y_embedder
        for k, v in kwargs.items():
            setattr(self, k, v)


def pos_embed_2d(dim: int, h: int, w: int, scale: float, base_size: int) -> torch.Tensor:
    """Constant table of OpenSoraPositionEmbedding2D (modules/embeddings.py:231-272); built once per resolution on
    the host exactly as the reference does (it lru_caches the same table), then uploaded."""
    half = dim // 2
    inv_freq = 1.0 / (10000 ** (torch.arange(0, half, 2).float() / half))
    gh = torch.arange(h) / scale
    gw = torch.arange(w) / scale
    gh = gh * (base_size / h)
    gw = gw * (base_size / w)
    gh, gw = torch.meshgrid(gw, gh, indexing="ij")
    gh = gh.t().reshape(-1)
    gw = gw.t().reshape(-1)

    def sincos(t):
        out = torch.einsum("i,d->id", t, inv_freq)
        return torch.cat((torch.sin(out), torch.cos(out)), dim=-1)

    return torch.cat([sincos(gh), sincos(gw)], dim=-1)  # [h*w, dim]


def rope_tables(freqs: torch.Tensor, T: int, pos_dtype: torch.dtype):
    """cos/sin tables [T, head_dim] of rotary_embedding_torch.RotaryEmbedding (third-party; positions arange(T) in the
    activation dtype, angles in the dtype of ``freqs`` — so a bf16-cast model reproduces the reference's bf16 angles)."""
    seq = torch.arange(T, dtype=pos_dtype)
    ang = torch.einsum("p,f->pf", seq.type(freqs.dtype), freqs)
    ang = ang.repeat_interleave(2, dim=-1)
    return ang.cos().float().contiguous(), ang.sin().float().contiguous()


class _BlockState:
    """PAB bookkeeping of one STDiT3Block (open_sora_transformer_3d.py:141-147)."""

    def __init__(self, block_idx, temporal):
        self.block_idx = block_idx
        self.temporal = temporal
        self.attn_count = 0
        self.cross_count = 0
        # does last_attn / last_cross hold the output of this block's LAST computed call?  (slab elision writes a slab only when the
        # next schedule entry will broadcast it: a caller that leaves the schedule must not be served a stale one)
        self.attn_valid = False
        self.cross_valid = False
        self.mlp_count = 0
        self.last_attn: Optional[torch.Tensor] = None
        self.last_cross: Optional[torch.Tensor] = None


class STDiT3:
    """Drop-in for the reference STDiT3 at the operator boundary ``model(z_in, t, **model_args)``."""

    config_class = STDiT3Config

    def __init__(self, config: STDiT3Config, device="cuda", dtype=torch.bfloat16):
        if dtype != torch.bfloat16:
            raise ValueError("the MI355X path computes in bf16 (fp32 accumulate); got %s" % dtype)
        from . import _lib

        _lib.load()  # fail loudly if the HIP library is missing
        self.config = config
        self.device = torch.device(device)
        self.dtype = dtype
        self.pred_sigma = config.pred_sigma
        self.in_channels = config.in_channels
        self.out_channels = config.in_channels * 2 if config.pred_sigma else config.in_channels
        self.depth = config.depth
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_heads
        self.patch_size = tuple(config.patch_size)
        self.input_sq_size = config.input_sq_size
        if self.hidden_size // self.num_heads != ops.HEAD_DIM:
            raise ValueError("attention kernels are built for head_dim 72 (STDiT3-XL/2)")
        if self.patch_size[0] != 1:
            raise ValueError("temporal patch size must be 1")
        self.w: Dict[str, torch.Tensor] = {}
        self.rope_freqs: Optional[torch.Tensor] = None  # kept in the checkpoint dtype on the host
        self.parallel_manager = SimpleNamespace(sp_size=1, cp_size=1, dp_size=1, dp_rank=0, sp_group=None, cp_group=None)
        self._sp: Optional[dsp.SequenceParallel] = None
        self._overlap, self._switch, self._scatter, self._side = False, "auto", "flat", None
        self.states = [_BlockState(i // 2, bool(i % 2)) for i in range(2 * self.depth)]
        self._pos_cache = {}
        self._rope_cache = {}
        self._text_cache = None
        self._fps_cache = {}
        self._ws = {}
        self._hidden_tap = None   # test hook: callable(pair_index, x_rows) after every (spatial, temporal) block pair
        # launch programs (program.py): a step is recorded once per (geometry, PAB decision pattern, parallel layout) and replayed
        # through vsys_program_run afterwards; VSYS_PROGRAMS=0 issues every launch from Python every step
        self.pab_elide_unused = os.environ.get("VSYS_PAB_ELIDE", "1") != "0"   # see _pab_plan: slabs nobody will read are not written
        # PAB broadcasts (`x = x + cached_output`) that follow a gate+residual GEMM in program order ride in that GEMM's store phase
        # (ops.gemm_gate_res_add) instead of being 270 MB passes of their own; same roundings, same bits (_block, `upcoming`)
        self.pab_fold_adds = os.environ.get("VSYS_PAB_FOLD", "1") != "0"
        self._bc, self._folded = None, {}
        self._kbounds = {}
        self.use_programs = os.environ.get("VSYS_PROGRAMS", "1") != "0"
        # AdaLN folded into the qkv / fc1 GEMMs (csrc/adaln_fold.hip): pre-scaled weights per step, row statistics from the
        # producing epilogue; VSYS_ADALN_FOLD=0 keeps the separate LayerNorm-modulate pass everywhere
        self.adaln_fold = os.environ.get("VSYS_ADALN_FOLD", "1") != "0"
        self.fold_spatial_qkv = True   # test hook: False = the spatial qkv site keeps the separate AdaLN pass on one GPU too (what a
        #                                sequence-parallel rank whose modulated activations travel computes, bit for bit)
        self._fold = None          # per-B site table + the W' / cs / cv buffers (built on first use)
        self._stats_fresh = False  # "the statistics buffer describes the current x" (within one step)
        self._programs = {}
        self.program_stats = dict(recorded=0, replayed=0, eager=0)
        # attribute paths the reference's callers read (scheduling_rflow_open_sora.py:221, pipeline_open_sora.py:295)
        self.x_embedder = SimpleNamespace(proj=SimpleNamespace(weight=torch.empty(0, dtype=dtype)))
        self.y_embedder = SimpleNamespace(y_embedding=None)

    # ------------------------------------------------------------------ weights
    def block_prefix(self, i):
        return f"{'temporal' if i % 2 else 'spatial'}_blocks.{i // 2}"

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        need = self.expected_keys()
        self._kbounds = {}     # (promises derived from the norm weights)
        missing = [k for k in need if k not in sd]
        if strict and missing:
            raise KeyError(f"missing keys: {missing[:8]}{'...' if len(missing) > 8 else ''}")
        for k in need:
            if k not in sd:
                continue
            t = sd[k]
            if k == "rope.freqs":
                self.rope_freqs = t.detach().cpu().clone()
                continue
            if k == "x_embedder.proj.weight":
                t = t.reshape(t.shape[0], -1)
            self.w[k] = t.detach().to(device=self.device, dtype=self.dtype).contiguous()
        tabs = [self.w[self.block_prefix(i) + ".scale_shift_table"].reshape(-1) for i in range(2 * self.depth)]
        self.w["_all_tables"] = torch.stack(tabs).contiguous()  # [2*depth, 6*C] in execution order
        self.x_embedder.proj.weight = self.w["x_embedder.proj.weight"]
        self.y_embedder.y_embedding = self.w["y_embedder.y_embedding"]
        self._text_cache = None
        self._rope_cache = {}
        self._programs = {}
        self._fold = None
        return self

    def expected_keys(self):
        keys = ["x_embedder.proj.weight", "x_embedder.proj.bias", "t_block.1.weight", "t_block.1.bias",
                "y_embedder.y_proj.fc1.weight", "y_embedder.y_proj.fc1.bias", "y_embedder.y_proj.fc2.weight",
                "y_embedder.y_proj.fc2.bias", "y_embedder.y_embedding", "rope.freqs",
                "final_layer.scale_shift_table", "final_layer.linear.weight", "final_layer.linear.bias"]
        for e in ("t_embedder", "fps_embedder"):
            for l in ("0", "2"):
                keys += [f"{e}.mlp.{l}.weight", f"{e}.mlp.{l}.bias"]
        for i in range(2 * self.depth):
            p = self.block_prefix(i)
            keys += [p + ".scale_shift_table", p + ".attn.q_norm.weight", p + ".attn.k_norm.weight"]
            for l in ("attn.qkv", "attn.proj", "cross_attn.q_linear", "cross_attn.kv_linear", "cross_attn.proj", "mlp.fc1",
                      "mlp.fc2"):
                keys += [f"{p}.{l}.weight", f"{p}.{l}.bias"]
        return keys

    def state_dict(self):
        sd = {k: v for k, v in self.w.items() if not k.startswith("_")}
        sd["rope.freqs"] = self.rope_freqs
        return sd

    # ------------------------------------------------------------------ parallel
    def enable_parallel(self, dp_size=None, sp_size=None, enable_cp=None, parallel_mgr=None, copy_executor=None, overlap=None):
        """open_sora_transformer_3d.py:466-482, incl. enable_cp: with an even sp_size the CFG batch is split over cp = 2 rank
        groups (each runs DSP over sp_size / 2 ranks on ONE sample) and the outputs are gathered along the batch (:546-557,621)."""
        if parallel_mgr is not None:
            self.parallel_manager = parallel_mgr
        else:
            sp_size, cp_size = sp_size or 1, 1
            if enable_cp and sp_size % 2 == 0:   # "update cfg parallel" (:470-475): the CFG pair goes to two rank groups
                sp_size, cp_size = sp_size // 2, 2
            self.parallel_manager = dsp.ParallelManager(dp_size or 1, cp_size, sp_size)
        self._programs = {}
        if self.parallel_manager.sp_size > 1:
            kw = {} if copy_executor is None else {"copy_executor": copy_executor}
            # (a replaced SequenceParallel keeps its few bytes of peer-to-peer flag memory: a peer may still be raising a flag of the old
            #  layout's last exchange; dsp.PeerExchange.close() is for the owner to call behind a barrier)
            self._sp = dsp.SequenceParallel(self.parallel_manager.sp_group, **kw)
        else:
            self._sp = None
        # comm/compute overlap around the spatial attention (two chunks of frames on two side streams): overlap=True / False, or
        # None = VSYS_DSP_OVERLAP (1 / 0 / auto, default auto).  "auto" overlaps when an exchange is big enough to be worth it:
        # cutting a block's frames in two halves every kernel of the section and adds cross-stream events — measured on one GPU with
        # the wire stubbed (tools/issue_time.py --dsp-rank 8): +2.0 ms per step at config 2 (12 MB per exchange, ~2 ms of wire time
        # per step to hide at best), +2.4 ms at 720p x 128f (73 MB per exchange, ~14 ms of wire time per step).
        # VSYS_DSP_SWITCH = activations | qkv | auto picks what travels (dsp.choose_spatial_switch).
        if overlap is None:
            env = os.environ.get("VSYS_DSP_OVERLAP", "auto")
            overlap = "auto" if env == "auto" else env != "0"
        self._overlap = overlap if self._sp is not None else False
        self._overlap_min_bytes = int(os.environ.get("VSYS_DSP_OVERLAP_MIN_MB", "32")) << 20
        self._switch = os.environ.get("VSYS_DSP_SWITCH", "auto")
        # which frames a rank attends over: "flat" = the (sample, frame) axis scattered as one (default), "sample" = per sample
        # as the reference lays it out (comm.py:282-304); same result bit for bit, fewer padded frames on the busiest rank
        self._scatter = os.environ.get("VSYS_DSP_SCATTER", "flat")
        if self._scatter not in ("flat", "sample"):
            raise ValueError("VSYS_DSP_SCATTER must be flat or sample")
        self._side = [torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device)] if self._overlap else None

    def _overlap_on(self, frames, S_full, width):
        """Is the two-chunk overlap used for a spatial block whose rank holds ``frames`` frames of S_full tokens, ``width`` wide?"""
        if self._overlap == "auto":
            return frames * S_full * width * 2 >= self._overlap_min_bytes   # bytes a rank sends in one exchange (bf16)
        return bool(self._overlap)

    # ------------------------------------------------------------------ helpers
    def get_dynamic_size(self, x):
        _, _, T, H, W = x.shape
        p = self.patch_size
        return (-(-T // p[0]), -(-H // p[1]), -(-W // p[2]))

    def _buf(self, name, shape):
        b = self._ws.get(name)
        n = 1
        for s in shape:
            n *= s
        if b is None or b.numel() < n:
            b = torch.empty(n, dtype=self.dtype, device=self.device)
            self._ws[name] = b
        return b[:n].view(*shape)

    def _pos(self, Hp, Wp, height, width):
        S = Hp * Wp
        base_size = round(S**0.5)
        scale = ((float(height) * float(width)) ** 0.5) / self.input_sq_size
        key = (Hp, Wp, scale, base_size)
        if key not in self._pos_cache:
            self._pos_cache[key] = pos_embed_2d(self.hidden_size, Hp, Wp, scale, base_size).to(
                device=self.device, dtype=self.dtype).contiguous()
        return self._pos_cache[key]

    def _rope(self, T):
        if T not in self._rope_cache:
            c, s = rope_tables(self.rope_freqs, T, self.dtype if self.rope_freqs.dtype != torch.float32 else torch.float32)
            self._rope_cache[T] = (c.to(self.device), s.to(self.device))
        return self._rope_cache[T]

    def _embed_vec(self, vals_f32, prefix):
        w = self.w
        f = ops.timestep_embedding(vals_f32.contiguous(), 256)
        h = ops.linear_small(f, w[prefix + ".mlp.0.weight"], w[prefix + ".mlp.0.bias"], act_out=ops.ACT_SILU)
        return ops.linear_small(h, w[prefix + ".mlp.2.weight"], w[prefix + ".mlp.2.bias"])

    def _encode_text(self, y, mask):
        """encode_text (open_sora_transformer_3d.py:526-537) + every block's kv_linear and attention layout, cached
        while the SAME (y, mask) tensor objects are presented unmodified (they are constant over the steps of one generate()).
        The cache keeps strong references and compares identity + ``_version``: a storage address alone may recur for a
        different prompt once the caching allocator recycles the block; ``reset_text_cache()`` (called by the pipeline at the
        start of every generate()) drops it explicitly."""
        c = self._text_cache
        skip = bool(getattr(self.config, "skip_y_embedder", False))
        if skip and mask is not None and not torch.is_tensor(mask):
            mask = torch.tensor([int(v) for v in mask], dtype=torch.long)   # y_lens handed over as a list (:586-588)
            if c is not None and c.get("lens_key") == tuple(mask.tolist()) and same_tensor(c["y"], y) and c["y_version"] == y._version:
                return c
        if (c is not None and same_tensor(c["y"], y) and c["y_version"] == y._version and same_tensor(c["mask"], mask)
                and (mask is None or c["mask_version"] == mask._version)):
            return c
        w = self.w
        self._programs = {}   # recorded steps hold the previous prompt's K/V addresses
        C, H = self.hidden_size, self.num_heads
        if skip:
            # config.skip_y_embedder (:585-590): ``y`` is ALREADY the y_embedder output, packed [1, sum(y_lens), C], and ``mask``
            # carries the per-sample token counts — the caption projection is skipped, the rest is the same
            if mask is None:
                raise ValueError("skip_y_embedder=True needs the per-sample text lengths in ``mask``")
            y_lens = [int(v) for v in mask.reshape(-1).tolist()]
            if len(set(y_lens)) != 1:
                raise ValueError("cross-attention needs equal text lengths per sample (as the reference's torch_impl view does)")
            B = len(y_lens)
            yp = y.to(device=self.device, dtype=self.dtype).reshape(-1, C).contiguous()
            if yp.shape[0] != sum(y_lens):
                raise ValueError(f"y holds {yp.shape[0]} tokens, y_lens sum to {sum(y_lens)}")
            return self._text_kv(y, mask, yp, y_lens, B, lens_key=tuple(y_lens))
        B, _, L, Cc = y.shape
        yb = y.to(device=self.device, dtype=self.dtype).reshape(B * L, Cc).contiguous()
        h = ops.linear_small(yb, w["y_embedder.y_proj.fc1.weight"], w["y_embedder.y_proj.fc1.bias"], act_out=ops.ACT_GELU_TANH)
        ye = ops.linear_small(h, w["y_embedder.y_proj.fc2.weight"], w["y_embedder.y_proj.fc2.bias"]).view(B, L, C)
        if mask is not None:
            m = mask
            if m.shape[0] != B:
                m = m.repeat(B // m.shape[0], 1)
            m = m.reshape(B, L)
            y_lens = [int(v) for v in m.sum(dim=1).tolist()]
            if len(set(y_lens)) != 1:
                raise ValueError("cross-attention needs equal text lengths per sample (as the reference's torch_impl view does)")
            idx = torch.nonzero(m.reshape(-1) != 0, as_tuple=False).reshape(-1).to(self.device)
            yp = ye.reshape(B * L, C).index_select(0, idx).contiguous()  # masked_select packing
        else:
            y_lens = [L] * B
            yp = ye.reshape(B * L, C)
        return self._text_kv(y, mask, yp, y_lens, B)

    def _text_kv(self, y, mask, yp, y_lens, B, lens_key=None):
        """Every block's kv_linear(y) and its attention layouts from the packed text tokens yp [B * Lk, C]; fills the cache."""
        w, C, H = self.w, self.hidden_size, self.num_heads
        Lk = y_lens[0]
        nblk = 2 * self.depth
        kv_pad = ops.kv_pad_len(Lk)
        kps = torch.zeros(nblk, B, H, kv_pad, ops.HEAD_DIM, dtype=self.dtype, device=self.device)
        vts = torch.zeros(nblk, B, H, ops.VT_ROWS, kv_pad, dtype=self.dtype, device=self.device)
        kv = torch.empty(B * Lk, 2 * C, dtype=self.dtype, device=self.device)
        for i in range(nblk):
            p = self.block_prefix(i) + ".cross_attn.kv_linear"
            if (2 * C) % 192 == 0 and C % 64 == 0:
                ops.gemm(yp, w[p + ".weight"], w[p + ".bias"], out=kv)
            else:
                ops.linear_small(yp, w[p + ".weight"], w[p + ".bias"], out=kv)
            ops.attn_prep_kv(kv[:, :C], kv[:, C:], None, kps[i], vts[i], B, H, Lk)
        self._text_cache = dict(y=y, y_version=y._version, mask=mask, mask_version=None if mask is None else mask._version,
                                y_lens=y_lens, kp=kps, vt=vts, Lk=Lk, lens_key=lens_key)
        return self._text_cache

    # ------------------------------------------------------------------ AdaLN fold
    def _fold_ok(self, ts_host, fkey, x_mask):
        """The fold needs ONE modulation per site: no per-frame conditioning mask, and every sample of the batch at the same
        timestep / fps (the CFG pair of a sampling step is).  Under sequence parallelism every site at rest folds; the spatial
        qkv site folds when the q|k|v travel (the GEMM then runs at rest), not when the modulated activations do."""
        C = self.hidden_size
        return (self.adaln_fold and x_mask is None and C % ops.LN_BLOCK == 0 and C // ops.LN_BLOCK <= 12
                and bool((ts_host == ts_host[0]).all()) and len(set(fkey[:-1])) == 1)

    def _fold_tables(self, B):
        """Site table of vsys_adaln_prescale for a modulation table laid out [2*depth, B, 6C] (sample-0 rows) and the per-site
        W' / cs / cv buffers (allocated once: 2 x the qkv + fc1 weight bytes)."""
        f = self._fold or None     # (HostOffload empties it when the weights leave the device: their addresses are in the table)
        if f is not None and f["B"] == B:
            return f
        w, C, dev = self.w, self.hidden_size, self.device
        bufs = f["bufs"] if f is not None else {}
        rows, blk = [], 0
        for i in range(2 * self.depth):
            p = self.block_prefix(i)
            for name, s_sh, s_sc in ((p + ".attn.qkv", 0, C), (p + ".mlp.fc1", 3 * C, 4 * C)):
                W, b = w[name + ".weight"], w[name + ".bias"]
                N, K = W.shape
                if name not in bufs:
                    bufs[name] = (torch.empty_like(W), torch.empty(N, dtype=torch.float32, device=dev),
                                  torch.empty(N, dtype=torch.float32, device=dev))
                Wp, cs, cv = bufs[name]
                base = i * B * 6 * C
                rows.append([W.data_ptr(), b.data_ptr(), Wp.data_ptr(), cs.data_ptr(), cv.data_ptr(), base + s_sh, base + s_sc,
                             N, K, blk])
                blk += -(-N // 4)
        self._fold = dict(B=B, bufs=bufs, sites=torch.tensor(rows, dtype=torch.int64).to(dev), nblocks=blk)
        return self._fold

    def _ln_stats(self, N):
        key = ("ln_stats", N)
        if key not in self._ws:
            self._ws[key] = ops.ln_stats_buffer(N, self.hidden_size, self.device)
        return self._ws[key]

    def reset_text_cache(self):
        """Forget the per-prompt text projections (and their references to the prompt tensors) and the recorded steps."""
        self._text_cache = None
        self._programs = {}

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, x, timestep, y, all_timesteps=None, mask=None, x_mask=None, fps=None, height=None, width=None,
                **kwargs):
        w, C, H = self.w, self.hidden_size, self.num_heads
        # === Split batch === (:545-557): rank group cp_rank keeps its rows of every per-sample input
        pm = self.parallel_manager
        cp = getattr(pm, "cp_size", 1) or 1
        if cp > 1:
            Bfull = x.shape[0]
            if Bfull % cp:
                raise ValueError(f"batch {Bfull} is not divisible by cp_size {cp}")
            Bl = Bfull // cp
            sl = slice(pm.cp_rank * Bl, (pm.cp_rank + 1) * Bl)
            rows = lambda v: v[sl] if (torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == Bfull) else v
            x, timestep, y, fps, height, width, mask, x_mask = (rows(v) for v in (x, timestep, y, fps, height, width, mask, x_mask))
        B, _, Tx, Hx, Wx = x.shape
        T, Hp, Wp = self.get_dynamic_size(x)
        dev = self.device
        if x_mask is not None and not x_mask.is_cuda and bool(x_mask.to(torch.bool).all()):
            x_mask = None        # an all-True mask IS the plain step (every frame takes the timestep's modulation)
        if x_mask is not None:   # [B, T] bool: False = conditioning frame (sees the timestep-0 modulation, :181-184,578-582)
            if tuple(x_mask.shape) != (B, T):
                raise ValueError(f"x_mask must be [B, T] = [{B}, {T}], got {tuple(x_mask.shape)}")
            x_mask = x_mask.to(device=dev, dtype=torch.bool).contiguous()

        # ---- host-side, per step: the timestep (the sampler hands timesteps over as HOST tensors, so the integer the PAB policy
        # needs is available without a device sync; timestep.to(dtype) (:562) then .float() inside the embedder), the PAB
        # decisions of every block, and the step-invariant tables (all cached: text K/V, position table, fps embedding, RoPE)
        ts_host = timestep.detach().to("cpu").to(self.dtype).float()
        timestep_int = int(ts_host[0]) if pab.enable_pab() else None
        valid_depth = kwargs.get("valid_depth", self.depth)
        plan = self._pab_plan(timestep_int, all_timesteps, valid_depth)
        txt = self._encode_text(y, mask)
        pos = self._pos(Hp, Wp, height[0], width[0])
        fkey = tuple(float(v) for v in fps.reshape(-1).tolist()) + (B,)
        if fkey not in self._fps_cache:
            f = fps.to(dev).float().reshape(-1)
            if f.numel() != B:
                f = f.repeat(B // f.numel())
            self._fps_cache[fkey] = self._embed_vec(f, "fps_embedder")
        self._rope(T)
        static = (txt, pos, self._fps_cache[fkey])
        fold = self._fold_ok(ts_host, fkey, x_mask)

        try:
            return self._issue_step(x, ts_host, static, plan, timestep_int, valid_depth, cp, x_mask, fold, fkey, height, width)
        except BaseException:
            # _pab_plan marked the slabs this step was going to write as valid; the step did not finish (launch error, OOM, an
            # interrupted caller), so no slab may be broadcast until its block has computed again
            for st in self.states:
                st.attn_valid = st.cross_valid = False
            raise

    def _issue_step(self, x, ts_host, static, plan, timestep_int, valid_depth, cp, x_mask, fold, fkey, height, width):
        """The device side of forward(): eagerly, or through the recorded launch program of this (geometry, PAB pattern, layout)."""
        dev = self.device
        B, _, Tx, Hx, Wx = x.shape
        mlp_action = plan is not None and any(d[2] or d[3] for d in plan)   # stores / replays host-side dict entries: eager
        # (a conditioning mask changes from step to step and selects rows with torch ops: the step is issued eagerly)
        if not (self.use_programs and self._hidden_tap is None and not mlp_action and x_mask is None):
            self.program_stats["eager"] += 1
            xz = x.to(device=dev, dtype=torch.float32).contiguous()
            out = self._forward_device(xz, ts_host.to(dev), static, plan, timestep_int, valid_depth, cp, x_mask, fold=fold)
        else:
            sp = self._sp
            key = (B, Tx, Hx, Wx, float(height[0]), float(width[0]), fkey, valid_depth, cp, fold, self.fold_spatial_qkv,
                   None if plan is None else tuple(d[:2] + d[5:7] for d in plan),
                   None if sp is None else (sp.P, sp.rank, self._scatter, self._switch, self._overlap))
            ent = self._programs.get(key)
            if ent is not None:       # replay: refresh the two per-step inputs, ONE C call per launch segment
                prog, xin, tin, out = ent
                xin.copy_(x, non_blocking=True)
                tin.copy_(ts_host, non_blocking=True)
                prog.run()
                self.program_stats["replayed"] += 1
                out = out.clone()     # the program owns its output buffer; callers keep what they are handed
            else:                     # first time for this key: run eagerly under the recorder
                xin = torch.empty(B, x.shape[1], Tx, Hx, Wx, dtype=torch.float32, device=dev)
                tin = torch.empty(B, dtype=torch.float32, device=dev)
                xin.copy_(x)
                tin.copy_(ts_host)
                with program.Recorder() as rec:
                    out = self._forward_device(xin, tin, static, plan, timestep_int, valid_depth, cp, fold=fold)
                prog = rec.finish()
                if prog is not None:
                    self._programs[key] = (prog, xin, tin, out)
                    self.program_stats["recorded"] += 1
                    out = out.clone()
                else:
                    self.program_stats["eager"] += 1
        return out

    def _pab_plan(self, timestep_int, all_timesteps, valid_depth):
        """The PAB decisions of every block of this step, made up front in the order the reference makes them inside its blocks
        (attention, cross, MLP per block: open_sora_transformer_3d.py:186-190,230-234,244-250) with the same per-block counters.
        None when PAB is off; else per block (broadcast_attn, broadcast_cross, broadcast_mlp, store_mlp, skip_range)."""
        if not pab.enable_pab():
            return None
        plan = []
        cfg = pab.PAB_MANAGER.config
        mlp_on = cfg.mlp_broadcast
        if mlp_on and all_timesteps is None:
            raise ValueError("PAB mlp_broadcast needs the sampler's schedule: call the model with all_timesteps=[...]")
        sched = [int(v) for v in all_timesteps] if all_timesteps is not None else None
        ats = sched if mlp_on else None
        # A computed attention output is kept (the GEMM epilogue writes the slab, +20 us per gated GEMM) only if the block's NEXT
        # call will broadcast it: by the reference's rule (pab_mgr.py:54-91) the next call either broadcasts or recomputes, and a
        # recompute rewrites the slab.  The next call's timestep is known when the sampler handed the schedule over and the current
        # timestep sits on it exactly once before its end; otherwise the slab is always written (as before).  Outside the
        # broadcast window — 12 of the 30 steps of config 3 — nothing is kept.
        t_next = None
        if self.pab_elide_unused and sched is not None and sched.count(timestep_int) == 1 and sched.index(timestep_int) + 1 < len(sched):
            t_next = sched[sched.index(timestep_int) + 1]

        def kept(kind, count_next):
            if t_next is None:
                return True
            on, every, thr = cfg.attn_rule(kind)
            return bool(on) and thr is not None and count_next % every != 0 and thr[0] < t_next < thr[1]

        for i in range(2 * valid_depth):
            st = self.states[i]
            kind = "temporal" if st.temporal else "spatial"
            fn = pab.if_broadcast_temporal if st.temporal else pab.if_broadcast_spatial
            b_attn, st.attn_count = fn(timestep_int, st.attn_count)
            b_cross, st.cross_count = pab.if_broadcast_cross(timestep_int, st.cross_count)
            b_mlp, b_next, rng = False, False, None
            if mlp_on:
                b_mlp, st.mlp_count, b_next, rng = pab.if_broadcast_mlp(timestep_int, st.mlp_count, st.block_idx, ats,
                                                                         is_temporal=st.temporal)
            # a broadcast of a slab that was never written (or was elided at the block's last computed call because the schedule
            # said nobody would read it): recompute instead — direct transformer use, a sampler that repeats or skips a step
            if b_attn and not st.attn_valid:
                b_attn = False
            if b_cross and not st.cross_valid:
                b_cross = False
            keep_a, keep_c = kept(kind, st.attn_count), kept("cross", st.cross_count)
            if not b_attn:
                st.attn_valid = keep_a       # (this call computes; the slab is written iff it is kept)
            if not b_cross:
                st.cross_valid = keep_c
            plan.append((bool(b_attn), bool(b_cross), bool(b_mlp), bool(b_next), rng, keep_a, keep_c))
        return plan

    def _forward_device(self, xz, ts, static, plan, timestep_int, valid_depth, cp, x_mask=None, fold=False):
        """Everything of a step that runs on the device, from resident inputs: xz fp32 [B, C_in, T, H, W], ts fp32 [B];
        x_mask bool [B, T] on the device or None."""
        w, C = self.w, self.hidden_size
        txt, pos, fps_emb = static
        B, _, Tx, Hx, Wx = xz.shape
        T, Hp, Wp = self.get_dynamic_size(xz)
        S = Hp * Wp
        dev = self.device
        pm = self.parallel_manager

        # ---- per-step vectors (tiny): t, fps, t_mlp and all modulation rows
        t = self._embed_vec(ts, "t_embedder")
        ops.add_rows(t, fps_emb)
        t_mlp = ops.linear_small(t, w["t_block.1.weight"], w["t_block.1.bias"], act_in=ops.ACT_SILU)  # [B, 6C]
        t0 = None
        if x_mask is not None:
            # conditioning frames are modulated with the embedding of timestep 0 (:578-582).  The blocks choose between the two
            # modulations per frame with torch.where on the modulated / gated activations (t_mask_select, :152-160); choosing the
            # modulation VECTOR per (sample, frame) and running every kernel with "rows per sample" = the rows of one frame is
            # the same selection, bit for bit: the table below has one row per (sample, frame).
            t0 = self._embed_vec(torch.zeros_like(ts), "t_embedder")
            ops.add_rows(t0, fps_emb)
            t0_mlp = ops.linear_small(t0, w["t_block.1.weight"], w["t_block.1.bias"], act_in=ops.ACT_SILU)
            t_mlp = torch.where(x_mask[:, :, None], t_mlp[:, None, :], t0_mlp[:, None, :]).reshape(B * T, -1).contiguous()
        mod = ops.mod_table(w["_all_tables"], t_mlp)  # [2*depth, B, 6C]  (x_mask: [2*depth, B*T, 6C])
        ftab = None
        if fold:   # W' = bf16(W (1 + scale)), cs, cv of all 2 x 2 x depth sites of this step: one launch
            ftab = self._fold_tables(B)
            ops.adaln_prescale(ftab["sites"], ftab["nblocks"], mod)
        self._stats_fresh = False

        # ---- x embed (+ pos).  Sequence parallel: split_sequence(x, dim=2) (:598-603) keeps tokens rank*Sl .. of every frame, so
        # only those are embedded (the reference embeds the whole frame on every rank and slices)
        sp = self._sp
        S_full = S
        if sp is not None:
            S = -(-S_full // sp.P)
            xe = ops.patch_embed_shard(xz, w["x_embedder.proj.weight"], w["x_embedder.proj.bias"], pos, B, self.patch_size, C,
                                       sp.rank * S, S)   # [B, T, S/P, C], zero rows past the frame's last token
        else:
            xe = ops.patch_embed(xz, w["x_embedder.proj.weight"], w["x_embedder.proj.bias"], pos, B, self.patch_size, C)
        xcur = xe.view(B * T * S, C)

        # which sub-blocks (attention, cross, MLP) of which block are PAB broadcasts this step: what `upcoming` in _block looks ahead
        # into.  (A hidden-state tap reads x between blocks: nothing may then be applied ahead of its place.)
        self._bc, self._folded = None, {}
        if plan is not None and self.pab_fold_adds and self._hidden_tap is None:
            self._bc = [(d[0], d[1], d[2], d[4]) for d in plan[:2 * valid_depth]]
        for d in range(valid_depth):
            for i in (2 * d, 2 * d + 1):
                xcur = self._block(i, xcur, mod[i], txt, B, T, S, S_full, None if plan is None else plan[i], timestep_int,
                                   rps=S if x_mask is not None else T * S, ftab=ftab)
            if self._hidden_tap is not None:
                self._hidden_tap(d, xcur)

        fin = (w["final_layer.scale_shift_table"], w["final_layer.linear.weight"], w["final_layer.linear.bias"])
        x_zero = None
        if x_mask is not None:
            # T2IFinalLayer with a mask (:82-85): as written there the conditioning branch normalises the ALREADY modulated x a
            # second time, x_zero = mod_0(LN(mod_t(LN(x)))) — kept, the golden fixture pins it.  mod_t(LN(x)) is one AdaLN
            # launch; the fused final-layer kernel then runs on it with the timestep-0 vector, and the frames are chosen below.
            ss = fin[0][None] + t[:, None]                         # bf16 [B, 2, C]: shift | scale of t
            x_zero = ops.adaln_modulate(xcur, ss[0, 0], ss[0, 1], T * S, 2 * C, out=self._buf("xm", (B * T * S, C)))
        if sp is not None:
            # gather_sequence (:615-619) + final layer + unpatchify: the final layer is per token, so it runs on the local rows and
            # its 32 fp32 values per token are gathered (0.6 MB per rank at config 2) instead of the 1152-wide hidden state (11 MB)
            tok = ops.final_layer_tokens(xcur, fin[0], t, fin[1], fin[2], B, T, S)
            if x_mask is not None:
                tok = torch.where(x_mask[:, :, None, None], tok, ops.final_layer_tokens(x_zero, fin[0], t0, fin[1], fin[2], B, T, S))
            allt = torch.empty(sp.P * B, *tok.shape[1:], dtype=tok.dtype, device=dev)   # [P][B, T, Sl, n] stacked along dim 0
            dsp.all_gather_into_tensor(allt, tok, sp.group)
            out = ops.unpatchify_tokens(allt, sp.P, B, T, S, Hp, Wp, Hx, Wx, self.patch_size, self.out_channels)
        else:
            out = ops.final_layer(xcur, fin[0], t, fin[1], fin[2], B, T, Hp, Wp, Hx, Wx, self.patch_size, self.out_channels)
            if x_mask is not None:
                out0 = ops.final_layer(x_zero, fin[0], t0, fin[1], fin[2], B, T, Hp, Wp, Hx, Wx, self.patch_size, self.out_channels)
                out = torch.where(x_mask[:, None, :, None, None], out, out0)
        if cp > 1:  # gather_sequence(x, cp_group, dim=0) (:621)
            parts = torch.empty(cp * B, *out.shape[1:], dtype=out.dtype, device=dev)
            dsp.all_gather_into_tensor(parts, out.contiguous(), pm.cp_group)
            out = parts
        return out

    __call__ = forward

    def _block(self, i, x, mod_i, txt, B, T, S, S_full, decisions, timestep_int, rps=None, ftab=None):
        """STDiT3Block.forward (open_sora_transformer_3d.py:162-286). x: [B*T*S, C] (S = local shard), updated in place.
        ``decisions`` = this block's entry of _pab_plan (None: PAB off).  ``rps``: rows that share one modulation row of ``mod_i``
        — a sample's T*S rows, or one frame's S rows when a conditioning mask picks the modulation per frame."""
        rps = T * S if rps is None else rps
        w, C, H = self.w, self.hidden_size, self.num_heads
        p = self.block_prefix(i)
        st = self.states[i]
        temporal = st.temporal
        N = B * T * S
        C6 = 6 * C
        shift_msa, scale_msa, gate_msa = mod_i[0, 0:C], mod_i[0, C:2 * C], mod_i[0, 2 * C:3 * C]
        shift_mlp, scale_mlp, gate_mlp = mod_i[0, 3 * C:4 * C], mod_i[0, 4 * C:5 * C], mod_i[0, 5 * C:6 * C]
        _buf = self._buf

        def slab(cur):   # a PAB slab the shape of x (allocated on first use)
            return cur if cur is not None and cur.shape == x.shape else torch.empty_like(x)

        use_pab = decisions is not None
        broadcast_attn, broadcast_cross, broadcast_mlp, broadcast_next, skip_range, keep_attn, keep_cross = \
            decisions or (False, False, False, False, None, False, False)
        sp = self._sp
        # AdaLN fold (ftab: this step's pre-scaled weights): the qkv / fc1 GEMMs read x itself and apply LayerNorm + modulation
        # in their epilogues; the row statistics come from the epilogue that last wrote x (gemm_stats) or, when x was last
        # written by something else (patch embedding, proj, a PAB broadcast), from one ln_row_stats pass.
        fold = ftab is not None
        stats = self._ln_stats(N) if fold else None

        def folded(site, gelu, out):
            if not self._stats_fresh:
                ops.ln_row_stats(x, stats)
                self._stats_fresh = True
            Wp, cs, cv = ftab["bufs"][site]
            return ops.gemm_ln(x, Wp, cs, cv, stats, gelu=gelu, out=out)

        def upcoming(k):
            """The cached outputs of the (at most two) broadcast sub-blocks that directly follow sub-block ``k`` (0 attention,
            1 cross, 2 MLP) of this block in program order — the next reader of x comes after them.  They are marked as applied:
            the GEMM that writes x at sub-block k adds them in its store phase, in order, one bf16 rounding each."""
            got = []
            if self._bc is None:
                return got
            j, kk = i, k
            while len(got) < 2:
                kk += 1
                if kk == 3:
                    j, kk = j + 1, 0
                if j >= len(self._bc) or not self._bc[j][kk]:
                    break
                sj = self.states[j]
                if kk == 2:
                    cached = pab.get_mlp_output(self._bc[j][3], timestep=timestep_int, block_idx=sj.block_idx, is_temporal=sj.temporal)
                else:
                    cached = sj.last_attn if kk == 0 else sj.last_cross
                if cached is None or cached.shape != x.shape or cached.stride() != x.stride():
                    if kk == 2:
                        self._folded[(j, kk)] = (cached, False)   # fetched (the store may have dropped it) but not applied
                    break
                self._folded[(j, kk)] = (cached, True)
                got.append(cached)
            return got

        def applied(k):   # was this broadcast sub-block already added by an earlier GEMM's store phase?
            return self._folded.get((i, k), (None, False))[1]

        def write_x(a, wname, k, gate=None, aux=None, want_stats=False):
            """x = x + gate * Linear(a) (+ the broadcasts that follow) — the GEMM that closes sub-block k"""
            adds = upcoming(k)
            W, bias = w[wname + ".weight"], w[wname + ".bias"]
            gk = dict(gate=gate, gate_stride=C6, rows_per_sample=rps) if gate is not None else {}
            if adds or (fold and aux is not None and (want_stats or adds)):
                ops.gemm_gate_res_add(a, W, bias, res=x, aux=aux, adds=adds, stats=stats if fold else None, out=x, **gk)
                self._stats_fresh = fold
            elif fold and aux is None and want_stats:
                ops.gemm_stats(a, W, bias, stats, res=x, out=x, **gk)
                self._stats_fresh = True
            else:
                ops.gemm(a, W, bias, epilogue=ops.EPI_GATE_RES, res=x, aux=aux, out=x, **gk)
                self._stats_fresh = False

        # which form the qkv site of this block takes (see _fold_ok)
        sp_order = None
        if sp is not None and not temporal:
            flat = T == 1 or self._scatter == "flat"
            sp_order = self._switch_order(*((1, B * T) if flat else (B, T)), S_full)
        fold_attn = fold and (temporal or (sp is None and self.fold_spatial_qkv) or sp_order == "qkv")

        # ---------------- self attention
        if broadcast_attn:
            if not applied(0):
                ops.add_rows(x, st.last_attn)
                self._stats_fresh = False
        else:
            xm = None
            if fold_attn and sp_order is None:
                qkv = folded(p + ".attn.qkv", False, _buf("qkv", (N, 3 * C)))
            elif not fold_attn:
                xm = ops.adaln_modulate(x, shift_msa, scale_msa, rps, C6, out=_buf("xm", (N, C)))
            aux = None
            if use_pab and keep_attn:
                st.last_attn = slab(st.last_attn)
                aux = st.last_attn
            if temporal:
                if not fold_attn:
                    qkv = ops.gemm(xm, w[p + ".attn.qkv.weight"], w[p + ".attn.qkv.bias"], out=_buf("qkv", (N, 3 * C)))
                ao = _buf("attn_out", (N, C))
                cos, sin = self._rope(T)
                ops.attn_temporal(qkv, C, w[p + ".attn.q_norm.weight"], w[p + ".attn.k_norm.weight"], cos, sin, ao, B, T, S, H)
            elif sp is None:
                if not fold_attn:
                    qkv = ops.gemm(xm, w[p + ".attn.qkv.weight"], w[p + ".attn.qkv.bias"], out=_buf("qkv", (N, 3 * C)))
                kp, vt = self._kv_spatial(B * T, S)
                ops.attn_prep_kv(qkv[:, C:2 * C], qkv[:, 2 * C:], w[p + ".attn.k_norm.weight"], kp, vt, B * T, H, S)
                ao = _buf("attn_out", (N, C))
                ops.flash_attn(qkv[:, :C], w[p + ".attn.q_norm.weight"], kp, vt, ao, B * T, H, S, S, k_norm_bound=self._kbound(p))
            else:
                qkv_rest = None
                if fold_attn:   # (order "qkv") the folded GEMM at rest produces what travels
                    qkv_rest = folded(p + ".attn.qkv", False, self._buf("qkv_rest", (N, 3 * C)))
                ao = self._spatial_attn_sharded(p, xm, B, T, S, S_full, qkv_rest=qkv_rest)
            write_x(ao, p + ".attn.proj", 0, gate=gate_msa, aux=aux)   # (cross attention reads x next: no statistics wanted)

        # ---------------- cross attention (no norm, no modulation, no gate)
        if broadcast_cross:
            if not applied(1):
                ops.add_rows(x, st.last_cross)
                self._stats_fresh = False
        else:
            q = ops.gemm(x, w[p + ".cross_attn.q_linear.weight"], w[p + ".cross_attn.q_linear.bias"], out=_buf("xm", (N, C)))
            ao = _buf("attn_out", (N, C))
            # (the hoisted text K / V were prepared for exactly Lk keys on zeroed buffers, _text_kv: the padding promise holds)
            ops.flash_attn(q, None, txt["kp"][i], txt["vt"][i], ao, B, H, T * S, txt["Lk"], keys_exact=True)
            aux = None
            if use_pab and keep_cross:
                st.last_cross = slab(st.last_cross)
                aux = st.last_cross
            # the MLP's norm2 reads this x next (unless the MLP is a broadcast nobody folded): emit its statistics here
            write_x(ao, p + ".cross_attn.proj", 1, aux=aux, want_stats=not broadcast_mlp or self._bc is not None)

        # ---------------- MLP (+ PAB MLP broadcast, open_sora_transformer_3d.py:232-280 / pab_mgr.py:93-174: inside a configured
        # window the block replays gate_mlp * mlp(...) of the window's first timestep).  ``all_timesteps`` reaches the blocks
        # here; the reference's STDiT3.forward forgets to pass it on and raises TypeError with mlp_broadcast=True.
        if broadcast_mlp:
            if (i, 2) in self._folded:   # looked at by an earlier GEMM of this step (the store hands an entry out once at a window's end)
                slab = self._folded[(i, 2)][0]
            else:
                slab = pab.get_mlp_output(skip_range, timestep=timestep_int, block_idx=st.block_idx, is_temporal=temporal)
            if not applied(2):
                ops.add_rows(x, slab)
                self._stats_fresh = False
            if timestep_int == skip_range[-1]:   # the window closed (the store dropped the entry): the slab is free again,
                self._ws.setdefault("mlp_slab_pool", []).append(slab)   # in stream order behind the add above
            return x
        hdim = w[p + ".mlp.fc1.weight"].shape[0]
        # (Measured in round 6 and not kept: the MLP one CFG sample at a time through a half-size hidden buffer, so that fc2's A operand is
        #  179 MB written a moment ago instead of 359 MB — same bits; fc2 + cross-proj 1005 -> 1021 TFLOP/s, but the half-size fc1 launches
        #  993 -> 922: 97.0 -> 97.8 ms per step.)
        if fold:
            hbuf = folded(p + ".mlp.fc1", True, _buf("mlp_h", (N, hdim)))
        else:
            xm = ops.adaln_modulate(x, shift_mlp, scale_mlp, rps, C6, out=_buf("xm", (N, C)))
            hbuf = ops.gemm(xm, w[p + ".mlp.fc1.weight"], w[p + ".mlp.fc1.bias"], epilogue=ops.EPI_BIAS_GELU,
                            out=_buf("mlp_h", (N, hdim)))
        aux = self._mlp_slab(x) if broadcast_next else None   # the post-gate output, written by the fc2 epilogue
        write_x(hbuf, p + ".mlp.fc2", 2, gate=gate_mlp, aux=aux, want_stats=True)   # the next block's norm1 reads this x
        if broadcast_next:
            pab.save_mlp_output(timestep=timestep_int, block_idx=st.block_idx, ff_output=aux, is_temporal=temporal)
        return x

    def _spatial_attn_sharded(self, p, xm, B, T, S, S_full, qkv_rest=None):
        """The DSP section of a spatial block (dynamic_switch, open_sora_transformer_3d.py:208-216,288-315): modulated activations
        [B,T,S/P,C] -> all-to-all -> T-shard [.., S, C] -> qkv GEMM -> spatial attention -> all-to-all -> [B,T,S/P,C].

        Which frames a rank gets ("scatter"): "sample" is the reference's layout (T padded and scattered per sample); "flat"
        (default) scatters the (sample, frame) axis as ONE axis of B*T frames — [B,T,S/P,C] is the same memory as
        [1,B*T,S/P,C], spatial attention is per frame and the qkv GEMM per row, so the result is the same bit for bit while the
        busiest rank holds ceil(B*T/P) instead of B*ceil(T/P) frames.  The reference's image case (T == 1: the batch is scattered,
        :288-303) is the flat layout by definition.

        What travels ("order"): the C-wide activations (reference order, qkv GEMM on the T-shard) or the 3C-wide q|k|v
        (GEMM at rest on the un-padded shard), dsp.choose_spatial_switch.

        Overlap: the block's frames are cut in two chunks that run on two side streams with the collectives ISSUED in the order
        A1, B1, A2, B2 (a communicator executes collectives in issue order): B's first exchange travels while A computes, A's
        second while B computes.  Every op is per frame / per row, so any cut gives the batched result bit for bit.  Chunks are
        the two CFG samples ("sample") or the two halves of every rank's frame block ("flat")."""
        w, C, H, sp = self.w, self.hidden_size, self.num_heads, self._sp
        flat = T == 1 or self._scatter == "flat"
        Bv, Tv = (1, B * T) if flat else (B, T)
        Tp = -(-Tv // sp.P)                        # frames of one sample view on this rank (padded)
        order = self._switch_order(Bv, Tv, S_full)
        wide = 3 * C if order == "qkv" else C
        src = xm
        if order == "qkv":   # qkv GEMM at rest on the un-padded S-shard; the 3C-wide q|k|v travels
            src = qkv_rest if qkv_rest is not None else ops.gemm(xm, w[p + ".attn.qkv.weight"], w[p + ".attn.qkv.bias"],
                                                                 out=self._buf("qkv_rest", (B * T * S, 3 * C)))
        src4 = src.view(Bv, Tv, S, wide)
        back = self._buf("attn_back", (Bv, Tv, S, C))

        # the chunks: (view of the source, chunk of the frame block or None, Bc, frames per sample view)
        overlap = self._overlap_on(Bv * Tp, S_full, wide)
        if overlap and order != "qkv" and not flat and B == 2:
            chunks = [(src4[i:i + 1], None, 1, Tp, back[i:i + 1]) for i in range(2)]
        elif overlap and order != "qkv" and flat and Tp >= 2:
            h = -(-Tp // 2)
            chunks = [(src4, (0, h), Bv, h, back), (src4, (h, Tp), Bv, Tp - h, back)]
        else:
            chunks = [(src4, None, Bv, Tp, back)]

        def attend(i, xt, Bc, Tc):
            """qkv -> K/V layouts -> flash attention on Bc*Tc whole frames; xt: [Bc, Tc, S_full, wide]"""
            nf = Bc * Tc
            Na = nf * S_full
            if order == "qkv":
                qkv = xt.view(Na, 3 * C)
            else:
                qkv = ops.gemm(xt.view(Na, C), w[p + ".attn.qkv.weight"], w[p + ".attn.qkv.bias"], out=self._buf(f"qkv_o{i}", (Na, 3 * C)))
            key = ("kv_spatial_o", i, nf, S_full)
            if key not in self._ws:
                self._ws[key] = ops.alloc_kv_buffers(nf, H, S_full, self.device)
            kp, vt = self._ws[key]
            ops.attn_prep_kv(qkv[:, C:2 * C], qkv[:, 2 * C:], w[p + ".attn.k_norm.weight"], kp, vt, nf, H, S_full)
            ao = self._buf(f"attn_out_o{i}", (Na, C))
            ops.flash_attn(qkv[:, :C], w[p + ".attn.q_norm.weight"], kp, vt, ao, nf, H, S_full, S_full, k_norm_bound=self._kbound(p))
            return ao.view(Bc, Tc, S_full, C)

        if len(chunks) == 1:
            x4, ck, Bc, Tc, out = chunks[0]
            xt = sp.to_temporal_shard(x4, S_full, out=self._buf("sp_xt0", (Bc, Tc, S_full, wide)))
            sp.to_spatial_shard(attend(0, xt, Bc, Tc), Tv, S, out=out)
            return back.view(B * T * S, C)

        # Cross-stream ordering is host-side work (event record / wait): under a launch-program recorder each of these runs again
        # on every replay (program.host_call), on the stream that is current here.
        ev_in, ev_done = torch.cuda.Event(), [torch.cuda.Event(), torch.cuda.Event()]
        program.host_call(lambda: ev_in.record())                       # main stream: xm (or q|k|v) is ready
        xts = [None, None]
        for i, (x4, ck, Bc, Tc, out) in enumerate(chunks):   # phase 1: both first exchanges
            with torch.cuda.stream(self._side[i]):
                program.host_call(lambda: torch.cuda.current_stream().wait_event(ev_in))
                xts[i] = sp.to_temporal_shard(x4, S_full, tag=f"_{i}", chunk=ck, out=self._buf(f"sp_xt{i}", (Bc, Tc, S_full, wide)))
        for i, (x4, ck, Bc, Tc, out) in enumerate(chunks):   # phase 2: per-chunk attention, then the exchange back
            with torch.cuda.stream(self._side[i]):
                ao = attend(i, xts[i], Bc, Tc)
                sp.to_spatial_shard(ao, x4.shape[1], S, out=out, tag=f"_{i}", chunk=ck, Tp=Tp if ck is not None else None)
                program.host_call(lambda e=ev_done[i]: e.record())
            program.host_call(lambda e=ev_done[i]: torch.cuda.current_stream().wait_event(e))   # main waits for the chunk
        return back.view(B * T * S, C)

    def _switch_order(self, B, T, S_full):
        """What travels through the DSP exchange of a spatial block: "activations" (reference order) or "qkv"."""
        if self._switch in ("activations", "qkv"):
            return self._switch
        key = ("switch", B, T, S_full)
        if key not in self._ws:
            self._ws[key] = dsp.choose_spatial_switch(B, T, S_full, self.hidden_size, self._sp.P, overlapped=bool(self._overlap),
                                                      scatter="sample")["order"]   # (called with the scattered view's B, T)
        return self._ws[key]

    def _kbound(self, p):
        """The promise about block ``p``'s spatial keys that lets the attention kernels drop the running max (ops.rms_key_bound:
        from the q / k norm weights, once; None = no promise, e.g. weights with outliers).  VSYS_FLASH_STATIC=0 never promises."""
        kb = self._kbounds.get(p, 0)
        if kb == 0:
            kb = None
            if os.environ.get("VSYS_FLASH_STATIC", "1") != "0":
                kb = ops.rms_key_bound(self.w[p + ".attn.q_norm.weight"], self.w[p + ".attn.k_norm.weight"])
            self._kbounds[p] = kb
        return kb

    def _kv_spatial(self, batch, kv_len):
        key = ("kv_spatial", batch, kv_len)
        if key not in self._ws:
            self._ws[key] = ops.alloc_kv_buffers(batch, self.num_heads, kv_len, self.device)
        return self._ws[key]

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
        """Counters to zero and every stored MLP output dropped (an aborted generate() must not pin its slabs, nor may the next
        prompt replay them)."""
        for st in self.states:
            st.attn_count = st.cross_count = st.mlp_count = 0
            st.attn_valid = st.cross_valid = False
        if pab.PAB_MANAGER is not None:
            pab.PAB_MANAGER.config.mlp_spatial_outputs.clear()
            pab.PAB_MANAGER.config.mlp_temporal_outputs.clear()


def _from_pretrained(cls, name, device="cuda", **kwargs):
    """STDiT3.from_pretrained (open_sora_transformer_3d.py:661-663; the hub download of HF's PreTrainedModel, third-party, is not
    available offline): ``name`` is a LOCAL checkpoint directory holding the ``*.safetensors`` file(s) (+ optional ``config.json``) with the
    reference's keys, or ``"synthetic:<seed>"`` for seeded random weights of the configured geometry."""
    from .utils import read_component

    file_cfg, sd = read_component(name)
    known = set(STDiT3Config().__dict__)
    cfg = {k: (tuple(v) if isinstance(v, list) else v) for k, v in file_cfg.items() if k in known}
    cfg.update(kwargs)
    model = cls(STDiT3Config(**cfg), device=device)
    if sd is None:
        if not (isinstance(name, str) and name.startswith("synthetic:")):
            raise FileNotFoundError(f"STDiT3.from_pretrained({name!r}): not a local checkpoint directory (*.safetensors) and not "
                                    "'synthetic:<seed>' — hub ids cannot be fetched on this box")
        sd = synth_state_dict(model.config, seed=int(name.split(":", 1)[1]))
    model.load_state_dict(sd)
    return model


STDiT3.from_pretrained = classmethod(_from_pretrained)


def STDiT3_XL_2(from_pretrained=None, **kwargs):
    """open_sora_transformer_3d.py:661-667: the XL/2 geometry (depth 28, width 1152, 16 heads, patch (1, 2, 2)); weights from
    ``from_pretrained`` (see STDiT3.from_pretrained) or left unset for the caller's ``load_state_dict``."""
    device = kwargs.pop("device", "cuda")
    if from_pretrained is not None:
        return STDiT3.from_pretrained(from_pretrained, device=device, **kwargs)
    return STDiT3(STDiT3Config(depth=28, hidden_size=1152, patch_size=(1, 2, 2), num_heads=16, **kwargs), device=device)


def synth_state_dict(config: STDiT3Config, seed: int = 1234) -> Dict[str, torch.Tensor]:
    """Seeded random-init weights with the checkpoint's key names/shapes (no pretrained weights offline).
    Generated on the host CPU generator so every rank / the oracle see identical values."""
    g = torch.Generator().manual_seed(seed)
    C, D = config.hidden_size, config.hidden_size // config.num_heads
    Hm = int(config.hidden_size * config.mlp_ratio)
    out_ch = config.in_channels * 2 if config.pred_sigma else config.in_channels
    sd: Dict[str, torch.Tensor] = {}

    def lin(name, n_out, n_in):
        s = min(0.08, 1.0 / math.sqrt(n_in))
        sd[name + ".weight"] = torch.randn(n_out, n_in, generator=g) * s
        sd[name + ".bias"] = torch.randn(n_out, generator=g) * 0.02

    sd["x_embedder.proj.weight"] = torch.randn(C, config.in_channels, *config.patch_size, generator=g) * 0.1
    sd["x_embedder.proj.bias"] = torch.randn(C, generator=g) * 0.02
    for e in ("t_embedder", "fps_embedder"):
        lin(e + ".mlp.0", C, 256)
        lin(e + ".mlp.2", C, C)
    lin("t_block.1", 6 * C, C)
    lin("y_embedder.y_proj.fc1", C, config.caption_channels)
    lin("y_embedder.y_proj.fc2", C, C)
    sd["y_embedder.y_embedding"] = torch.randn(config.model_max_length, config.caption_channels, generator=g) / config.caption_channels**0.5
    sd["rope.freqs"] = 1.0 / (10000 ** (torch.arange(0, D, 2)[: (D // 2)].float() / D))
    for kind in ("spatial_blocks", "temporal_blocks"):
        for i in range(config.depth):
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
    lin("final_layer.linear", int(math.prod(config.patch_size)) * out_ch, C)
    return sd
