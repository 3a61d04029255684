"""CPU (-m "not gpu"): the HOST control flow of the STDiT3 step — block sequencing, PAB plan, launch-program record / replay keys,
sequence-parallel layouts with all ranks in one process — with every kernel wrapper of videosys_amd.ops replaced by a
shape-checking stand-in that computes nothing.  TEST INFRASTRUCTURE: the stand-ins live here, the product has no CPU path
(tests/test_host_cpu.py::test_no_cpu_fallback); what this catches is Python-level breakage (a renamed variable, a wrong shape
handed to an op, a key that fails to separate two decision patterns) before a GPU minute is spent on it."""
import contextlib

import pytest
import torch

from oracle import stdit3_oracle as O


def _zeros(shape, like=None, dtype=torch.bfloat16):
    return torch.zeros(*shape, dtype=dtype)


class FakeOps:
    """Stand-ins with the signatures of videosys_amd.ops; each checks the shapes it is handed and returns a tensor of the
    shape / dtype the real kernel produces.  ``calls`` counts launches by name."""

    def __init__(self):
        import threading

        self.calls = {}
        self._tl = threading.local()   # "the statistics buffer describes the current x", per thread (= per in-process rank)

    @property
    def stats_valid(self):
        return getattr(self._tl, "v", False)

    @stats_valid.setter
    def stats_valid(self, v):
        self._tl.v = v

    def _n(self, name):
        self.calls[name] = self.calls.get(name, 0) + 1

    def gemm(self, x, w, bias=None, *, epilogue=0, gate=None, gate_stride=0, rows_per_sample=0, res=None, aux=None, out=None):
        self._n("gemm")
        assert x.dim() == 2 and w.dim() == 2 and x.shape[1] == w.shape[1], (x.shape, w.shape)
        M, N = x.shape[0], w.shape[0]
        for t in (res, aux, out):
            assert t is None or tuple(t.shape) == (M, N), (t.shape, M, N)
        assert bias is None or bias.shape == (N,)
        if epilogue == 2 and gate is not None:
            assert rows_per_sample > 0 and gate.shape == (N,)
        if epilogue == 2:
            self.stats_valid = False   # x was rewritten without statistics
        return out if out is not None else _zeros((M, N))

    LN_BLOCK = 96

    def ln_stats_buffer(self, rows, C, device):
        assert C % 96 == 0
        return torch.zeros(C // 96, rows, 2)

    def gemm_ln(self, x, wp, cs, cv, stats, *, gelu=False, eps=1e-6, out=None):
        self._n("gemm_ln")
        M, K = x.shape
        N = wp.shape[0]
        assert wp.shape == (N, K) and cs.shape == cv.shape == (N,) and stats.shape == (K // 96, M, 2)
        assert self.stats_valid, "gemm_ln reads statistics nobody wrote for the current x"
        assert out is None or tuple(out.shape) == (M, N)
        return out if out is not None else _zeros((M, N))

    def gemm_stats(self, x, w, bias, stats, *, gate=None, gate_stride=0, rows_per_sample=0, res=None, out=None):
        self._n("gemm_stats")
        M, N = x.shape[0], w.shape[0]
        assert x.shape[1] == w.shape[1] and stats.shape == (N // 96, M, 2) and tuple(out.shape) == (M, N)
        self.stats_valid = True
        return out

    def gemm_gate_res_add(self, x, w, bias, *, res, gate=None, gate_stride=0, rows_per_sample=0, aux=None, adds=(), stats=None, out=None):
        self._n("gemm_gate_res_add")
        M, N = x.shape[0], w.shape[0]
        assert x.shape[1] == w.shape[1] and tuple(out.shape) == (M, N) and tuple(res.shape) == (M, N) and len(adds) <= 2
        for t in tuple(adds) + ((aux,) if aux is not None else ()):
            assert tuple(t.shape) == (M, N)
        assert len(adds) > 0 or (aux is not None and stats is not None), "the general store phase without a reason"
        if gate is not None:
            assert rows_per_sample > 0 and gate.shape == (N,)
        self.calls["folded_adds"] = self.calls.get("folded_adds", 0) + len(adds)
        self.stats_valid = stats is not None
        return out

    def adaln_prescale(self, sites, nblocks, mod):
        self._n("adaln_prescale")
        assert sites.dtype == torch.int64 and sites.shape[1] == 10 and int(sites[-1, 9]) + -(-int(sites[-1, 7]) // 4) == nblocks
        assert int(sites[:, 6].max()) + int(sites[0, 8]) <= mod.numel()

    def ln_row_stats(self, x, stats):
        self._n("ln_row_stats")
        assert stats.shape == (x.shape[1] // 96, x.shape[0], 2)
        self.stats_valid = True
        return stats

    def linear_small(self, x, w, bias=None, act_in=0, act_out=0, out=None):
        self._n("linear_small")
        assert x.shape[1] == w.shape[1]
        return out if out is not None else _zeros((x.shape[0], w.shape[0]))

    def adaln_modulate(self, x, shift, scale, rows_per_sample, mod_stride, eps=1e-6, out=None):
        self._n("adaln_modulate")
        assert shift.shape == scale.shape == (x.shape[1],) and x.shape[0] % rows_per_sample == 0
        assert out is None or out.shape == x.shape
        return out if out is not None else torch.zeros_like(x)

    def mod_table(self, table, t_mlp, out=None):
        self._n("mod_table")
        return _zeros((table.shape[0], t_mlp.shape[0], table.shape[1]))

    def timestep_embedding(self, t, dim=256):
        self._n("timestep_embedding")
        assert t.dtype == torch.float32
        return _zeros((t.numel(), dim))

    def patch_embed(self, z, w, bias, pos, B, patch, C):
        self._n("patch_embed")
        _, _, T, H, W = z.shape
        S = -(-H // patch[1]) * -(-W // patch[2])
        assert pos.shape == (S, C)
        return _zeros((B, T, S, C))

    def patch_embed_shard(self, z, w, bias, pos, B, patch, C, s0, Sl):
        self._n("patch_embed_shard")
        return _zeros((B, z.shape[2], Sl, C))

    def final_layer(self, x, table, tvec, w, bias, B, T, Hp, Wp, H, W, patch, Cout, eps=1e-6):
        self._n("final_layer")
        assert x.shape[0] == B * T * Hp * Wp
        return torch.zeros(B, Cout, T, H, W)

    def final_layer_tokens(self, x, table, tvec, w, bias, B, T, Sl, eps=1e-6):
        self._n("final_layer_tokens")
        assert x.shape[0] == B * T * Sl
        return torch.zeros(B, T, Sl, w.shape[0])

    def unpatchify_tokens(self, tokens, P, B, T, Sl, Hp, Wp, H, W, patch, Cout):
        self._n("unpatchify_tokens")
        assert tokens.numel() == P * B * T * Sl * patch[1] * patch[2] * Cout and P * Sl >= Hp * Wp
        return torch.zeros(B, Cout, T, H, W)

    def add_rows(self, x, y):
        self._n("add_rows")
        assert x.numel() == y.numel()
        self.stats_valid = False
        return x

    def alloc_kv_buffers(self, batch, heads, kv_len, device):
        kv_pad = (kv_len + 63) // 64 * 64
        return _zeros((batch, heads, kv_pad, 72)), _zeros((batch, heads, 96, kv_pad))

    def kv_pad_len(self, kv_len):
        return (kv_len + 63) // 64 * 64

    def attn_prep_kv(self, k, v, k_norm_w, kp, vt, batch, heads, kv_len, eps=1e-6):
        self._n("attn_prep_kv")
        assert k.shape[0] == v.shape[0] == batch * kv_len and kp.shape[0] == batch and kp.shape[2] >= kv_len

    def rms_key_bound(self, q_norm_w, k_norm_w, head_dim=72):
        assert q_norm_w.shape == k_norm_w.shape == (head_dim,)
        return 1.5

    def flash_attn(self, q, q_norm_w, kp, vt, out, batch, heads, q_len, kv_len, eps=1e-6, k_norm_bound=None, keys_exact=False):
        self._n("flash_attn")
        assert k_norm_bound is None or q_norm_w is not None
        # the padding promise is only ever given for the hoisted text K / V (no q norm), prepared for exactly kv_len keys
        assert not keys_exact or (q_norm_w is None and kp.shape[2] == -(-kv_len // 64) * 64)
        assert q.shape[0] == out.shape[0] == batch * q_len and kp.shape[0] == batch and kp.shape[2] >= kv_len, (q.shape, batch, q_len, kp.shape)
        return out

    def attn_temporal(self, qkv, C, q_norm_w, k_norm_w, cos, sin, out, B, T, S, heads, eps=1e-6):
        self._n("attn_temporal")
        assert qkv.shape == (B * T * S, 3 * C) and out.shape == (B * T * S, C) and cos.shape == (T, 72)
        return out

    def cfg_euler_step(self, z, out, g, dt):
        self._n("cfg_euler_step")
        return z


@contextlib.contextmanager
def fake_ops():
    from videosys_amd import ops

    f = FakeOps()
    names = [n for n in dir(f) if not n.startswith("_") and n not in ("calls", "stats_valid")]
    saved = {n: getattr(ops, n) for n in names}
    for n in names:
        setattr(ops, n, getattr(f, n))
    try:
        yield f
    finally:
        for n, v in saved.items():
            setattr(ops, n, v)


CFG = dict(depth=2, hidden_size=576, num_heads=8, caption_channels=64, model_max_length=16)


def _model():
    from videosys_amd.stdit3 import STDiT3, STDiT3Config

    m = STDiT3(STDiT3Config(**CFG), device="cpu")
    m.load_state_dict(O.synth_state_dict(**CFG, seed=3))
    return m


def _inputs(T=5, HW=8):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 4, T, HW, HW, generator=g)
    y = torch.randn(2, 1, 16, 64, generator=g)
    mask = torch.zeros(1, 16, dtype=torch.long)
    mask[:, :11] = 1
    kw = dict(mask=mask, fps=torch.tensor([24.0, 24.0]), height=torch.tensor([64.0, 64.0]), width=torch.tensor([64.0, 64.0]))
    return x, y, kw


def test_step_launch_counts_and_program_replay():
    x, y, kw = _inputs()
    with fake_ops() as f:
        m = _model()
        t = torch.tensor([500.0, 500.0])
        out = m(x, t, y, **kw)
        assert out.shape == (2, 8, 5, 8, 8)
        # per block (AdaLN fold, the default when the batch shares one timestep): no AdaLN pass; qkv and fc1 are folded GEMMs,
        # cross-proj and fc2 emit the statistics the next folded GEMM reads, attention proj and cross q are plain; one pre-scale
        # launch per step; one stand-alone statistics pass in front of block 0 (x comes from the patch embedding there)
        nblk = 2 * CFG["depth"]
        assert "adaln_modulate" not in f.calls and f.calls["attn_temporal"] == CFG["depth"]
        assert f.calls["gemm_ln"] == 2 * nblk and f.calls["gemm_stats"] == 2 * nblk and f.calls["adaln_prescale"] == 1
        assert f.calls["ln_row_stats"] == 1
        assert f.calls["flash_attn"] == nblk + CFG["depth"]
        gemms_text = f.calls["gemm"] - 2 * nblk     # the once-per-prompt kv_linear projections (GEMM or small linear by shape)
        assert gemms_text in (0, nblk)
        before = dict(f.calls)
        assert m.program_stats == dict(recorded=1, replayed=0, eager=0)
        m(x, t, y, **kw)                            # same key: replayed — no Python-level launch at all
        assert m.program_stats["replayed"] == 1 and f.calls == before
        m(x[:, :, :3], t, y, **kw)                  # another geometry: its own program
        assert m.program_stats["recorded"] == 2
        m.use_programs = False
        m(x, t, y, **kw)
        assert m.program_stats["eager"] == 1
        m.reset_text_cache()
        assert not m._programs
        # two samples at different timesteps have different modulations: the fold is off for that call (own program key),
        # and VSYS_ADALN_FOLD=0 (model.adaln_fold) keeps the separate LayerNorm-modulate pass everywhere
        m.use_programs = True
        g0 = dict(f.calls)
        m(x, torch.tensor([500.0, 400.0]), y, **kw)
        assert f.calls["adaln_modulate"] == 2 * nblk and f.calls["gemm_ln"] == g0["gemm_ln"]
        m.adaln_fold = False
        g1, rep = dict(f.calls), m.program_stats["replayed"]
        m(x, t, y, **kw)                            # the unfused step at this geometry was recorded by the call above: replayed
        assert f.calls == g1 and m.program_stats["replayed"] == rep + 1
        m.use_programs = False
        m(x, t, y, **kw)
        assert f.calls["adaln_modulate"] == 4 * nblk and f.calls["gemm_ln"] == g0["gemm_ln"]


def test_pab_patterns_get_their_own_programs_and_mlp_steps_run_eagerly():
    from videosys_amd import pab

    x, y, kw = _inputs()
    sched = [1000, 900, 800, 700, 600, 500, 400, 300]
    rule = {800: {"block": [0, 1], "skip_count": 2}}
    pab.set_pab_manager(pab.PABConfig(spatial_broadcast=True, spatial_threshold=[450, 930], spatial_range=2,
                                      temporal_broadcast=True, temporal_threshold=[450, 930], temporal_range=4,
                                      cross_broadcast=True, cross_threshold=[450, 930], cross_range=6,
                                      mlp_broadcast=True, mlp_spatial_broadcast_config=rule, mlp_temporal_broadcast_config=rule))
    pab.update_steps(len(sched))
    try:
        with fake_ops() as f:
            m = _model()
            with pytest.raises(ValueError):
                m(x, torch.tensor([900.0, 900.0]), y, **kw)       # the MLP broadcast needs the schedule
            m.reset_pab_state()
            patterns = []
            for rep in range(2):
                m.reset_pab_state()
                for t in sched:
                    adds = f.calls.get("add_rows", 0)
                    m(x, torch.tensor([float(t)] * 2), y, all_timesteps=sched, **kw)
                    if rep == 0:
                        patterns.append(f.calls.get("add_rows", 0) - adds)
            # steps 800 / 700 / 600 open, replay and close the MLP windows: they run eagerly, every other pattern is recorded once
            # in the first pass and replayed in the second
            assert m.program_stats["eager"] == 2 * 3, m.program_stats
            assert m.program_stats["replayed"] >= len(sched) - 3, m.program_stats
            # broadcast steps add slabs (+ the fps embedding once per eager / recorded step); a step whose decision pattern was
            # seen before is replayed and issues nothing from Python
            assert max(patterns) > 1 and patterns[0] == 1 and min(patterns) == 0
            assert pab.PAB_MANAGER.config.mlp_spatial_outputs == {} and pab.PAB_MANAGER.config.mlp_temporal_outputs == {}
            assert len(m._ws.get("mlp_slab_pool", [])) >= 1       # closed windows handed their slabs back
    finally:
        pab.set_pab_manager(None)


@pytest.mark.parametrize("scatter,order", [("flat", "activations"), ("sample", "activations"), ("flat", "qkv")])
def test_sequence_parallel_host_flow_four_ranks_in_process(scatter, order):
    """All four ranks of a DSP group as threads (tools/local_group.py), kernels faked: every rank must walk the same collective
    sequence (a mismatch deadlocks the barrier -> timeout error) with consistent shard shapes, and replay it from the program."""
    from types import SimpleNamespace

    from test_host_cpu import torch_copy_executor
    from tools.local_group import LocalWorld

    x, y, kw = _inputs(T=5, HW=12)        # S = 36 tokens over 4 ranks; T = 5 frames x 2 samples
    P = 4

    def rank_fn(r, group):
        m = _model()
        pm = SimpleNamespace(sp_size=P, cp_size=1, dp_size=1, dp_rank=0, sp_rank=r, cp_rank=0, sp_group=group, cp_group=None)
        m.enable_parallel(parallel_mgr=pm, copy_executor=torch_copy_executor, overlap=False)
        m._scatter, m._switch = scatter, order
        t = torch.tensor([500.0, 500.0])
        out = m(x, t, y, **kw)
        out2 = m(x, t, y, **kw)
        assert out.shape == out2.shape == (2, 8, 5, 12, 12)
        return dict(m.program_stats)

    with fake_ops():
        stats = LocalWorld(P, timeout=60).run(rank_fn)
    assert all(s == dict(recorded=1, replayed=1, eager=0) for s in stats), stats


def test_x_mask_runs_eagerly_with_one_modulation_row_per_frame():
    """Conditioning mask (open_sora_transformer_3d.py:181-184,578-582): the modulation table gets one row per (sample, frame),
    every modulated / gated launch addresses it with one frame's rows per modulation row, the final layer runs twice (the
    t and the t = 0 branch, T2IFinalLayer :82-85), and no launch program is recorded for such a step."""
    from types import SimpleNamespace

    from test_host_cpu import torch_copy_executor
    from tools.local_group import LocalWorld

    x, y, kw = _inputs(T=5, HW=12)
    xm = torch.tensor([[1, 0, 1, 1, 0], [0, 1, 1, 0, 1]], dtype=torch.bool)
    t = torch.tensor([500.0, 500.0])
    with fake_ops() as f:
        m = _model()
        seen = []
        real_adaln, real_gemm, real_mod = f.adaln_modulate, f.gemm, f.mod_table
        from videosys_amd import ops

        def adaln(x_, shift, scale, rows_per_sample, mod_stride, **k):
            seen.append(("adaln", rows_per_sample, mod_stride))
            return real_adaln(x_, shift, scale, rows_per_sample, mod_stride, **k)

        def gemm(x_, w, bias=None, **k):
            if k.get("gate") is not None:
                seen.append(("gate", k["rows_per_sample"], k["gate_stride"]))
            return real_gemm(x_, w, bias, **k)

        def mod_table(table, t_mlp, out=None):
            seen.append(("mod", t_mlp.shape[0]))
            return real_mod(table, t_mlp, out)

        ops.adaln_modulate, ops.gemm, ops.mod_table = adaln, gemm, mod_table
        out = m(x, t, y, x_mask=xm, **kw)
        assert out.shape == (2, 8, 5, 12, 12)
        assert m.program_stats == dict(recorded=0, replayed=0, eager=1)
        C, S = CFG["hidden_size"], 36
        assert ("mod", 2 * 5) in seen
        blk = [e for e in seen if e[0] in ("adaln", "gate") and e[2] == 6 * C]
        assert len(blk) == 4 * 2 * CFG["depth"] and all(e[1] == S for e in blk), blk
        assert ("adaln", 5 * S, 2 * C) in seen           # the final layer's first modulation, per sample
        assert f.calls["final_layer"] == 2
        with pytest.raises(ValueError):
            m(x, t, y, x_mask=xm[:, :4], **kw)
        seen.clear()
        m(x, t, y, **kw)                                  # without a mask: per-sample rows again, recorded as a program
        assert ("mod", 2) in seen and all(e[1] == 5 * S for e in seen if e[0] in ("adaln", "gate"))
        assert m.program_stats["recorded"] == 1

    P = 4

    def rank_fn(r, group):
        mm = _model()
        pm = SimpleNamespace(sp_size=P, cp_size=1, dp_size=1, dp_rank=0, sp_rank=r, cp_rank=0, sp_group=group, cp_group=None)
        mm.enable_parallel(parallel_mgr=pm, copy_executor=torch_copy_executor, overlap=False)
        o = mm(x, t, y, x_mask=xm, **kw)
        assert o.shape == (2, 8, 5, 12, 12)
        return dict(mm.program_stats)

    with fake_ops() as f:
        stats = LocalWorld(P, timeout=60).run(rank_fn)
        assert f.calls["final_layer_tokens"] == 2 * P
    assert all(s == dict(recorded=0, replayed=0, eager=1) for s in stats), stats


def test_pab_slab_elision_never_reads_a_stale_slab():
    """STDiT3._pab_plan keeps a computed attention output only when the block's next call will broadcast it.  With elision on,
    every broadcast must read a slab written at exactly the step it is read from when every computed output is kept (elision
    off = the reference's behaviour): the (step, step-of-last-write) sequence of all broadcast adds must be identical, while
    the number of slab writes drops."""
    from videosys_amd import pab
    from videosys_amd.rflow import RFLOW

    x, y, kw = _inputs()
    sched = RFLOW(num_sampling_steps=30, cfg_scale=7.0, use_timestep_transform=True)
    geom = dict(height=torch.tensor([512.0]), width=torch.tensor([512.0]), num_frames=torch.tensor([64.0]))
    ts = [int(t.to(torch.bfloat16)[0]) for t in sched.prepare_timesteps(1, geom)]

    def run(elide):
        pab.set_pab_manager(pab.PABConfig(spatial_broadcast=True, spatial_threshold=[450, 930], spatial_range=2,
                                          temporal_broadcast=True, temporal_threshold=[450, 930], temporal_range=4,
                                          cross_broadcast=True, cross_threshold=[450, 930], cross_range=6))
        pab.update_steps(len(ts))
        try:
            with fake_ops() as f:
                m = _model()
                m.use_programs = False          # every launch goes through the stand-ins
                m.pab_elide_unused = elide
                state = dict(step=-1, written={}, reads=[], writes=0)
                real_gemm, real_add, real_gra = f.gemm, f.add_rows, f.gemm_gate_res_add

                def gemm(x_, w, bias=None, **k):
                    if k.get("aux") is not None:
                        state["written"][k["aux"].data_ptr()] = state["step"]
                        state["writes"] += 1
                    return real_gemm(x_, w, bias, **k)

                def add_rows(x_, y_):
                    if y_.shape == x_.shape and y_.dim() == 2 and y_.shape[0] > 2:      # a slab (not the fps embedding)
                        state["reads"].append((state["step"], state["written"].get(y_.data_ptr())))
                    return real_add(x_, y_)

                def gemm_gate_res_add(x_, w, bias, **k):   # a GEMM whose store phase performs the broadcasts that follow it
                    if k.get("aux") is not None:
                        state["written"][k["aux"].data_ptr()] = state["step"]
                        state["writes"] += 1
                    for y_ in k.get("adds", ()):
                        state["reads"].append((state["step"], state["written"].get(y_.data_ptr())))
                    state["folded"] = state.get("folded", 0) + len(k.get("adds", ()))
                    return real_gra(x_, w, bias, **k)

                from videosys_amd import ops

                ops.gemm, ops.add_rows, ops.gemm_gate_res_add = gemm, add_rows, gemm_gate_res_add
                for rep in range(2):            # two videos back to back: the counters wrap
                    m.reset_pab_state()
                    for i, t in enumerate(ts):
                        state["step"] = rep * len(ts) + i
                        m(x, torch.tensor([float(t)] * 2), y, all_timesteps=ts, **kw)
                return state
        finally:
            pab.set_pab_manager(None)

    keep_all, elided = run(False), run(True)
    assert keep_all["reads"] and all(w is not None for _, w in keep_all["reads"])
    assert elided["reads"] == keep_all["reads"], "a broadcast read a slab written at another step than the reference's"
    assert elided["writes"] < keep_all["writes"], (elided["writes"], keep_all["writes"])
    # broadcasts ride in the store phase of the GEMM in front of them: only those at the head of block 0 remain passes of their own
    assert elided["folded"] > 0.8 * len(elided["reads"]), (elided["folded"], len(elided["reads"]))
    # an off-schedule call (timestep not on the schedule) keeps everything
    pab.set_pab_manager(pab.PABConfig(spatial_broadcast=True, spatial_threshold=[450, 930], spatial_range=2))
    pab.update_steps(30)
    try:
        with fake_ops():
            m = _model()
            plan = m._pab_plan(555, ts, CFG["depth"])
            assert all(d[5] and d[6] for d in plan)
    finally:
        pab.set_pab_manager(None)


def test_rflow_mask_bookkeeping_matches_the_reference():
    """RFLOW.sample(mask=...) with the model and the Euler kernel faked (zero velocity): the x_mask of every step equals what
    the reference's sampler handed ITS model on the same schedule (fixture minted by oracle/make_golden_xmask.py), a held frame
    comes back untouched, a frame with edit ratio 0.6 is noised exactly once — at the first step with t <= 600, to that step's
    level — and an all-True step drops the mask so the plain (program) path runs."""
    from types import SimpleNamespace

    from conftest import load_golden
    from videosys_amd.rflow import RFLOW

    s = load_golden("stdit3_xmask_small.pt")["sample"]
    seen = []

    class Model:
        device = torch.device("cpu")
        x_embedder = SimpleNamespace(proj=SimpleNamespace(weight=torch.zeros(1, dtype=torch.float32)))

        def __call__(self, z_in, t, **k):
            seen.append(k.get("x_mask"))
            return torch.zeros(z_in.shape[0], 8, *z_in.shape[2:])

    margs = dict(y=s["y"], mask=s["mask"], height=s["height"], width=s["width"], num_frames=s["num_frames"], fps=s["fps"])
    sched = RFLOW(num_sampling_steps=s["steps"], cfg_scale=s["cfg_scale"], use_timestep_transform=True)
    g = torch.Generator().manual_seed(3)
    noises = [torch.randn(s["z0"].shape, generator=g) for _ in range(s["steps"])]
    it = iter(noises)
    with fake_ops():
        z = sched.sample(Model(), s["z0"], margs, s["y_null"], mask=s["cond_mask"], noise_fn=lambda shape: next(it))
        assert [m.tolist() for m in seen] == [m.tolist() for m in s["x_masks"]]
        ts = sched.prepare_timesteps(1, margs)
        join = next(i for i, t in enumerate(ts) if float(t[0]) <= 600.0)
        keep = 1 - float(ts[join][0]) / 1000
        assert torch.equal(z[:, :, 0], s["z0"][:, :, 0]) and torch.equal(z[:, :, 2:], s["z0"][:, :, 2:])
        torch.testing.assert_close(z[:, :, 1], keep * s["z0"][:, :, 1] + (1 - keep) * noises[join][:, :, 1], rtol=1e-6, atol=1e-6)
        seen.clear()
        it = iter(noises)
        sched.sample(Model(), s["z0"], margs, s["y_null"], mask=torch.tensor([[0.99, 1.0, 1.0, 1.0, 1.0]]), noise_fn=lambda shape: next(it))
        assert seen[0] is not None and seen[0].tolist() == [[False, True, True, True, True]] * 2 and all(m is None for m in seen[1:])
        with pytest.raises(ValueError):
            sched.sample(Model(), s["z0"], margs, s["y_null"], mask=torch.ones(1, 4))


def test_skip_y_embedder_takes_projected_text_and_lengths():
    """config.skip_y_embedder (open_sora_transformer_3d.py:585-590): y arrives projected and packed, mask carries the per-sample
    token counts (a list, as the training data loader hands them over): the caption MLP is not run, the K/V cache is built from y."""
    from videosys_amd.stdit3 import STDiT3, STDiT3Config

    x, _, kw = _inputs()
    kw = dict(kw)
    with fake_ops() as f:
        m = STDiT3(STDiT3Config(skip_y_embedder=True, **CFG), device="cpu")
        m.load_state_dict(O.synth_state_dict(**CFG, seed=3))
        y = torch.randn(1, 2 * 11, CFG["hidden_size"])
        kw["mask"] = [11, 11]
        before = f.calls.get("linear_small", 0)
        out = m(x, torch.tensor([500.0, 500.0]), y, **kw)
        assert out.shape == (2, 8, 5, 8, 8)
        used = f.calls.get("linear_small", 0) - before
        m2 = _model()
        before = f.calls.get("linear_small", 0)
        _, y_raw, kw_raw = _inputs()
        m2(x, torch.tensor([500.0, 500.0]), y_raw, **kw_raw)
        assert f.calls.get("linear_small", 0) - before >= used + 2      # the two caption-projection linears are the difference
        m(x, torch.tensor([500.0, 500.0]), y, **kw)                      # same y, same lengths: the text cache holds, program replays
        assert m.program_stats["replayed"] == 1
        with pytest.raises(ValueError):
            m(x, torch.tensor([500.0, 500.0]), y, **dict(kw, mask=[10, 12]))


def test_generate_conditioning_and_loop_host_flow():
    """OpenSoraPipeline.generate with references, a mask strategy and loop = 2 on CPU — kernels and the VAE faked — checks the
    host sequence of pipeline_open_sora.py:528-535,607-645: references are collected (latents pass through, pixels are encoded),
    the strategy is pasted into the start noise and held frames come back untouched, the second loop is conditioned on the
    re-encoded tail of the first clip and the clips are joined in time with the 17-frame overlap removed."""
    from videosys_amd import OpenSoraConfig
    from videosys_amd.pipeline_open_sora import OpenSoraPipeline

    class FakeVAE:
        has_encoder = True

        def __init__(self):
            self.encoded, self.decoded = [], []

        def encode(self, v):                       # [B, 3, T, H, W] -> [B, 4, ceil-ish(T / 17 * 5), H / 8, W / 8]
            self.encoded.append(tuple(v.shape))
            T = v.shape[2]
            tz = (T // 17) * 5 + (-(-(T % 17) // 4) if T % 17 else 0)
            return torch.full((v.shape[0], 4, tz, v.shape[3] // 8, v.shape[4] // 8), 0.5)

        def __call__(self, z, num_frames):
            self.decoded.append(tuple(z.shape))
            return torch.zeros(z.shape[0], 3, num_frames, z.shape[3] * 8, z.shape[4] * 8)

    with fake_ops():
        vae = FakeVAE()
        pipe = OpenSoraPipeline(OpenSoraConfig(transformer="synthetic:3", num_sampling_steps=3, transformer_config=CFG), device="cpu",
                                text_encoder=None, vae_decoder=vae)
        emb = torch.randn(1, 1, 16, 64)
        pm = torch.ones(1, 16, dtype=torch.long)
        kw = dict(prompt_embeds=emb, prompt_mask=pm, height=64, width=64, num_frames=34, seed=3, verbose=False)
        lat_ref = torch.full((4, 2, 8, 8), 2.0)
        out = pipe.generate(output_type="latent", refs=[lat_ref], ms="0,0,0,0,2,0", align=None, **kw).video
        assert tuple(out.shape) == (1, 4, 10, 8, 8) and vae.encoded == []
        assert torch.equal(out[0, :, :2].cpu(), lat_ref)                     # held (mask 0): pasted in, never touched by the sampler
        img = torch.zeros(3, 1, 64, 64)
        out = pipe.generate(output_type="latent", refs=[img], ms="0", **kw).video
        assert vae.encoded == [(1, 3, 1, 64, 64)] and float(out[0, :, 0].mean()) == 0.5
        vae.encoded.clear()
        video = pipe.generate(loop=2, condition_frame_length=5, **kw).video
        assert vae.encoded == [(1, 3, 34, 64, 64)] and len(vae.decoded) == 2
        assert video.dtype == torch.uint8 and tuple(video.shape) == (1, 34 + 34 - 17, 64, 64, 3)
        with pytest.raises(RuntimeError):
            pipe.generate(loop=2, output_type="latent", **kw)
        vae.has_encoder = False
        with pytest.raises(RuntimeError):
            pipe.generate(refs=[img], ms="0", **kw)


def test_pipeline_adopts_the_reference_transformer_module():
    """OpenSoraPipeline(config, transformer=<the reference's own STDiT3 module>): geometry from its config, weights from its
    state_dict() (INTEGRATION.md "Handing over components") — the live reference class, imported through oracle/ref_loader.py."""
    from oracle import ref_loader

    if not ref_loader.reference_available():
        pytest.skip("reference tree not present on this box")
    from videosys_amd import OpenSoraConfig, OpenSoraPipeline
    from videosys_amd.stdit3 import STDiT3

    cfg = dict(CFG)
    sd = {k: v.to(torch.bfloat16).float() for k, v in O.synth_state_dict(**cfg, seed=3).items()}
    ref = ref_loader.build_reference_stdit3(cfg, sd)
    with fake_ops():
        pipe = OpenSoraPipeline(OpenSoraConfig(num_sampling_steps=2), transformer=ref, device="cpu")
        m = pipe.transformer
        assert isinstance(m, STDiT3) and m.config.depth == cfg["depth"] and m.config.hidden_size == cfg["hidden_size"]
        assert m.config.caption_channels == cfg["caption_channels"] and m.config.model_max_length == cfg["model_max_length"]
        for k in ("spatial_blocks.1.attn.qkv.weight", "temporal_blocks.0.mlp.fc2.bias", "final_layer.linear.weight", "y_embedder.y_embedding"):
            assert torch.equal(m.w[k].float(), ref.state_dict()[k].to(torch.bfloat16).float()), k
        assert torch.equal(m.rope_freqs, ref.state_dict()["rope.freqs"].float())
        own = STDiT3.from_pretrained("synthetic:3", device="cpu", **cfg)                # this build's own object: used as it is
        assert OpenSoraPipeline(OpenSoraConfig(), transformer=own, device="cpu").transformer is own


def test_pab_broadcast_of_an_elided_slab_recomputes():
    """A caller that leaves the schedule it announced (the same timestep twice: the reference's counters then ask for a broadcast at a
    call whose predecessor elided its slab because the SCHEDULE said nobody would read it) gets a recompute, never a stale or missing
    slab: _pab_plan downgrades a broadcast whose slab is not valid."""
    from videosys_amd import pab

    x, y, kw = _inputs()
    ts = [900, 800, 700, 600, 500, 400]
    pab.set_pab_manager(pab.PABConfig(spatial_broadcast=True, spatial_threshold=[450, 930], spatial_range=2,
                                      temporal_broadcast=True, temporal_threshold=[450, 930], temporal_range=2,
                                      cross_broadcast=True, cross_threshold=[450, 930], cross_range=2))
    pab.update_steps(len(ts))
    try:
        with fake_ops() as f:
            m = _model()
            m.use_programs = False
            seen = []
            real_add = f.add_rows
            from videosys_amd import ops

            def add_rows(x_, y_):
                seen.append(y_ is not None)
                assert y_ is not None, "a broadcast read a slab that was never written"
                return real_add(x_, y_)

            ops.add_rows = add_rows
            # 900 computes (and keeps: 800 will broadcast), 800 broadcasts; 700 computes and ELIDES nothing it should keep ... then the
            # caller repeats 700 and jumps to 400: every decision must still find its slab or recompute
            for t in (900, 800, 700, 700, 400, 400, 600):
                m(x, torch.tensor([float(t)] * 2), y, all_timesteps=ts, **kw)
            plan_states = [(st.attn_valid, st.cross_valid) for st in m.states]
            assert all(isinstance(a, bool) and isinstance(c, bool) for a, c in plan_states)
    finally:
        pab.set_pab_manager(None)


def test_pab_step_that_fails_leaves_no_slab_marked_valid():
    """_pab_plan marks the slabs a step is about to write while planning; when the step raises half way (launch error, OOM) no slab may
    stay marked valid, so the next call recomputes instead of broadcasting a slab that was never written."""
    from videosys_amd import ops, pab

    x, y, kw = _inputs()
    ts = [900, 800, 700, 600]
    pab.set_pab_manager(pab.PABConfig(spatial_broadcast=True, spatial_threshold=[450, 930], spatial_range=2,
                                      temporal_broadcast=True, temporal_threshold=[450, 930], temporal_range=2,
                                      cross_broadcast=True, cross_threshold=[450, 930], cross_range=2))
    pab.update_steps(len(ts))
    try:
        with fake_ops():
            m = _model()
            m.use_programs = False
            real = ops.final_layer

            def boom(*a, **k):
                raise RuntimeError("launch failed")

            ops.final_layer = boom
            with pytest.raises(RuntimeError, match="launch failed"):
                m(x, torch.tensor([900.0] * 2), y, all_timesteps=ts, **kw)
            assert not any(st.attn_valid or st.cross_valid for st in m.states)
            ops.final_layer = real
            real_add = ops.add_rows

            def add_rows(a, b):
                assert b is not None, "a broadcast read a slab that was never written"
                return real_add(a, b)

            ops.add_rows = add_rows
            m(x, torch.tensor([800.0] * 2), y, all_timesteps=ts, **kw)     # the counters ask for broadcasts: all downgraded to recomputes
    finally:
        pab.set_pab_manager(None)


def test_sequence_parallel_with_pab_folds_broadcasts_on_every_rank():
    """PAB under DSP, four ranks in one process with faked kernels: every rank makes the same PAB decisions (they depend on the
    timestep only), so every rank folds the same broadcasts into the GEMM in front of them, walks the same collective sequence (a
    mismatch would deadlock the barrier) and hands slabs of its OWN shard shape to the folded store phase."""
    from types import SimpleNamespace

    from test_host_cpu import torch_copy_executor
    from tools.local_group import LocalWorld
    from videosys_amd import pab

    x, y, kw = _inputs(T=5, HW=12)
    P = 4
    ts = [900, 800, 700, 600, 500]
    pab.set_pab_manager(pab.PABConfig(spatial_broadcast=True, spatial_threshold=[450, 930], spatial_range=2,
                                      temporal_broadcast=True, temporal_threshold=[450, 930], temporal_range=3,
                                      cross_broadcast=True, cross_threshold=[450, 930], cross_range=4))
    pab.update_steps(len(ts))

    def rank_fn(r, group):
        m = _model()
        pm = SimpleNamespace(sp_size=P, cp_size=1, dp_size=1, dp_rank=0, sp_rank=r, cp_rank=0, sp_group=group, cp_group=None)
        m.enable_parallel(parallel_mgr=pm, copy_executor=torch_copy_executor, overlap=False)
        for t in ts:
            out = m(x, torch.tensor([float(t)] * 2), y, all_timesteps=ts, **kw)
            assert out.shape == (2, 8, 5, 12, 12)
        return dict(m.program_stats)

    try:
        with fake_ops() as f:
            stats = LocalWorld(P, timeout=120).run(rank_fn)
            folded, passes = f.calls.get("folded_adds", 0), f.calls.get("add_rows", 0)
    finally:
        pab.set_pab_manager(None)
    assert all(s["eager"] == 0 for s in stats), stats
    assert folded > 0 and folded % P == 0, (folded, passes)      # the same number on every rank
