"""-m gpu: the AdaLN fold (csrc/adaln_fold.hip, EPI_LN_* / EPI_GATE_RES_STATS epilogues) against fp32 torch on the same
bf16-rounded inputs and against the unfused HIP path it replaces.

Replaces open_sora_transformer_3d.py:196-197 (+ attentions.py:59) and :260-261 (+ timm Mlp fc1).  Tolerances: the folded GEMM
vs the fp32 reference max|err| <= 2^-7 max|ref| (the per-op bound of tests/test_gpu_parity.py); its error may not exceed 1.25 x the
unfused HIP path's error on the same problem (+ 2^-10 max|ref| of slack); statistics partials vs torch fp64 1e-5 relative;
the statistics-emitting GEMM must store the SAME BITS as the plain gate + residual epilogue, and its partials must be the SAME BITS
the row pass computes from the stored rows."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 2.0**-7


def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from videosys_amd import ops as o

    return o


def _rows(M, C, g, offset=0.0, outliers=False):
    x = torch.randn(M, C, generator=g)
    x = x * (0.5 + torch.rand(M, 1, generator=g) * 2.0) + offset * torch.randn(M, 1, generator=g)
    if outliers:
        x[:, 7] *= 40.0
        x[:, C - 3] += 25.0
    return x.to(torch.bfloat16)


def _combine(stats, C):
    """(mu, var) per row from the partials, in fp64 on the host (the formula the epilogue uses)."""
    st = stats.double().cpu()
    mean_b, m2_b = st[..., 0], st[..., 1]
    mu = mean_b.mean(0)
    m2 = m2_b.sum(0) + 96.0 * ((mean_b - mu[None]) ** 2).sum(0)
    return mu, m2 / C


@pytest.mark.parametrize("M,C,offset", [(515, 1152, 0.0), (300, 576, 8.0), (1000, 1152, 30.0), (64, 96, 0.0), (129, 1536, 3.0)])
def test_ln_row_stats_matches_torch(ops, M, C, offset):
    g = torch.Generator().manual_seed(M + C)
    x = _rows(M, C, g, offset, outliers=True)
    st = ops.ln_stats_buffer(M, C, dev())
    st.fill_(float("nan"))
    ops.ln_row_stats(x.to(dev()), st)
    mu, var = _combine(st, C)
    xd = x.double()
    assert torch.allclose(mu, xd.mean(1), rtol=0, atol=1e-5 * xd.abs().max().item())
    assert torch.allclose(var, xd.var(1, unbiased=False), rtol=2e-5, atol=0)
    # every partial is the (mean, M2) of its own 96 columns
    blk = xd.view(M, C // 96, 96)
    assert torch.allclose(st[..., 0].double().cpu().t(), blk.mean(2), rtol=0, atol=1e-5 * xd.abs().max().item())
    assert torch.allclose(st[..., 1].double().cpu().t(), ((blk - blk.mean(2, keepdim=True)) ** 2).sum(2), rtol=5e-5, atol=1e-6)


@pytest.mark.parametrize("variant", [0, 8, 16, 24, 103, 113, 118, 119])
@pytest.mark.parametrize("M,N,K,rps", [(1100, 576, 1152, 400), (2048, 1152, 1152, 1024), (300, 1152, 4608, 300), (25856, 1152, 1152, 12928)])
def test_gemm_stats_same_bits_and_right_statistics(ops, M, N, K, rps, variant):
    """The statistics-emitting epilogue stores what the plain gate + residual epilogue stores (in place, as the model calls it),
    and its partials are the statistics of exactly those stored values.  The last shape has >= 400 tiles (the 8-wave kernel; variant
    8 / 16 = its 32x32x16 / 16x16x32 form, 24 = the two-workgroup 16x16x32 kernel, whose partials come from the LDS image), the
    others take the 128-row geometry under every id."""
    from videosys_amd import _lib

    lib = _lib.load()
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev())
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(dev())
    b = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16).to(dev())
    nsamp = -(-M // rps)
    gate = torch.randn(nsamp, N, generator=g).to(torch.bfloat16).to(dev())
    res = _rows(M, N, g, offset=4.0).to(dev())
    for use_gate in (True, False):
        r0, r1 = res.clone(), res.clone()
        kw = dict(gate=gate[0] if use_gate else None, gate_stride=N if use_gate else 0, rows_per_sample=rps if use_gate else 0)
        assert lib.vsys_tune_gemm_variant(8) == 0      # the reference bits: schedule 8, 32x32x16
        try:
            ops.gemm(x, w, b, epilogue=ops.EPI_GATE_RES, res=r0, out=r0, **kw)
            assert lib.vsys_tune_gemm_variant(variant) == 0
            st = ops.ln_stats_buffer(M, N, dev())
            st.fill_(float("nan"))
            ops.gemm_stats(x, w, b, st, res=r1, out=r1, **kw)
            r2 = res.clone()
            ops.gemm(x, w, b, epilogue=ops.EPI_GATE_RES, res=r2, out=r2, **kw)     # the plain gated epilogue of the same kernel family
        finally:
            lib.vsys_tune_gemm_variant(0)
        assert torch.equal(r0, r1), "statistics epilogue changed the stored bits"
        assert torch.equal(r0, r2), f"gate + residual epilogue of variant {variant} differs from schedule 8"
        want = ops.ln_stats_buffer(M, N, dev())
        ops.ln_row_stats(r1, want)
        # the row pass accumulates the same 48-column halves in the same order as the epilogue: the SAME partial bits
        assert torch.equal(st, want), "epilogue partials differ from the row pass on the same rows"
        mu, var = _combine(st, N)
        xd = r1.double().cpu()
        assert torch.allclose(mu, xd.mean(1), rtol=0, atol=1e-5 * xd.abs().max().item())
        assert torch.allclose(var, xd.var(1, unbiased=False), rtol=2e-5, atol=0)


def _prescale(ops, W, bias, shift, scale):
    """One-site call of vsys_adaln_prescale; shift | scale are laid out like one row of the modulation table."""
    N, K = W.shape
    mod = torch.cat([shift, scale]).contiguous()
    Wp = torch.empty_like(W)
    cs = torch.empty(N, dtype=torch.float32, device=W.device)
    cv = torch.empty(N, dtype=torch.float32, device=W.device)
    sites = torch.tensor([[W.data_ptr(), bias.data_ptr(), Wp.data_ptr(), cs.data_ptr(), cv.data_ptr(), 0, K, N, K, 0]],
                         dtype=torch.int64).to(W.device)
    ops.adaln_prescale(sites, -(-N // 4), mod)
    return Wp, cs, cv


def test_adaln_prescale_matches_torch(ops):
    g = torch.Generator().manual_seed(5)
    for N, K in ((3456, 1152), (1728, 576), (196, 96)):
        W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(dev())
        bias = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16).to(dev())
        shift = (torch.randn(K, generator=g) * 0.3).to(torch.bfloat16).to(dev())
        scale = (torch.randn(K, generator=g) * 0.3).to(torch.bfloat16).to(dev())
        Wp, cs, cv = _prescale(ops, W, bias, shift, scale)
        want = (W.float() * (1.0 + scale.float())).to(torch.bfloat16)
        assert torch.equal(Wp, want)
        assert torch.allclose(cs.double(), want.double().sum(1), rtol=0, atol=1e-4)
        assert torch.allclose(cv.double(), W.double() @ shift.double() + bias.double(), rtol=0, atol=1e-4)


def test_adaln_prescale_many_sites_one_launch(ops):
    """Sites of different N in one launch (block -> site lookup), shift / scale taken at their offsets of a shared table."""
    g = torch.Generator().manual_seed(6)
    K = 576
    mod = (torch.randn(5, 6 * K, generator=g) * 0.3).to(torch.bfloat16).to(dev())
    rows, keep, blk = [], [], 0
    for i, N in enumerate((1728, 2304, 192, 1728, 2304)):
        W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(dev())
        b = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16).to(dev())
        Wp, cs, cv = torch.empty_like(W), torch.empty(N, device=dev()), torch.empty(N, device=dev())
        so, co = i * 6 * K + (3 * K if i % 2 else 0), i * 6 * K + (4 * K if i % 2 else K)
        rows.append([W.data_ptr(), b.data_ptr(), Wp.data_ptr(), cs.data_ptr(), cv.data_ptr(), so, co, N, K, blk])
        keep.append((W, b, Wp, cs, cv, so, co))
        blk += -(-N // 4)
    ops.adaln_prescale(torch.tensor(rows, dtype=torch.int64).to(dev()), blk, mod)
    flat = mod.reshape(-1)
    for W, b, Wp, cs, cv, so, co in keep:
        want = (W.float() * (1.0 + flat[co:co + K].float())).to(torch.bfloat16)
        assert torch.equal(Wp, want)
        assert torch.allclose(cv.double(), W.double() @ flat[so:so + K].double() + b.double(), rtol=0, atol=1e-4)


@pytest.mark.parametrize("gelu", [False, True])
@pytest.mark.parametrize("M,N,K,offset", [(300, 1728, 576, 0.0), (1000, 3456, 1152, 6.0), (6144, 3456, 1152, 0.0),
                                          (5900, 4608, 1152, 20.0)])
def test_gemm_ln_vs_fp32_and_unfused(ops, M, N, K, offset, gelu):
    """Folded LayerNorm + modulate + Linear against fp32 torch on the same bf16 inputs, and against the unfused HIP pair
    (adaln_modulate, gemm).  (6144 / 5900 rows x 18 / 24 column tiles take the two-workgroups-per-CU kernel, the rest the
    128-row geometry; 5900 has a partial last tile.)"""
    from oracle import stdit3_oracle as O

    g = torch.Generator().manual_seed(M + N + int(gelu))
    x = _rows(M, K, g, offset, outliers=True).to(dev())
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(dev())
    bias = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16).to(dev())
    shift = (torch.randn(K, generator=g) * 0.3).to(torch.bfloat16).to(dev())
    scale = (torch.randn(K, generator=g) * 0.3).to(torch.bfloat16).to(dev())
    xf = x.float()
    ln = torch.nn.functional.layer_norm(xf, (K,), eps=1e-6)
    ref = (ln * (1.0 + scale.float()) + shift.float()) @ W.float().t() + bias.float()
    if gelu:
        ref = O.gelu_tanh(ref.cpu()).to(dev())
    Wp, cs, cv = _prescale(ops, W, bias, shift, scale)
    st = ops.ln_stats_buffer(M, K, dev())
    ops.ln_row_stats(x, st)
    out = ops.gemm_ln(x, Wp, cs, cv, st, gelu=gelu)
    xm = ops.adaln_modulate(x, shift, scale, M, 0)
    unf = ops.gemm(xm, W, bias, epilogue=ops.EPI_BIAS_GELU if gelu else ops.EPI_BIAS)
    scale_ref = ref.abs().max().item()
    e_fold = (out.float() - ref).abs().max().item() / scale_ref
    e_unf = (unf.float() - ref).abs().max().item() / scale_ref
    assert e_fold <= TOL, f"folded GEMM vs fp32: {e_fold:.3e}"
    assert e_fold <= 1.25 * e_unf + 2.0**-10, f"folded {e_fold:.3e} vs unfused {e_unf:.3e}"
    rms_fold = ((out.float() - ref) ** 2).mean().sqrt().item()
    rms_unf = ((unf.float() - ref) ** 2).mean().sqrt().item()
    assert rms_fold <= 1.1 * rms_unf + 1e-6 * scale_ref, f"rms folded {rms_fold:.3e} vs unfused {rms_unf:.3e}"


def test_stdit3_fold_on_off_agree_and_record(ops):
    """The model with the fold (default) and with VSYS_ADALN_FOLD=0 semantics (model.adaln_fold = False): same output within
    the bf16 noise of one block pair, both inside the oracle tolerance; the folded step replays from its launch program."""
    from oracle import stdit3_oracle as O
    from videosys_amd.stdit3 import STDiT3, STDiT3Config

    cfg = dict(depth=2, hidden_size=576, num_heads=8, caption_channels=64, model_max_length=16)
    sd = O.synth_state_dict(**cfg, seed=21)
    sd = {k: (v if k == "rope.freqs" else v.to(torch.bfloat16).float()) for k, v in sd.items()}
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 4, 5, 16, 16, generator=g).to(torch.bfloat16).float()
    y = torch.randn(2, 1, 16, 64, generator=g).to(torch.bfloat16).float()
    mask = torch.zeros(1, 16, dtype=torch.long)
    mask[:, :11] = 1
    kw = dict(mask=mask, fps=torch.tensor([24.0, 24.0]), height=torch.tensor([128.0, 128.0]), width=torch.tensor([128.0, 128.0]))
    t = torch.tensor([500.0, 500.0])
    ref = O.STDiT3Oracle(sd, cfg["depth"], cfg["hidden_size"], cfg["num_heads"]).forward(x, t, y, **kw)
    m = STDiT3(STDiT3Config(**cfg), device="cuda:0")
    m.load_state_dict(sd)
    assert m.adaln_fold
    out_f = m(x, t, y, **kw).float().cpu()
    out_f2 = m(x, t, y, **kw).float().cpu()      # replayed from the recorded program
    assert m.program_stats["recorded"] == 1 and m.program_stats["replayed"] == 1
    assert torch.equal(out_f, out_f2)
    m.adaln_fold = False
    out_u = m(x, t, y, **kw).float().cpu()
    scale = ref.abs().max().item()
    for name, o in (("folded", out_f), ("unfused", out_u)):
        err = (o - ref).abs().max().item()
        cos = torch.nn.functional.cosine_similarity(o.flatten(), ref.flatten(), dim=0).item()
        assert err <= 3e-2 * scale and cos >= 0.999, f"{name}: max|err| {err:.3e} / {scale:.3f}, cosine {cos:.6f}"
    assert (out_f - out_u).abs().max().item() <= 1.5e-2 * scale


@pytest.mark.parametrize("M,N,K,rps", [(1100, 576, 1152, 400), (2048, 1152, 1152, 1024), (300, 1152, 4608, 300), (25856, 1152, 1152, 12928)])
def test_gemm_gate_res_add_same_bits_as_separate_passes(ops, M, N, K, rps):
    """PAB broadcasts folded into the store phase of the GEMM in front of them (vsys_gemm_bf16_gate_res_add): the stored rows, the
    slab copy and the LayerNorm partials are the bits the separate launches give — gate + residual GEMM (with its slab copy), one
    `x += cached` pass per broadcast, one row-statistics pass.  Shapes as the statistics test (128-row and 8-wave geometry, ragged
    last tile)."""
    g = torch.Generator().manual_seed(M + 1)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev())
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(dev())
    b = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16).to(dev())
    gate = torch.randn(-(-M // rps), N, generator=g).to(torch.bfloat16).to(dev())
    res = _rows(M, N, g, offset=4.0).to(dev())
    a1, a2 = _rows(M, N, g, offset=-1.0).to(dev()), _rows(M, N, g, offset=0.5).to(dev())
    for use_gate, nadd, use_aux, use_stats in ((True, 2, True, True), (False, 1, False, True), (True, 1, True, False),
                                               (False, 2, False, False), (True, 0, True, True)):
        kw = dict(gate=gate[0] if use_gate else None, gate_stride=N if use_gate else 0, rows_per_sample=rps if use_gate else 0)
        adds = (a1, a2)[:nadd]
        r0, r1 = res.clone(), res.clone()
        aux0 = torch.full_like(res, float("nan")) if use_aux else None
        aux1 = torch.full_like(res, float("nan")) if use_aux else None
        ops.gemm(x, w, b, epilogue=ops.EPI_GATE_RES, res=r0, aux=aux0, out=r0, **kw)
        for a in adds:
            ops.add_rows(r0, a)
        st = None
        if use_stats:
            st = ops.ln_stats_buffer(M, N, dev())
            st.fill_(float("nan"))
        ops.gemm_gate_res_add(x, w, b, res=r1, aux=aux1, adds=adds, stats=st, out=r1, **kw)
        assert torch.equal(r0, r1), (use_gate, nadd, use_aux, use_stats)
        if use_aux:
            assert torch.equal(aux0, aux1)
        if use_stats:
            want = ops.ln_stats_buffer(M, N, dev())
            ops.ln_row_stats(r0, want)
            assert torch.equal(st, want), "partials of the folded store phase differ from the row pass on the same rows"


def test_stdit3_pab_folded_broadcasts_same_bits(ops):
    """A PAB run with the broadcasts folded into the preceding GEMMs (default) against the same run with one pass per broadcast
    (model.pab_fold_adds = False): identical outputs at every step, far fewer add_rows launches, with and without the AdaLN fold."""
    from oracle import stdit3_oracle as O
    from videosys_amd import pab
    from videosys_amd.stdit3 import STDiT3, STDiT3Config

    cfg = dict(depth=3, hidden_size=576, num_heads=8, caption_channels=64, model_max_length=16)
    sd = O.synth_state_dict(**cfg, seed=23)
    sd = {k: (v if k == "rope.freqs" else v.to(torch.bfloat16).float()) for k, v in sd.items()}
    g = torch.Generator().manual_seed(10)
    x = torch.randn(2, 4, 5, 16, 16, generator=g).to(torch.bfloat16).float()
    y = torch.randn(2, 1, 16, 64, generator=g).to(torch.bfloat16).float()
    mask = torch.zeros(1, 16, dtype=torch.long)
    mask[:, :11] = 1
    kw = dict(mask=mask, fps=torch.tensor([24.0, 24.0]), height=torch.tensor([128.0, 128.0]), width=torch.tensor([128.0, 128.0]))
    ts = [900, 860, 820, 780, 740, 700, 660, 620, 580, 540, 500, 300]

    def run(fold_adds, adaln_fold, mlp):
        extra = {}
        if mlp:
            extra = dict(mlp_broadcast=True, mlp_spatial_broadcast_config={820: {"block": [0, 1], "skip_count": 2}},
                         mlp_temporal_broadcast_config={700: {"block": [1, 2], "skip_count": 2}})
        pab.set_pab_manager(pab.PABConfig(spatial_broadcast=True, spatial_threshold=[400, 950], spatial_range=2,
                                          temporal_broadcast=True, temporal_threshold=[400, 950], temporal_range=3,
                                          cross_broadcast=True, cross_threshold=[400, 950], cross_range=4, **extra))
        pab.update_steps(len(ts))
        try:
            m = STDiT3(STDiT3Config(**cfg), device="cuda:0")
            m.load_state_dict(sd)
            m.pab_fold_adds, m.adaln_fold = fold_adds, adaln_fold
            real, n = ops.add_rows, [0]

            def counting(a, b_):
                n[0] += 1
                return real(a, b_)

            ops.add_rows = counting
            try:
                outs = [m(x, torch.tensor([float(t)] * 2), y, all_timesteps=ts, **kw).float().cpu() for t in ts]
            finally:
                ops.add_rows = real
            return outs, n[0], dict(m.program_stats)
        finally:
            pab.set_pab_manager(None)

    for adaln_fold in (True, False):
        for mlp in (False, True):
            a, na, _ = run(False, adaln_fold, mlp)
            b, nb, stats = run(True, adaln_fold, mlp)
            for t, u, v in zip(ts, a, b):
                assert torch.isfinite(u).all() and torch.equal(u, v), f"t={t} (AdaLN fold {adaln_fold}, MLP broadcast {mlp})"
            assert nb < 0.5 * na, (na, nb)
