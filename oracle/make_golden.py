"""TEST INFRASTRUCTURE ONLY — mints tests/golden/*.pt from the REAL reference.

Run in the build container (needs /root/reference):

    python oracle/make_golden.py

Imports the reference's own Python through ``oracle/ref_loader.py`` (stubs for
the un-vendored deps), runs it on CPU in fp32 on seeded synthetic inputs, and
stores inputs + reference outputs.  Weights of the model-level fixtures are NOT
stored (size): they are regenerated from ``synth_state_dict(seed=...)`` and a
checksum in the fixture guards against generator drift.  All tensors that the
HIP path consumes in bf16 are rounded to bf16-representable values *before* the
reference runs, so "reference fp32 on bf16-rounded inputs" is what is pinned.
"""
from __future__ import annotations

import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_loader  # noqa: E402
from oracle import stdit3_oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def bf16r(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def pack(obj):
    """Store bf16-exact fp32 tensors as bf16 (lossless, halves the fixture); tests call .float()."""
    if isinstance(obj, torch.Tensor):
        if obj.dtype == torch.float32 and torch.equal(bf16r(obj), obj):
            return obj.to(torch.bfloat16)
        return obj
    if isinstance(obj, dict):
        return {k: pack(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(pack(v) for v in obj)
    return obj


def sd_checksum(sd):
    return float(sum(v.double().abs().sum() for v in sd.values()))


def round_sd(sd):
    return {k: (v if k == "rope.freqs" else bf16r(v)) for k, v in sd.items()}


# model-level fixture geometry: C = 8 heads x 72 (head_dim 72 as in STDiT3-XL/2), K dims multiples of 64
SMALL_CFG = dict(depth=2, hidden_size=576, num_heads=8, caption_channels=64, model_max_length=20)
SMALL_SEED = 4321


def small_inputs(seed=7, T=5, HW=16, L=20, L_valid=13):
    g = torch.Generator().manual_seed(seed)
    x = bf16r(torch.randn(2, 4, T, HW, HW, generator=g))
    y = bf16r(torch.randn(2, 1, L, SMALL_CFG["caption_channels"], generator=g))
    mask = torch.zeros(1, L, dtype=torch.long)
    mask[:, :L_valid] = 1
    timestep = torch.tensor([712.0, 712.0])  # bf16-exact
    fps = torch.tensor([24.0, 24.0])
    height = torch.tensor([float(HW * 8)] * 2)
    width = torch.tensor([float(HW * 8)] * 2)
    return dict(x=x, y=y, mask=mask, timestep=timestep, fps=fps, height=height, width=width)


def make_stdit3_fwd():
    sd = round_sd(O.synth_state_dict(**SMALL_CFG, seed=SMALL_SEED))
    ref = ref_loader.build_reference_stdit3(SMALL_CFG, sd)
    inp = small_inputs()
    hidden = []
    hooks = []
    for blk in ref.temporal_blocks:
        hooks.append(blk.register_forward_hook(lambda m, i, o: hidden.append(o.detach().clone())))
    with torch.no_grad():
        out = ref(inp["x"], inp["timestep"], inp["y"], mask=inp["mask"], fps=inp["fps"],
                  height=inp["height"], width=inp["width"])
    for h in hooks:
        h.remove()
    fix = dict(cfg=SMALL_CFG, seed=SMALL_SEED, sd_checksum=sd_checksum(sd), inputs=inp, out=out,
               hidden_rows=[h[:, ::97, :].clone() for h in hidden], hidden_stride=97)
    torch.save(pack(fix), os.path.join(OUT, "stdit3_fwd_small.pt"))
    print("stdit3_fwd_small", out.shape, float(out.abs().mean()))

    # PAB fixture: 8 denoise calls of the same small model with attention-only PAB (mlp_broadcast off)
    mods = ref_loader.load_reference_modules()
    pab = mods["pab_mgr"]
    cfg = pab.PABConfig(
        cross_broadcast=True, cross_threshold=[100, 930], cross_range=3,
        spatial_broadcast=True, spatial_threshold=[100, 930], spatial_range=2,
        temporal_broadcast=True, temporal_threshold=[100, 930], temporal_range=2,
    )
    pab.set_pab_manager(cfg)
    pab.update_steps(6)
    ts = [980.0, 900.0, 800.0, 640.0, 400.0, 96.0]  # bf16-exact ints; first/last outside the window
    outs = []
    with torch.no_grad():
        for t in ts:
            tt = torch.tensor([t, t])
            outs.append(ref(inp["x"], tt, inp["y"], mask=inp["mask"], fps=inp["fps"], height=inp["height"],
                            width=inp["width"]))
    pab.PAB_MANAGER = None
    torch.save(pack(dict(cfg=SMALL_CFG, seed=SMALL_SEED, inputs=inp, timesteps=ts, steps=6,
                         pab=dict(spatial=(100, 930, 2), temporal=(100, 930, 2), cross=(100, 930, 3)),
                         outs=torch.stack(outs))), os.path.join(OUT, "stdit3_pab_small.pt"))
    print("stdit3_pab_small", len(outs))


def make_rflow():
    """RFLOW.sample (reference, scheduling_rflow_open_sora.py:188-257) with the small reference STDiT3, 4 steps."""
    mods = ref_loader.load_reference_modules()
    rf = mods["rflow"]
    import torch.distributed as dist

    sd = round_sd(O.synth_state_dict(**SMALL_CFG, seed=SMALL_SEED))
    ref = ref_loader.build_reference_stdit3(SMALL_CFG, sd)
    inp = small_inputs(seed=11)
    z = inp["x"][:1].clone()
    y = inp["y"][:1].clone()
    y_null = bf16r(sd["y_embedder.y_embedding"])[None, None]
    sched = rf.RFLOW(num_sampling_steps=4, cfg_scale=7.0, use_timestep_transform=True)
    margs = dict(y=y, mask=inp["mask"], height=inp["height"][:1], width=inp["width"][:1],
                 num_frames=torch.tensor([17.0]), fps=inp["fps"][:1])
    zs = []
    orig_fwd = ref.forward

    def fwd(*a, **k):
        return orig_fwd(*a, **k)

    # dist.get_rank() is only touched when progress=True
    with torch.no_grad():
        z_out = sched.sample(ref, z.clone(), margs, y_null, device="cpu", progress=False)
    all_ts = margs["all_timesteps"]
    # full-size timestep tables for BASELINE config 2 (512x512x64f, 30 steps): pure scalar math
    ts30 = [rf.timestep_transform(torch.tensor([(1.0 - i / 30) * 1000]),
                                  dict(height=torch.tensor([512.0]), width=torch.tensor([512.0]),
                                       num_frames=torch.tensor([64.0])), num_timesteps=1000) for i in range(30)]
    ts30 = torch.cat(ts30)
    fix = dict(cfg=SMALL_CFG, seed=SMALL_SEED, z0=z, y=y, y_null=y_null, mask=inp["mask"],
               height=inp["height"][:1], width=inp["width"][:1], num_frames=torch.tensor([17.0]),
               fps=inp["fps"][:1], steps=4, cfg_scale=7.0, z_out=z_out, all_timesteps=all_ts,
               ts30_c2=ts30, ts30_c2_bf16_int=[int(t.to(torch.bfloat16).item()) for t in ts30])
    torch.save(pack(fix), os.path.join(OUT, "rflow_small.pt"))
    print("rflow_small", z_out.shape, all_ts, fix["ts30_c2_bf16_int"][:6])


def make_ops():
    """Per-op goldens from the reference's own modules (fp32, bf16-rounded inputs/weights)."""
    mods = ref_loader.load_reference_modules()
    att, norm, emb, st = mods["attentions"], mods["normalization"], mods["embeddings"], mods["stdit3"]
    from rotary_embedding_torch import RotaryEmbedding  # the restated stub (third-party)

    g = torch.Generator().manual_seed(99)
    fx = {}

    # K3 LlamaRMSNorm (normalization.py:19-33), D=72
    rn = norm.LlamaRMSNorm(72)
    rn.weight.data = bf16r(1 + 0.1 * torch.randn(72, generator=g))
    x = bf16r(torch.randn(3, 4, 37, 72, generator=g) * 2)
    fx["rmsnorm"] = dict(x=x, w=rn.weight.data.clone(), out=rn(x).detach())

    # K4 rotary (third-party restated) on [B',H,T,72]
    rope = RotaryEmbedding(dim=72)
    xq = bf16r(torch.randn(5, 2, 19, 72, generator=g))
    fx["rope"] = dict(x=xq, freqs=rope.freqs.data.clone(), out=rope.rotate_queries_or_keys(xq).detach())

    # K1 LayerNorm + t2i_modulate (open_sora_transformer_3d.py:47-48,117)
    C = 576
    xl = bf16r(torch.randn(2, 70, C, generator=g) * 1.5 + 0.3)
    sh = bf16r(torch.randn(2, 1, C, generator=g) * 0.3)
    sc = bf16r(torch.randn(2, 1, C, generator=g) * 0.3)
    ln = torch.nn.LayerNorm(C, eps=1e-6, elementwise_affine=False)
    fx["adaln"] = dict(x=xl, shift=sh, scale=sc, out=st.t2i_modulate(ln(xl), sh, sc).detach())

    # K2-K7 OpenSoraAttention spatial (no rope) and temporal (rope), C=144, H=2, D=72 (small: fixture size)
    Ca, Ha = 144, 2

    def rnd_linear(m):
        m.weight.data = bf16r(torch.randn(m.weight.shape, generator=g) * (1.0 / m.weight.shape[1] ** 0.5))
        m.bias.data = bf16r(torch.randn(m.bias.shape, generator=g) * 0.05)

    for name, N, Bp, use_rope in (("attn_spatial", 200, 2, False), ("attn_temporal", 19, 16, True),
                                  ("attn_temporal38", 38, 6, True)):
        a = att.OpenSoraAttention(Ca, num_heads=Ha, qkv_bias=True, qk_norm=True,
                                  norm_layer=norm.LlamaRMSNorm,
                                  rope=rope.rotate_queries_or_keys if use_rope else None)
        rnd_linear(a.qkv)
        rnd_linear(a.proj)
        a.q_norm.weight.data = bf16r(1 + 0.1 * torch.randn(72, generator=g))
        a.k_norm.weight.data = bf16r(1 + 0.1 * torch.randn(72, generator=g))
        xi = bf16r(torch.randn(Bp, N, Ca, generator=g))
        pre = []
        hk = a.proj.register_forward_hook(lambda m, i, o: pre.append(i[0].detach().clone()))
        with torch.no_grad():
            qkv = a.qkv(xi)
            out = a(xi)
        hk.remove()
        fx[name] = dict(x=xi, heads=Ha, attn_out=pre[0], qkv_w=a.qkv.weight.data.clone(), qkv_b=a.qkv.bias.data.clone(),
                        proj_w=a.proj.weight.data.clone(), proj_b=a.proj.bias.data.clone(),
                        q_norm=a.q_norm.weight.data.clone(), k_norm=a.k_norm.weight.data.clone(),
                        rope_freqs=rope.freqs.data.clone() if use_rope else None, qkv=qkv, out=out)

    # K8-K10 cross attention (attentions.py:135-270), packed text with per-sample valid length
    ca = att.OpenSoraMultiHeadCrossAttention(Ca, Ha)
    for m in (ca.q_linear, ca.kv_linear, ca.proj):
        rnd_linear(m)
    xq = bf16r(torch.randn(2, 150, Ca, generator=g))
    Lv = 13
    cond = bf16r(torch.randn(1, 2 * Lv, Ca, generator=g))
    pre = []
    hk = ca.proj.register_forward_hook(lambda m, i, o: pre.append(i[0].detach().clone()))
    with torch.no_grad():
        out = ca(xq, cond, [Lv, Lv])
        q_lin = ca.q_linear(xq)
        kv_lin = ca.kv_linear(cond)
    hk.remove()
    fx["attn_cross"] = dict(x=xq, cond=cond, y_lens=[Lv, Lv], heads=Ha, attn_out=pre[0], q=q_lin, kv=kv_lin,
                            q_w=ca.q_linear.weight.data.clone(), q_b=ca.q_linear.bias.data.clone(),
                            kv_w=ca.kv_linear.weight.data.clone(), kv_b=ca.kv_linear.bias.data.clone(),
                            proj_w=ca.proj.weight.data.clone(), proj_b=ca.proj.bias.data.clone(), out=out)

    # a4 embedders: timestep sinusoid, 2-D pos-emb (embeddings.py:107-146,231-280)
    te = emb.TimestepEmbedder.timestep_embedding(torch.tensor([712.0, 3.0, 999.0]), 256)
    pe = emb.OpenSoraPositionEmbedding2D(C)
    pos = pe._get_cached_emb(torch.device("cpu"), torch.float32, 8, 8, scale=0.25, base_size=8)
    fx["embed"] = dict(t=torch.tensor([712.0, 3.0, 999.0]), t_freq=te, pos_hw=(8, 8), pos_scale=0.25,
                       pos_base=8, pos=pos)

    # a11 final layer + unpatchify (open_sora_transformer_3d.py:51-87,634-658)
    fl = st.T2IFinalLayer(C, 4, 8)
    rnd_linear(fl.linear)
    fl.scale_shift_table.data = bf16r(fl.scale_shift_table.data)
    xf = bf16r(torch.randn(2, 3 * 16, C, generator=g))
    tf = bf16r(torch.randn(2, C, generator=g) * 0.3)
    with torch.no_grad():
        of = fl(xf, tf)
    fx["final"] = dict(x=xf, t=tf, table=fl.scale_shift_table.data.clone(), w=fl.linear.weight.data.clone(),
                       b=fl.linear.bias.data.clone(), out=of)

    torch.save(pack(fx), os.path.join(OUT, "ops_small.pt"))
    print("ops_small", {k: tuple(v["out"].shape) if "out" in v else None for k, v in fx.items()})


def make_pab_schedule():
    """Flag sequences of the reference PABManager (pab_mgr.py:54-91) for Open-Sora defaults
    (pipeline_open_sora.py:32-69: spatial [450,930]/2, temporal [450,930]/4, cross [450,930]/6), 30 steps, C2."""
    mods = ref_loader.load_reference_modules()
    pab, rf = mods["pab_mgr"], mods["rflow"]
    cfg = pab.PABConfig(cross_broadcast=True, cross_threshold=[450, 930], cross_range=6,
                        spatial_broadcast=True, spatial_threshold=[450, 930], spatial_range=2,
                        temporal_broadcast=True, temporal_threshold=[450, 930], temporal_range=4)
    mgr = pab.PABManager(cfg)
    cfg.steps = 30
    ts = [rf.timestep_transform(torch.tensor([(1.0 - i / 30) * 1000]),
                                dict(height=torch.tensor([512.0]), width=torch.tensor([512.0]),
                                     num_frames=torch.tensor([64.0])), num_timesteps=1000) for i in range(30)]
    ts_int = [int(t.to(torch.bfloat16)[0]) for t in ts]
    flags = {"spatial": [], "temporal": [], "cross": []}
    cnt = {"spatial": 0, "temporal": 0, "cross": 0}
    for rep in range(2):  # two generate() calls back to back: counters wrap modulo steps
        for t in ts_int:
            for kind, fn in (("spatial", mgr.if_broadcast_spatial), ("temporal", mgr.if_broadcast_temporal),
                             ("cross", mgr.if_broadcast_cross)):
                f, cnt[kind] = fn(t, cnt[kind])
                flags[kind].append(bool(f))
    with open(os.path.join(OUT, "pab_schedule_c2.json"), "w") as f:
        json.dump(dict(steps=30, timesteps_int=ts_int, flags=flags,
                       cfg=dict(spatial=[450, 930, 2], temporal=[450, 930, 4], cross=[450, 930, 6])), f)
    print("pab_schedule_c2", ts_int[:5], sum(flags["spatial"]), sum(flags["temporal"]), sum(flags["cross"]))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    make_ops()
    make_pab_schedule()
    make_stdit3_fwd()
    make_rflow()
