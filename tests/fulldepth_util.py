"""Full-depth latent parity at BASELINE geometry (north_star: "outputs match the reference pipeline's latents within a stated fp
tolerance on fixed seeds").  TEST INFRASTRUCTURE: used by tests/test_gpu_fulldepth.py (assertions) and tools/parity_full_depth.py
(writes the tables committed under profiles/).

Method.  The oracle (oracle/stdit3_oracle.py, oracle/latte_oracle.py: restatements pinned against the real reference classes by
tests/test_oracle_vs_golden.py) is plain PyTorch, so it runs ON THE GPU AS THE CHECKER:
  * ``ref``   = oracle in fp32 (the reference's arithmetic without rounding) on the same bf16-rounded weights and inputs;
  * ``floor`` = the same oracle executed with torch's bf16 kernels, i.e. how the reference itself runs this model
                (``model.to(bf16)``); its distance from ``ref`` is the bf16 noise floor of the REFERENCE;
  * ``hip``   = the product (hand-written HIP kernels through the C ABI).
Stated tolerance: rel-rms(hip, ref) <= 1.15 x rel-rms(floor, ref) on the output latents (1.5 x on every block pair's hidden state),
and cosine(hip, ref) >= 0.999 whenever the reference's own bf16 run reaches 0.999 (with untrained random weights a 56-block
stack amplifies rounding noise; where the floor itself drops below 0.999 the requirement is cosine(hip) >= cosine(floor) - 5e-4).
"""
from __future__ import annotations

import torch


def stats(out: torch.Tensor, ref: torch.Tensor) -> dict:
    o, r = out.detach().float().flatten(), ref.detach().float().flatten()
    d = o - r
    return dict(rel_rms=float(d.norm() / r.norm().clamp_min(1e-30)), max_abs=float(d.abs().max()), ref_max=float(r.abs().max()),
                cosine=float(torch.nn.functional.cosine_similarity(o, r, dim=0)))


def bf16_round(sd):
    return {k: (v if k == "rope.freqs" else v.to(torch.bfloat16).float()) for k, v in sd.items()}


# ------------------------------------------------------------------------------------------------ Open-Sora, config 2 / 3
def opensora_inputs(T=19, HW=64, L=300, caption_channels=4096, seed=0):
    """SURVEY §8d 'C2': z ~ N(0,1) [1,4,T,HW,HW] with manual_seed(0), y = 0.1 randn [1,1,300,4096], mask = L ones."""
    g = torch.Generator().manual_seed(seed)
    if isinstance(HW, tuple):   # (latent height, latent width, frames): a non-square geometry, e.g. configs[3] 720p x 128f = (90, 160, 128)
        Hl, Wl, frames = HW
        z = torch.randn(1, 4, T, Hl, Wl, generator=g).to(torch.bfloat16).float()
        y = (torch.randn(1, 1, 300, caption_channels, generator=g) * 0.1).to(torch.bfloat16).float()
        mask = torch.zeros(1, 300, dtype=torch.long)
        mask[:, :L] = 1
        geom = dict(fps=torch.tensor([24.0]), height=torch.tensor([float(Hl * 8)]), width=torch.tensor([float(Wl * 8)]),
                    num_frames=torch.tensor([float(frames)]))
        return z, y, mask, geom
    z = torch.randn(1, 4, T, HW, HW, generator=g).to(torch.bfloat16).float()
    y = (torch.randn(1, 1, 300, caption_channels, generator=g) * 0.1).to(torch.bfloat16).float()
    mask = torch.zeros(1, 300, dtype=torch.long)
    mask[:, :L] = 1
    px = float(HW * 8)
    geom = dict(fps=torch.tensor([24.0]), height=torch.tensor([px]), width=torch.tensor([px]),
                num_frames=torch.tensor([float({19: 64, 5: 17}.get(T, 64))]))
    return z, y, mask, geom


def opensora_models(depth=28, seed=1234, device="cuda:0"):
    from oracle import stdit3_oracle as O
    from videosys_amd.stdit3 import STDiT3, STDiT3Config

    sd = bf16_round(O.synth_state_dict(depth, 1152, 16, seed=seed))
    hip = STDiT3(STDiT3Config(depth=depth), device=device)
    hip.load_state_dict(sd)
    ref = O.STDiT3Oracle(sd, depth, 1152, 16, device=device, dtype=torch.float32)
    floor = O.STDiT3Oracle(sd, depth, 1152, 16, device=device, dtype=torch.bfloat16)
    y_null = sd["y_embedder.y_embedding"][None, None]   # [1,1,300,4096]  (pipeline_open_sora.py:295)
    return hip, ref, floor, y_null


def opensora_one_step(hip, ref, floor, y_null, z, y, mask, geom, t_value=700.0, x_mask=None):
    """One CFG-batched forward at one timestep: output + per-block-pair hidden-state error growth.  ``x_mask`` [2, T] bool: the
    conditioning-mask form of the step (frames with False see the timestep-0 modulation)."""
    x = torch.cat([z, z], 0)
    yy = torch.cat([y, y_null], 0)
    t = torch.tensor([t_value, t_value]).to(torch.bfloat16).float()   # STDiT3.forward casts timestep to the model dtype (:562)
    kw = dict(mask=mask, fps=geom["fps"].repeat(2), height=geom["height"].repeat(2), width=geom["width"].repeat(2))
    if x_mask is not None:
        kw["x_mask"] = x_mask
    ref_h = []
    out_ref = ref.forward(x, t, yy, return_hidden=lambda d, h: ref_h.append(h.detach().clone()), **kw)
    rows = []
    floor_stats = []
    out_floor = floor.forward(x, t, yy, return_hidden=lambda d, h: floor_stats.append(stats(h, ref_h[d])), **kw)
    hip_stats = []
    hip._hidden_tap = lambda d, h: hip_stats.append(stats(h.view(ref_h[d].shape), ref_h[d]))
    try:
        out_hip = hip(x, t, yy, **kw)
    finally:
        hip._hidden_tap = None
    torch.cuda.synchronize()
    for d, (a, b) in enumerate(zip(hip_stats, floor_stats)):
        rows.append(dict(pair=d, hip_rel_rms=a["rel_rms"], floor_rel_rms=b["rel_rms"], hip_cos=a["cosine"], floor_cos=b["cosine"]))
    return dict(out_hip=stats(out_hip, out_ref), out_floor=stats(out_floor, out_ref), per_pair=rows)


def opensora_rflow(hip, ref, floor, y_null, z, y, mask, geom, steps=3, cfg_scale=7.0, cond_mask=None):
    """RFLOW.sample through the product sampler vs the oracle sampler around ref / floor (same bf16 timesteps).  ``cond_mask``
    [1, T] float: mask-conditioned sampling (frames of ``z`` with a value < 1 are conditioning frames), same noise in all three."""
    from oracle import stdit3_oracle as O
    from videosys_amd.rflow import RFLOW

    sched = RFLOW(num_sampling_steps=steps, cfg_scale=cfg_scale, use_timestep_transform=True)
    margs = dict(y=y, mask=mask, **geom)
    extra, okw = {}, {}
    if cond_mask is not None:
        g = torch.Generator().manual_seed(99)
        noises = [torch.randn(z.shape, generator=g) for _ in range(steps)]
        fn = lambda: (lambda it: (lambda shape: next(it)))(iter(noises))
        extra = dict(mask=cond_mask, noise_fn=fn())
    z_hip = sched.sample(hip, z, margs, y_null, **extra).float().cpu()
    res = {}
    for name, m in (("ref", ref), ("floor", floor)):
        if cond_mask is not None:
            okw = dict(cond_mask=cond_mask, noise_fn=fn())
        res[name] = O.rflow_sample(m, z, y, y_null, mask, geom["fps"], geom["height"], geom["width"], geom["num_frames"],
                                   num_sampling_steps=steps, cfg_scale=cfg_scale, model_dtype=torch.bfloat16, **okw)
    out = dict(steps=steps, z_hip=stats(z_hip, res["ref"]), z_floor=stats(res["floor"], res["ref"]))
    if cond_mask is not None:
        held = cond_mask[0] == 0
        out["held_frames_bit_exact"] = bool(torch.equal(z_hip[:, :, held], z[:, :, held].float().cpu()))
    return out


# ------------------------------------------------------------------------------------------------ Latte, config 1
def latte_config1(depth=28, device="cuda:0", t_value=500, seed=4321):
    """BASELINE config 1 geometry (latent [4,16,32,32], 120 text tokens, CFG batch 2), one full-depth forward."""
    from oracle import latte_oracle as LO
    from videosys_amd.latte import LatteT2V

    cfg = dict(num_attention_heads=16, attention_head_dim=72, num_layers=depth, caption_channels=4096, sample_size=64,
               video_length=16)
    sd = {k: v.to(torch.bfloat16).float() for k, v in LO.synth_state_dict(depth, 16, 72, seed=seed).items()}
    hip = LatteT2V(**cfg, device=device)
    hip.load_state_dict(sd)
    ref = LO.LatteOracle(sd, depth, 16, 72, sample_size=64, video_length=16, device=device, dtype=torch.float32)
    floor = LO.LatteOracle(sd, depth, 16, 72, sample_size=64, video_length=16, device=device, dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(1, 4, 16, 32, 32, generator=g).to(torch.bfloat16).float()
    pos = (torch.randn(1, 120, 4096, generator=g) * 0.1).to(torch.bfloat16).float()
    neg = (torch.randn(1, 120, 4096, generator=g) * 0.1).to(torch.bfloat16).float()
    pmask = torch.ones(1, 120, dtype=torch.long)
    pmask[:, 77:] = 0
    nmask = torch.ones(1, 120, dtype=torch.long)
    nmask[:, 9:] = 0
    x = torch.cat([lat, lat])
    emb, mask = torch.cat([neg, pos]), torch.cat([nmask, pmask])
    tt = torch.tensor([t_value, t_value])
    ref_h = []
    out_ref = ref(x, tt, emb, mask, on_hidden=lambda i, h: ref_h.append(h.detach().clone()))
    floor_stats, hip_stats = [], []
    out_floor = floor(x, tt, emb, mask, on_hidden=lambda i, h: floor_stats.append(stats(h, ref_h[i])))
    hip._hidden_tap = lambda i, h: hip_stats.append(stats(h.view(ref_h[i].shape), ref_h[i]))
    try:
        out_hip = hip(x, timestep=tt, encoder_hidden_states=emb, encoder_attention_mask=mask, return_dict=False)[0]
    finally:
        hip._hidden_tap = None
    torch.cuda.synchronize()
    rows = [dict(pair=d, hip_rel_rms=a["rel_rms"], floor_rel_rms=b["rel_rms"], hip_cos=a["cosine"], floor_cos=b["cosine"])
            for d, (a, b) in enumerate(zip(hip_stats, floor_stats))]
    return dict(out_hip=stats(out_hip, out_ref), out_floor=stats(out_floor, out_ref), per_pair=rows)


def verdict(hip: dict, floor: dict, factor=1.15) -> str:
    """'' if the product is inside the stated tolerance, else the reason."""
    if not hip["rel_rms"] <= factor * floor["rel_rms"]:
        return f"rel-rms {hip['rel_rms']:.4e} > {factor} x reference-bf16 floor {floor['rel_rms']:.4e}"
    need = 0.999 if floor["cosine"] >= 0.999 else floor["cosine"] - 5e-4
    if not hip["cosine"] >= need:
        return f"cosine {hip['cosine']:.6f} < {need:.6f} (reference-bf16 floor {floor['cosine']:.6f})"
    return ""


# ------------------------------------------------------------------------------------------------ config 3: PAB over the schedule
def opensora_pab_schedule(hip, ref, floor, y_null, z, y, mask, geom, steps=30, cfg_scale=7.0, mlp_rule=None, oracle=True):
    """BASELINE config 3 (OpenSoraPABConfig defaults: spatial [450,930]/2, temporal /4, cross /6; ``mlp_rule`` = None for the
    attention-only form BASELINE names, or the {timestep: {"block", "skip_count"}} dict used for BOTH the spatial and the temporal
    MLP broadcast, pipeline_open_sora.py:44-54) over the whole 30-step RFLOW schedule; ref / floor run the SAME broadcast
    schedule (oracle PABSchedule = pab_mgr.py:54-141 restated).  ``oracle=False`` runs only the product (returns its latent)."""
    from oracle import stdit3_oracle as O
    from videosys_amd import pab
    from videosys_amd.rflow import RFLOW

    sched = RFLOW(num_sampling_steps=steps, cfg_scale=cfg_scale, use_timestep_transform=True)
    margs = dict(y=y, mask=mask, **geom)
    kw = dict(mlp_broadcast=True, mlp_spatial_broadcast_config=mlp_rule, mlp_temporal_broadcast_config=mlp_rule) if mlp_rule else {}
    pab.set_pab_manager(pab.PABConfig(spatial_broadcast=True, spatial_threshold=[450, 930], spatial_range=2,
                                      temporal_broadcast=True, temporal_threshold=[450, 930], temporal_range=4,
                                      cross_broadcast=True, cross_threshold=[450, 930], cross_range=6, **kw))
    pab.update_steps(steps)
    try:
        hip.reset_pab_state()
        z_hip = sched.sample(hip, z, margs, y_null).float().cpu()
        left = (len(pab.PAB_MANAGER.config.mlp_spatial_outputs), len(pab.PAB_MANAGER.config.mlp_temporal_outputs))
    finally:
        pab.set_pab_manager(None)
        hip.reset_pab_state()
    if not oracle:
        return dict(steps=steps, z=z_hip, mlp_left=left)
    res = {}
    for name, m in (("ref", ref), ("floor", floor)):
        m.set_pab(O.PABSchedule(steps, spatial=(450, 930, 2), temporal=(450, 930, 4), cross=(450, 930, 6),
                                mlp_spatial=mlp_rule, mlp_temporal=mlp_rule))
        try:
            res[name] = O.rflow_sample(m, z, y, y_null, mask, geom["fps"], geom["height"], geom["width"], geom["num_frames"],
                                       num_sampling_steps=steps, cfg_scale=cfg_scale, model_dtype=torch.bfloat16)
        finally:
            m.set_pab(None)
        torch.cuda.empty_cache()
    return dict(steps=steps, z_hip=stats(z_hip, res["ref"]), z_floor=stats(res["floor"], res["ref"]), z=z_hip, mlp_left=left)


# ------------------------------------------------------------------------------------------------ CogVideoX, config 5 geometry
def cogvideox_models(model="5b", layers=None, seed=777, device="cuda:0"):
    """CogVideoX-5B (48 heads x 64, 42 layers, 3-D RoPE) / -2B (30 heads, 30 layers, sincos table) geometry, seeded synthetic
    weights (bf16-representable): product, fp32 oracle and the oracle's bf16 run, all on the GPU."""
    from oracle import cogvideox_oracle as CO
    from videosys_amd.cogvideox import CogVideoXTransformer3DModel

    geo = dict(num_attention_heads=48, num_layers=42, use_rotary_positional_embeddings=True) if model == "5b" else \
        dict(num_attention_heads=30, num_layers=30, use_rotary_positional_embeddings=False)
    if layers:
        geo["num_layers"] = layers
    sd = CO.synth_state_dict(geo["num_layers"], geo["num_attention_heads"], seed=seed)
    sd = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    hip = CogVideoXTransformer3DModel(**geo, device=torch.device(device))
    hip.load_state_dict(sd)
    kw = dict(num_layers=geo["num_layers"], num_heads=geo["num_attention_heads"],
              use_rotary_positional_embeddings=geo["use_rotary_positional_embeddings"], device=device)
    ref = CO.CogVideoXOracle(sd, dtype=torch.float32, **kw)
    floor = CO.CogVideoXOracle(sd, dtype=torch.bfloat16, **kw)
    return hip, ref, floor, geo


def cogvideox_one_step(hip, ref, floor, geo, frames=13, hh=60, ww=90, text_len=226, t_value=801, seed=0):
    """One CFG-batched forward (BASELINE config 5: 720x480x49f -> latent [13, 16, 60, 90], 17 550 video + 226 text rows per
    sample): output + per-block hidden-state error against the fp32 oracle, next to the oracle's own bf16 run."""
    from oracle import cogvideox_oracle as CO

    g = torch.Generator().manual_seed(seed)
    x = torch.randn(1, frames, 16, hh, ww, generator=g).to(torch.bfloat16).float().repeat(2, 1, 1, 1, 1)
    y = (torch.randn(2, text_len, 4096, generator=g) * 0.1).to(torch.bfloat16).float()
    t = torch.tensor([t_value, t_value])
    rope = CO.prepare_rope(hh * 8, ww * 8, frames, 64) if geo["use_rotary_positional_embeddings"] else None
    ref_h = []
    out_ref = ref.forward(x, y, t, image_rotary_emb=rope, on_hidden=lambda i, h: ref_h.append(h.detach().clone()))
    floor_stats = []
    out_floor = floor.forward(x, y, t, image_rotary_emb=rope, on_hidden=lambda i, h: floor_stats.append(stats(h, ref_h[i])))
    hip_stats = []
    hip._hidden_tap = lambda i, h: hip_stats.append(stats(h, ref_h[i]))
    try:
        out_hip = hip(x, y, t, image_rotary_emb=rope, return_dict=False)[0]
    finally:
        hip._hidden_tap = None
    torch.cuda.synchronize()
    rows = [dict(block=i, hip_rel_rms=a["rel_rms"], floor_rel_rms=b["rel_rms"], hip_cos=a["cosine"], floor_cos=b["cosine"])
            for i, (a, b) in enumerate(zip(hip_stats, floor_stats))]
    return dict(out_hip=stats(out_hip, out_ref), out_floor=stats(out_floor, out_ref), per_block=rows)
