"""Pins oracle/stdit3_oracle.py (the CPU restatement) against fixtures minted from the REAL reference
(oracle/make_golden.py) and, when /root/reference is present, against the live reference."""
import json
import os

import pytest
import torch

from conftest import GOLDEN, load_golden
from oracle import stdit3_oracle as O


def close(a, b, rtol=2e-5, atol=2e-5):
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol)


def test_rmsnorm(golden_ops):
    f = golden_ops["rmsnorm"]
    close(O.rms_norm(f["x"], f["w"]), f["out"])


def test_rope(golden_ops):
    f = golden_ops["rope"]
    close(O.rope_rotate(f["x"], f["freqs"]), f["out"])


def test_adaln(golden_ops):
    f = golden_ops["adaln"]
    close(O.t2i_modulate(O.layer_norm(f["x"]), f["shift"], f["scale"]), f["out"], 1e-4, 1e-4)


@pytest.mark.parametrize("name", ["attn_spatial", "attn_temporal", "attn_temporal38"])
def test_self_attention(golden_ops, name):
    f = golden_ops[name]
    sd = {"a.qkv.weight": f["qkv_w"], "a.qkv.bias": f["qkv_b"], "a.proj.weight": f["proj_w"],
          "a.proj.bias": f["proj_b"], "a.q_norm.weight": f["q_norm"], "a.k_norm.weight": f["k_norm"]}
    out = O.self_attention(f["x"], sd, "a", f["heads"], f["rope_freqs"])
    close(out, f["out"], 1e-4, 1e-4)


def test_cross_attention(golden_ops):
    f = golden_ops["attn_cross"]
    sd = {"c.q_linear.weight": f["q_w"], "c.q_linear.bias": f["q_b"], "c.kv_linear.weight": f["kv_w"],
          "c.kv_linear.bias": f["kv_b"], "c.proj.weight": f["proj_w"], "c.proj.bias": f["proj_b"]}
    out = O.cross_attention(f["x"], f["cond"], f["y_lens"], sd, "c", f["heads"])
    close(out, f["out"], 1e-4, 1e-4)


def test_embed(golden_ops):
    f = golden_ops["embed"]
    close(O.timestep_embedding(f["t"]), f["t_freq"])
    close(O.pos_embed_2d(576, *f["pos_hw"], f["pos_scale"], f["pos_base"]), f["pos"])


def test_final_layer(golden_ops):
    f = golden_ops["final"]
    sd = {"final_layer.scale_shift_table": f["table"], "final_layer.linear.weight": f["w"],
          "final_layer.linear.bias": f["b"]}
    close(O.final_layer(f["x"], f["t"], sd), f["out"], 1e-4, 1e-4)


def _small_model(fx):
    cfg = fx["cfg"]
    sd = O.synth_state_dict(**cfg, seed=fx["seed"])
    sd = {k: (v if k == "rope.freqs" else v.to(torch.bfloat16).float()) for k, v in sd.items()}
    return O.STDiT3Oracle(sd, cfg["depth"], cfg["hidden_size"], cfg["num_heads"]), sd


def test_stdit3_forward_small():
    fx = load_golden("stdit3_fwd_small.pt")
    model, sd = _small_model(fx)
    chk = float(sum(v.double().abs().sum() for v in sd.values()))
    assert abs(chk - fx["sd_checksum"]) <= 1e-9 * fx["sd_checksum"], "synthetic weight generator drifted"
    i = fx["inputs"]
    out, hidden = model.forward(i["x"], i["timestep"], i["y"], mask=i["mask"], fps=i["fps"], height=i["height"],
                                width=i["width"], return_hidden=True)
    for h, g in zip(hidden, fx["hidden_rows"]):
        close(h[:, :: fx["hidden_stride"], :], g, 2e-4, 2e-4)
    close(out, fx["out"], 2e-4, 2e-4)


def test_stdit3_pab_small():
    fx = load_golden("stdit3_pab_small.pt")
    model, _ = _small_model(fx)
    p = fx["pab"]
    model.set_pab(O.PABSchedule(fx["steps"], p["spatial"], p["temporal"], p["cross"]))
    i = fx["inputs"]
    for t, ref in zip(fx["timesteps"], fx["outs"]):
        tt = torch.tensor([t, t])
        out = model.forward(i["x"], tt, i["y"], mask=i["mask"], fps=i["fps"], height=i["height"], width=i["width"])
        close(out, ref, 3e-4, 3e-4)


def test_stdit3_pab_mlp_small():
    """MLP broadcast (pab_mgr.py:93-174): oracle restatement vs the reference model with ``all_timesteps`` handed to its blocks
    (oracle/make_golden_pab_mlp.py), and the fixture really exercises it: without the MLP rules the replayed steps differ."""
    fx = load_golden("stdit3_pab_mlp_small.pt")
    model, _ = _small_model(fx)
    p = fx["pab"]
    i = fx["inputs"]

    def run(mlp):
        kw = dict(mlp_spatial=fx["mlp_spatial"], mlp_temporal=fx["mlp_temporal"]) if mlp else {}
        model.set_pab(O.PABSchedule(fx["steps"], p["spatial"], p["temporal"], p["cross"], **kw))
        outs = []
        for t in fx["timesteps"]:
            tt = torch.tensor([float(t), float(t)])
            outs.append(model.forward(i["x"], tt, i["y"], mask=i["mask"], fps=i["fps"], height=i["height"], width=i["width"],
                                      all_timesteps=fx["timesteps"]))
        return outs

    outs = run(True)
    for out, ref in zip(outs, fx["outs"]):
        close(out, ref, 3e-4, 3e-4)
    assert model.pab.mlp_store == {False: {}, True: {}} and tuple(fx["stored_left"]) == (0, 0)
    plain = run(False)
    changed = [float((a - b).abs().max()) > 1e-2 for a, b in zip(outs, plain)]
    assert changed == [False, False, True, True, True, True, False], changed   # replays: spatial 900->(800, 704), 640->400; temporal 800->(704, 640)


def test_rflow_small():
    fx = load_golden("rflow_small.pt")
    model, _ = _small_model(fx)
    z, zs, all_ts = O.rflow_sample(model, fx["z0"], fx["y"], fx["y_null"], fx["mask"], fx["fps"], fx["height"],
                                   fx["width"], fx["num_frames"], num_sampling_steps=fx["steps"],
                                   cfg_scale=fx["cfg_scale"], return_all=True)
    assert all_ts == fx["all_timesteps"]
    close(z, fx["z_out"], 5e-4, 5e-4)
    ts30 = O.rflow_timesteps(30, 1, torch.tensor([512.0]), torch.tensor([512.0]), torch.tensor([64.0]))
    close(torch.cat(ts30), fx["ts30_c2"], 1e-6, 1e-4)
    assert [int(t.to(torch.bfloat16).item()) for t in ts30] == fx["ts30_c2_bf16_int"]


def test_stdit3_xmask_small():
    """x_mask conditioning (open_sora_transformer_3d.py:181-184,198-200,220-222,262-273,578-582 + T2IFinalLayer :75-87)."""
    fx = load_golden("stdit3_xmask_small.pt")
    model, _ = _small_model(fx)
    i = fx["inputs"]
    kw = dict(mask=i["mask"], fps=i["fps"], height=i["height"], width=i["width"])
    out = model.forward(i["x"], i["timestep"], i["y"], x_mask=fx["x_mask"], **kw)
    close(out, fx["out"], 2e-4, 2e-4)
    out1 = model.forward(i["x"], i["timestep"], i["y"], x_mask=torch.ones(2, 5, dtype=torch.bool), **kw)
    close(out1, fx["out_all_true"], 2e-4, 2e-4)
    close(out1, model.forward(i["x"], i["timestep"], i["y"], **kw), 1e-6, 1e-6)   # an all-True mask is no mask
    assert float((fx["out"] - fx["out_all_true"]).abs().max()) > 0.5             # and the fixture's mask does matter


def test_rflow_masked_small():
    """Mask-conditioned sampling (scheduling_rflow_open_sora.py:215-236,254-255): the oracle draws the per-step noise from the
    same seeded global generator the reference's randn_like used."""
    fx = load_golden("stdit3_xmask_small.pt")
    model, _ = _small_model(fx)
    s = fx["sample"]
    seen = []

    def spy(*a, **k):
        seen.append(k["x_mask"].clone())
        return model(*a, **k)

    torch.manual_seed(s["noise_seed"])
    z, zs, all_ts = O.rflow_sample(spy, s["z0"], s["y"], s["y_null"], s["mask"], s["fps"], s["height"], s["width"],
                                   s["num_frames"], num_sampling_steps=s["steps"], cfg_scale=s["cfg_scale"], return_all=True,
                                   cond_mask=s["cond_mask"])
    assert all_ts == s["all_timesteps"]
    assert torch.equal(torch.stack(seen), s["x_masks"])
    close(z, s["z_out"], 5e-4, 5e-4)
    assert torch.equal(z[:, :, 0], s["z0"][:, :, 0])          # the held reference frame comes back untouched
    assert not torch.equal(z[:, :, 1], s["z0"][:, :, 1])      # the edited frame joined (ratio 0.6) and moved


def test_pab_schedule_c2():
    with open(os.path.join(GOLDEN, "pab_schedule_c2.json")) as f:
        g = json.load(f)
    c = g["cfg"]
    sch = O.PABSchedule(g["steps"], tuple(c["spatial"]), tuple(c["temporal"]), tuple(c["cross"]))
    cnt = {"spatial": 0, "temporal": 0, "cross": 0}
    flags = {k: [] for k in cnt}
    for _ in range(2):
        for t in g["timesteps_int"]:
            for k in cnt:
                fl, cnt[k] = sch.decide(k, t, cnt[k])
                flags[k].append(fl)
    assert flags == g["flags"]


def test_dsp_layout_roundtrip():
    """dynamic_switch semantics (open_sora_transformer_3d.py:288-315, comm.py:282-304) on in-process shards."""
    torch.manual_seed(0)
    B, T, S, C, P = 2, 5, 12, 8, 4
    x = torch.randn(B, T, S, C)
    shards = O.dsp_split_sequence(x, P, dim=2)  # S-shard at rest
    tp, sp = O.dsp_pad(T, P), O.dsp_pad(S, P)
    t_shards = O.dsp_all_to_all(shards, scatter_dim=1, gather_dim=2, scatter_pad=tp, gather_pad=sp)
    # every rank now holds whole frames: rank r has frames r*(T+tp)/P ...
    Tp = (T + tp) // P
    xp = torch.cat([x, torch.zeros(B, tp, S, C)], dim=1)
    for r in range(P):
        torch.testing.assert_close(t_shards[r], xp[:, r * Tp:(r + 1) * Tp])
    back = O.dsp_all_to_all(t_shards, scatter_dim=2, gather_dim=1, scatter_pad=sp, gather_pad=tp)
    for r in range(P):
        torch.testing.assert_close(back[r], shards[r])
    torch.testing.assert_close(O.dsp_gather_sequence(back, 2, sp), x)


@pytest.mark.skipif(not os.path.isdir("/root/reference/videosys"), reason="live reference only in the build container")
def test_live_reference_forward():
    from oracle import ref_loader

    cfg = dict(depth=1, hidden_size=144, num_heads=2, caption_channels=32, model_max_length=8)
    sd = O.synth_state_dict(**cfg, seed=5)
    ref = ref_loader.build_reference_stdit3(cfg, sd)
    model = O.STDiT3Oracle(sd, 1, 144, 2)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 4, 3, 9, 7, generator=g)  # odd H/W exercises the patch padding + unpatchify crop
    y = torch.randn(2, 1, 8, 32, generator=g)
    mask = torch.tensor([[1, 1, 1, 1, 1, 0, 0, 0]])
    kw = dict(mask=mask, fps=torch.tensor([24.0, 24.0]), height=torch.tensor([72.0, 72.0]),
              width=torch.tensor([56.0, 56.0]))
    t = torch.tensor([333.0, 333.0])
    with torch.no_grad():
        r = ref(x, t, y, **kw)
    close(model.forward(x, t, y, **kw), r, 2e-4, 2e-4)
