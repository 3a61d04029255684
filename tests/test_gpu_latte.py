"""GPU parity for the Latte path: the HIP LatteT2V / DDIM loop / PAB schedule against fixtures
minted from the reference's LatteT2V class, plus the three kernels the path adds (or generalises).
Tolerances (stated by this repo, the reference pins none): bf16 kernels vs fp32 reference on bf16-rounded inputs —
max|err| <= 3e-2 * max|ref| and cosine >= 0.999 for whole-model outputs."""
import math

import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def check(out, ref, rel=3e-2, cos_min=0.999, what=""):
    out, ref = out.float().cpu(), ref.float().cpu()
    err = (out - ref).abs().max().item()
    scale = ref.abs().max().item()
    cos = torch.nn.functional.cosine_similarity(out.flatten(), ref.flatten(), dim=0).item()
    assert err <= rel * scale and cos >= cos_min, f"{what}: max|err| {err:.4e} vs {scale:.3f}, cosine {cos:.6f}"


def build(fx):
    from oracle import latte_oracle as LO
    from videosys_amd.latte import LatteT2V

    cfg = fx["cfg"]
    sd = LO.synth_state_dict(cfg["num_layers"], cfg["num_attention_heads"], cfg["attention_head_dim"],
                             caption_channels=cfg["caption_channels"], seed=fx["seed"])
    sd = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    m = LatteT2V(**cfg, device=dev())
    m.load_state_dict(sd)
    return m


def test_latte_forward_golden():
    from videosys_amd import pab

    pab.set_pab_manager(None)
    fx = load_golden("latte_fwd_small.pt")
    m = build(fx)
    out = m(fx["x"], timestep=fx["t"], encoder_hidden_states=fx["y"], encoder_attention_mask=fx["mask"], return_dict=False)[0]
    check(out, fx["out"], what="latte forward (masked text)")
    out = m(fx["x"], timestep=fx["t"], encoder_hidden_states=fx["y"].clone(), encoder_attention_mask=None, return_dict=False)[0]
    check(out, fx["out_nomask"], what="latte forward (no mask)")
    # CFG duplicate without torch.cat: a latent batch of 1 with a text batch of 2
    x1 = fx["x"][:1].repeat(2, 1, 1, 1, 1)
    a = m(x1, timestep=fx["t"], encoder_hidden_states=fx["y"].clone(), encoder_attention_mask=fx["mask"], return_dict=False)[0]
    b = m(fx["x"][:1], timestep=fx["t"], encoder_hidden_states=fx["y"].clone(), encoder_attention_mask=fx["mask"],
          return_dict=False)[0]
    assert torch.equal(a, b)


def test_latte_sampling_golden():
    from videosys_amd import LatteConfig, LattePipeline

    fx = load_golden("latte_sample_small.pt")
    pipe = LattePipeline(LatteConfig(model_path=f"synthetic:{fx['seed']}", transformer_config=fx["cfg"]), device=dev())
    out = pipe.generate(prompt_embeds=fx["pos"], negative_prompt_embeds=fx["neg"], prompt_mask=fx["pmask"],
                        negative_mask=fx["nmask"], latents=fx["latents"], num_inference_steps=fx["steps"],
                        guidance_scale=fx["guidance"], output_type="latent").video
    assert pipe.scheduler.timesteps == fx["timesteps"]
    check(out, fx["out"], rel=5e-2, what=f"latte {fx['steps']}-step DDIM latents")


def test_latte_without_classifier_free_guidance():
    """guidance_scale <= 1 (do_classifier_free_guidance False, pipeline_latte.py:749,828-831): the model runs on the prompt batch
    alone.  The CFG run with the negative prompt EQUAL to the prompt computes uncond + g (cond - uncond) with cond == uncond bit
    for bit, i.e. the same prediction: both runs must give the same latents exactly."""
    from videosys_amd import LatteConfig, LattePipeline

    fx = load_golden("latte_sample_small.pt")
    pipe = LattePipeline(LatteConfig(model_path=f"synthetic:{fx['seed']}", transformer_config=fx["cfg"]), device=dev())
    kw = dict(prompt_embeds=fx["pos"], prompt_mask=fx["pmask"], latents=fx["latents"], num_inference_steps=fx["steps"], output_type="latent")
    a = pipe.generate(negative_prompt_embeds=fx["pos"], negative_mask=fx["pmask"], guidance_scale=fx["guidance"], **kw).video
    b = pipe.generate(guidance_scale=1.0, **kw).video
    assert torch.equal(a, b)
    assert torch.isfinite(b).all() and not torch.equal(b.cpu(), fx["latents"].to(b.dtype))


def test_latte_text_prompts_follow_encode_prompt_masking():
    """Text prompts through generate() (encode_prompt, pipeline_latte.py:287-445): a single prompt's embeddings — and the negative
    ones with them — are cut to the prompt's token count and the transformer sees no attention mask; with mask_feature=False the
    full padded length is attended.  Checked against generate() fed the equivalent embeddings directly (bit-equal)."""
    from videosys_amd import LatteConfig, LattePipeline

    class FakeText:   # (embeddings [B, 1, 16, 64], mask [B, 16]) like t5.T5TextEncoder; deterministic in the prompt text
        def __call__(self, prompts):
            prompts = [prompts] if isinstance(prompts, str) else list(prompts)
            emb, mask = [], torch.zeros(len(prompts), 16, dtype=torch.long)
            for b, q in enumerate(prompts):
                g = torch.Generator().manual_seed(sum(q.encode()) + 7)
                emb.append(torch.randn(1, 16, 64, generator=g).to(torch.bfloat16))
                mask[b, :min(len(q) + 1, 16)] = 1
            return torch.stack(emb, 0), mask

    fx = load_golden("latte_sample_small.pt")
    te = FakeText()
    pipe = LattePipeline(LatteConfig(model_path=f"synthetic:{fx['seed']}", transformer_config=fx["cfg"]), device=dev(), text_encoder=te)
    kw = dict(latents=fx["latents"], num_inference_steps=fx["steps"], guidance_scale=fx["guidance"], output_type="latent")
    a = pipe.generate(prompt="a red cat", negative_prompt="", clean_caption=False, **kw).video
    pe, pm = te("a red cat")
    ne, _ = te("")
    keep = int(pm.sum())
    assert keep == 10
    b = pipe.generate(prompt_embeds=pe[:, 0, :keep], negative_prompt_embeds=ne[:, 0, :keep], **kw).video
    assert torch.equal(a, b)
    c = pipe.generate(prompt="a red cat", negative_prompt="", clean_caption=False, mask_feature=False, **kw).video
    d = pipe.generate(prompt_embeds=pe[:, 0], negative_prompt_embeds=ne[:, 0], **kw).video
    assert torch.equal(c, d) and not torch.equal(a, c)
    seen = []
    out = pipe.generate(prompt="a red cat", clean_caption=True, return_dict=False, callback=lambda i, t, z: seen.append((i, t)), callback_steps=2, **kw)
    assert isinstance(out, tuple) and torch.equal(out[0], a) and [i for i, _ in seen] == list(range(0, fx["steps"], 2))
    with pytest.raises(NotImplementedError):
        pipe.generate(prompt="x", eta=0.5, **kw)


def test_latte_pab_golden():
    from videosys_amd import pab

    fx = load_golden("latte_pab_small.pt")
    m = build(fx)
    pab.set_pab_manager(pab.PABConfig(spatial_broadcast=True, temporal_broadcast=True, cross_broadcast=True, mlp_broadcast=True,
                                      **fx["pab"]))
    try:
        pab.update_steps(fx["steps"])
        m.reset_pab_state()
        ats = torch.tensor(fx["timesteps"])
        for i, t in enumerate(fx["timesteps"]):
            out = m(fx["x"], timestep=torch.tensor([t, t]), all_timesteps=ats, encoder_hidden_states=fx["y"],
                    encoder_attention_mask=fx["mask"], return_dict=False)[0]
            check(out, fx["outs"][i], what=f"latte PAB step {i} (t={t})")
    finally:
        pab.set_pab_manager(None)


def test_cfg_linear_step_and_bcast_add():
    from videosys_amd import ops

    g = torch.Generator().manual_seed(0)
    z = torch.randn(2, 4, 3, 5, 7, generator=g)
    mo = torch.randn(4, 8, 3, 5, 7, generator=g)
    for cond_first in (False, True):
        zc = z.to(dev()).clone()
        ops.cfg_linear_step(zc, mo.to(dev()), 7.5, 0.93, -0.21, cond_first=cond_first)
        a, b = mo[:2, :4], mo[2:, :4]
        cond, unc = (a, b) if cond_first else (b, a)
        torch.testing.assert_close(zc.cpu(), 0.93 * z - 0.21 * (unc + 7.5 * (cond - unc)), rtol=1e-5, atol=1e-5)
    x = torch.randn(2 * 3 * 5, 576, generator=g).to(torch.bfloat16)
    e = torch.randn(4, 576, generator=g).to(torch.bfloat16)
    xd = x.to(dev()).clone()
    ops.add_bcast_rows(xd, e.to(dev()), 5, 3)  # rows (b, f, s): S = 5, F = 3
    ref = (x.float().view(2, 3, 5, 576) + e.float()[:3].view(1, 3, 1, 576)).to(torch.bfloat16)
    assert torch.equal(xd.cpu().view(2, 3, 5, 576), ref)


def test_attn_temporal_without_qk_norm():
    from videosys_amd import ops

    B, T, S, H, D = 2, 16, 24, 8, 72
    C = H * D
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(B * T * S, 3 * C, generator=g).to(torch.bfloat16)
    out = torch.empty(B * T * S, C, dtype=torch.bfloat16, device=dev())
    ops.attn_temporal(qkv.to(dev()), C, None, None, None, None, out, B, T, S, H)
    q, k, v = [t.float().view(B, T, S, H, D).permute(0, 2, 3, 1, 4) for t in qkv.split(C, dim=1)]  # [B, S, H, T, D]
    ref = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(D), -1) @ v
    ref = ref.permute(0, 3, 1, 2, 4).reshape(B * T * S, C)
    check(out, ref, rel=2 ** -7, what="temporal attention, no qk-norm / no rope")


def test_flash_attn_short_key_length_inside_a_longer_buffer():
    """Per-sample text lengths: kv_len far below the buffer's padded length must ignore whole tiles of stale keys."""
    from videosys_amd import ops

    H, D, Lbuf, Lk, Nq = 8, 72, 150, 40, 200
    C = H * D
    g = torch.Generator().manual_seed(2)
    q = torch.randn(Nq, C, generator=g).to(torch.bfloat16)
    kv = torch.randn(Lbuf, 2 * C, generator=g).to(torch.bfloat16)
    kp, vt = ops.alloc_kv_buffers(1, H, Lbuf, dev())
    kvd = kv.to(dev())
    ops.attn_prep_kv(kvd[:, :C], kvd[:, C:], None, kp, vt, 1, H, Lbuf)
    out = torch.empty(Nq, C, dtype=torch.bfloat16, device=dev())
    ops.flash_attn(q.to(dev()), None, kp, vt, out, 1, H, Nq, Lk)
    qh = q.float().view(Nq, H, D).transpose(0, 1)
    kh = kv[:Lk, :C].float().view(Lk, H, D).transpose(0, 1)
    vh = kv[:Lk, C:].float().view(Lk, H, D).transpose(0, 1)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(D), -1) @ vh).transpose(0, 1).reshape(Nq, C)
    check(out, ref, rel=2 ** -7, what="flash attention with kv_len << kv_pad")
