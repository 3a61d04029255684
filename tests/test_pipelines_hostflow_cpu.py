"""CPU (-m "not gpu"): the HOST FLOW of LattePipeline.generate and CogVideoXPipeline.generate — prompt handling, CFG batch order,
schedule, per-step coefficients, latent scaling, callbacks, interrupt — against the sampling fixtures minted from the reference
(tests/golden/latte_sample_small.pt, cogvideox_sample_small.pt).  There is no CPU execution path in the product: the transformer
handed to the pipeline constructor is the oracle restatement behind the transformer's call interface, and the one fused step
kernel the loops launch (vsys_cfg_linear_step) is emulated in torch — test scaffolding, like fake_ops in
test_stdit3_hostflow_cpu.py.  The same flows run on the real kernels in tests/test_gpu_latte.py / test_gpu_cogvideox.py."""
import contextlib
from types import SimpleNamespace

import pytest
import torch

from conftest import load_golden
from oracle import cogvideox_oracle as CO, latte_oracle as LO


@contextlib.contextmanager
def torch_step_kernel():
    """ops.cfg_linear_step with the arithmetic of vsys_cfg_linear_step: z = c_z z + c (u + g (c - u)) on the first Cin channels."""
    from videosys_amd import ops

    def step(z, out, guidance, c_z, c_eps, cond_first=False):
        B, cin = z.shape[0], z.shape[1]
        first, second = out[:B, :cin], out[B:, :cin]
        cond, unc = (first, second) if cond_first else (second, first)
        z.copy_(c_z * z + c_eps * (unc + guidance * (cond - unc)).reshape(z.shape))
        return z

    saved = ops.cfg_linear_step
    ops.cfg_linear_step = step
    try:
        yield
    finally:
        ops.cfg_linear_step = saved


class OracleLatte:
    """oracle.latte_oracle.LatteOracle behind LatteT2V's call interface (latents batch B, text batch B or 2 B)."""

    def __init__(self, fx):
        cfg = fx["cfg"]
        sd = LO.synth_state_dict(cfg["num_layers"], cfg["num_attention_heads"], cfg["attention_head_dim"],
                                 caption_channels=cfg["caption_channels"], seed=fx["seed"])
        sd = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
        self.o = LO.LatteOracle(sd, cfg["num_layers"], cfg["num_attention_heads"], cfg["attention_head_dim"],
                                sample_size=cfg["sample_size"], video_length=cfg["video_length"])
        self.in_channels = 4
        self.config = SimpleNamespace(caption_channels=cfg["caption_channels"], **{k: v for k, v in cfg.items() if k != "caption_channels"})
        self.parallel_manager = SimpleNamespace(dp_rank=0)
        self.calls, self.masks = 0, []

    def reset_pab_state(self):
        pass

    def reset_text_cache(self):
        pass

    def __call__(self, z, timestep=None, all_timesteps=None, encoder_hidden_states=None, encoder_attention_mask=None,
                 added_cond_kwargs=None, enable_temporal_attentions=True, return_dict=True):
        self.calls += 1
        self.masks.append(encoder_attention_mask)
        nb = encoder_hidden_states.shape[0]
        zin = z if z.shape[0] == nb else torch.cat([z] * (nb // z.shape[0]), 0)
        return (self.o(zin.float(), timestep, encoder_hidden_states.float(), encoder_attention_mask),)


def _latte_pipe(fx, **kw):
    from videosys_amd import LatteConfig, LattePipeline

    return LattePipeline(LatteConfig(model_path="maxin-cn/Latte-1"), transformer=OracleLatte(fx), device="cpu", **kw)


def test_latte_generate_host_flow_matches_the_reference_sampling_fixture():
    fx = load_golden("latte_sample_small.pt")
    with torch_step_kernel():
        pipe = _latte_pipe(fx)
        kw = dict(latents=fx["latents"], num_inference_steps=fx["steps"], guidance_scale=fx["guidance"], output_type="latent")
        out = pipe.generate(prompt_embeds=fx["pos"], negative_prompt_embeds=fx["neg"], prompt_mask=fx["pmask"],
                            negative_mask=fx["nmask"], **kw).video
        assert pipe.scheduler.timesteps == fx["timesteps"] and pipe.transformer.calls == fx["steps"]
        torch.testing.assert_close(out, fx["out"], rtol=5e-4, atol=5e-4)                  # the reference's own loop output
        # callback cadence, tuple return, the start latents of a caller's generator, check_inputs wired in
        seen = []
        tup = pipe.generate(prompt_embeds=fx["pos"], negative_prompt_embeds=fx["neg"], prompt_mask=fx["pmask"], negative_mask=fx["nmask"],
                            callback=lambda i, t, z: seen.append((i, t)), callback_steps=2, return_dict=False, **kw)
        assert isinstance(tup, tuple) and torch.equal(tup[0], out) and seen == [(i, fx["timesteps"][i]) for i in range(0, fx["steps"], 2)]
        kw2 = {k: v for k, v in kw.items() if k != "latents"}
        shp = dict(video_length=fx["latents"].shape[2], height=8 * fx["latents"].shape[3], width=8 * fx["latents"].shape[4])
        a = pipe.generate(prompt_embeds=fx["pos"], negative_prompt_embeds=fx["neg"], generator=torch.Generator().manual_seed(9), **shp, **kw2).video
        b = pipe.generate(prompt_embeds=fx["pos"], negative_prompt_embeds=fx["neg"], generator=torch.Generator().manual_seed(9), seed=5, **shp, **kw2).video
        c = pipe.generate(prompt_embeds=fx["pos"], negative_prompt_embeds=fx["neg"], seed=5, **shp, **kw2).video
        d = pipe.generate(prompt_embeds=fx["pos"], negative_prompt_embeds=fx["neg"], seed=5, **shp, **kw2).video
        assert torch.equal(a, b) and torch.equal(c, d) and not torch.equal(a, c)
        with pytest.raises(ValueError):
            pipe.generate(prompt_embeds=fx["pos"], negative_prompt_embeds=fx["neg"][:, :-1], **kw)
        with pytest.raises(ValueError):
            pipe.generate(prompt="a", prompt_embeds=fx["pos"], **kw)
        with pytest.raises(RuntimeError):     # a text prompt without a text encoder
            pipe.generate(prompt="a cat", **kw)


def test_latte_generate_text_prompts_on_cpu():
    """Text prompts: cleaned, encoded, cut to the prompt's token count (negative with it), no attention mask reaches the model."""
    fx = load_golden("latte_sample_small.pt")
    d = fx["cfg"]["caption_channels"]

    class Text:
        max_length = 16

        def __init__(self):
            self.asked = []

        def __call__(self, prompts):
            self.asked.append(list(prompts))
            emb, mask = [], torch.zeros(len(prompts), 16, dtype=torch.long)
            for b, q in enumerate(prompts):
                g = torch.Generator().manual_seed(sum(q.encode()) + 7)
                emb.append(torch.randn(1, 16, d, generator=g).to(torch.bfloat16))
                mask[b, :min(len(q.split()) + 1, 16)] = 1
            return torch.stack(emb, 0), mask

    with torch_step_kernel():
        te = Text()
        pipe = _latte_pipe(fx, text_encoder=te)
        kw = dict(latents=fx["latents"], num_inference_steps=2, guidance_scale=fx["guidance"], output_type="latent")
        a = pipe.generate(prompt="  A Red Cat <b>runs</b> ", negative_prompt="", **kw).video
        assert te.asked == [["a red cat runs"], [""]] and all(m is None for m in pipe.transformer.masks)
        pe, pm = Text()(["a red cat runs"])
        ne, _ = Text()([""])
        keep = int(pm.sum())
        b = pipe.generate(prompt_embeds=pe[:, 0, :keep], negative_prompt_embeds=ne[:, 0, :keep], **kw).video
        assert keep == 5 and torch.equal(a, b)
        c = pipe.generate(prompt="  A Red Cat <b>runs</b> ", negative_prompt="", clean_caption=False, **kw).video
        assert te.asked[-2] == ["a red cat <b>runs</b>"] and not torch.equal(a, c)


class OracleCog:
    """oracle.cogvideox_oracle.CogVideoXOracle behind CogVideoXTransformer3DModel's call interface."""

    def __init__(self, fx):
        cfg = fx["cfg"]
        sd = {k: v.to(torch.bfloat16).float() for k, v in CO.synth_state_dict(cfg["num_layers"], cfg["num_attention_heads"],
                                                                              text_embed_dim=cfg["text_embed_dim"], seed=fx["seed"]).items()}
        self.o = CO.CogVideoXOracle(sd, cfg["num_layers"], cfg["num_attention_heads"], max_text_seq_length=cfg["max_text_seq_length"],
                                    sample_width=cfg["sample_width"], sample_height=cfg["sample_height"],
                                    sample_frames=cfg["sample_frames"],
                                    use_rotary_positional_embeddings=cfg["use_rotary_positional_embeddings"])
        self.config = SimpleNamespace(in_channels=16, out_channels=16, patch_size=2, attention_head_dim=64, num_layers=cfg["num_layers"],
                                      text_embed_dim=cfg["text_embed_dim"],
                                      use_rotary_positional_embeddings=cfg["use_rotary_positional_embeddings"])
        self.parallel_manager = SimpleNamespace(dp_rank=0)
        self.calls = 0

    def reset_pab_state(self):
        pass

    def reset_text_cache(self):
        pass

    def __call__(self, z, emb, timestep, image_rotary_emb=None, return_dict=True):
        self.calls += 1
        nb = emb.shape[0]
        zin = z if z.shape[0] == nb else torch.cat([z] * (nb // z.shape[0]), 0)
        return (self.o(zin.float(), emb.float(), timestep, image_rotary_emb),)


def test_cogvideox_generate_host_flow_matches_the_reference_sampling_fixture():
    from videosys_amd import CogVideoXConfig, CogVideoXPipeline

    fx = load_golden("cogvideox_sample_small.pt")
    with torch_step_kernel():
        pipe = CogVideoXPipeline(CogVideoXConfig(model_path="THUDM/CogVideoX-5b"), transformer=OracleCog(fx), device="cpu")
        F = (fx["latents"].shape[1] - 1) * 4 + 1
        kw = dict(prompt_embeds=fx["pos"], latents=fx["latents"], height=8 * fx["latents"].shape[3], width=8 * fx["latents"].shape[4],
                  num_frames=F, num_inference_steps=fx["steps"], output_type="latent")
        out = pipe.generate(negative_prompt_embeds=fx["neg"], guidance_scale=fx["guidance"], use_dynamic_cfg=True, **kw).video
        assert pipe.scheduler.timesteps == fx["timesteps"] and pipe.num_timesteps == fx["steps"] and pipe.guidance_scale == fx["guidance"]
        torch.testing.assert_close(out, fx["out"], rtol=2e-2, atol=2e-2)                  # bf16 re-rounding of the latents every step
        # callbacks: the requested tensors arrive, returned latents replace the loop's, interrupt skips the remaining steps
        got = []

        def cb(p, i, t, tensors):
            got.append((i, sorted(tensors)))
            if i == 1:
                p._interrupt = True
            return {"latents": tensors["latents"]}

        n0 = pipe.transformer.calls
        pipe.generate(negative_prompt_embeds=fx["neg"], guidance_scale=fx["guidance"], callback_on_step_end=cb,
                      callback_on_step_end_tensor_inputs=["latents", "prompt_embeds"], **kw)
        assert got == [(0, ["latents", "prompt_embeds"]), (1, ["latents", "prompt_embeds"])] and pipe.transformer.calls - n0 == 2
        assert pipe.interrupt is True
        pipe.generate(guidance_scale=1.0, **kw)                                          # a new call starts un-interrupted
        assert pipe.interrupt is False
        with pytest.raises(ValueError):
            pipe.generate(negative_prompt_embeds=fx["neg"], callback_on_step_end_tensor_inputs=["frames"], **kw)
        with pytest.raises(NotImplementedError):
            pipe.generate(eta=0.3, **kw)


def test_cogvideox_generate_with_the_dpm_scheduler_host_flow():
    """CogVideoXPipeline.generate with a CogVideoXDPMScheduler handed in (pipeline_cogvideox.py:679-680,711-721): the loop keeps the
    previous step's x0 and timestep, draws the scheduler's noise from the caller's generator in the reference's order (two draws on a
    second-order step, the second one used) and expresses x0 and the update through the fused guidance + linear-step kernel.  Checked
    on CPU (kernel emulated, oracle transformer) against the direct formulas of scheduling_dpm_cogvideox.py:402-447 driven by the same
    generator — the scheduler class itself is pinned against the reference's in tests/test_cogvideox_cpu.py."""
    from videosys_amd import CogVideoXConfig, CogVideoXPipeline
    from videosys_amd.pipeline_cogvideox import CogVideoXDPMScheduler

    fx = load_golden("cogvideox_sample_small.pt")
    skw = dict(snr_shift_scale=1.0)
    steps, guidance = 5, 6.0
    F = (fx["latents"].shape[1] - 1) * 4 + 1
    with torch_step_kernel():
        model = OracleCog(fx)
        pipe = CogVideoXPipeline(CogVideoXConfig(model_path="THUDM/CogVideoX-5b"), transformer=model, scheduler=CogVideoXDPMScheduler(**skw),
                                 device="cpu")
        assert isinstance(pipe.scheduler, CogVideoXDPMScheduler)
        out = pipe.generate(prompt_embeds=fx["pos"], negative_prompt_embeds=fx["neg"], latents=fx["latents"], height=8 * fx["latents"].shape[3],
                            width=8 * fx["latents"].shape[4], num_frames=F, num_inference_steps=steps, guidance_scale=guidance,
                            generator=torch.Generator().manual_seed(77), output_type="latent").video
        # the same loop written out with the scheduler's multipliers
        sched = CogVideoXDPMScheduler(**skw)
        sched.set_timesteps(steps)
        g = torch.Generator().manual_seed(77)
        z = fx["latents"].float().clone()
        emb = torch.cat([fx["neg"], fx["pos"]], 0)
        rope = pipe._prepare_rotary_positional_embeddings(8 * z.shape[3], 8 * z.shape[4], z.shape[1]) \
            if model.config.use_rotary_positional_embeddings else None
        x0_old, t_back = None, None
        for t in sched.timesteps:
            o = model(z, emb, torch.full((2 * z.shape[0],), t, dtype=torch.int64), image_rotary_emb=rope)[0]
            u, c = o.chunk(2)
            v = u + guidance * (c - u)
            sa, sb, m1, m2, m3, m4, mn, second = sched.multipliers(t, t_back)
            x0 = sa * z - sb * v
            n = torch.randn(z.shape, generator=g, dtype=torch.bfloat16).float()
            if x0_old is None or not second:
                z = m1 * z - m2 * x0 + mn * n
            else:
                n = torch.randn(z.shape, generator=g, dtype=torch.bfloat16).float()
                z = m1 * z - m2 * (m3 * x0 - m4 * x0_old) + mn * n
            z = z.to(torch.bfloat16).float()
            x0_old, t_back = x0, t
    assert torch.isfinite(out).all()
    torch.testing.assert_close(out, z, rtol=2e-2, atol=2e-2)
    assert (out - fx["latents"].float()).abs().max().item() > 0.1


def test_component_names_typo_paths_raise_hub_ids_warn(caplog):
    """config.transformer / config.vae: a Hugging Face hub id (the reference's defaults) cannot be fetched offline and is replaced with
    a logged warning; a string that looks like a filesystem path and does not exist is a typo and raises instead of silently
    producing video from random weights."""
    import logging

    from videosys_amd.pipeline_open_sora import OpenSoraPipeline

    for bad in ("/no/such/dir", "./ckpts/stdit3", "~/models/x", "checkpoints", "a/b/c"):
        with pytest.raises(FileNotFoundError):
            OpenSoraPipeline._hub_fallback(bad, "transformer", "synthetic:1234")
    with caplog.at_level(logging.WARNING, logger="videosys_amd"):
        OpenSoraPipeline._hub_fallback("hpcai-tech/OpenSora-STDiT-v3", "transformer", "synthetic:1234")
    assert any("hub id" in r.getMessage() for r in caplog.records)
    # "ckpts/stdit3" has the shape of a hub id, but with a directory ckpts/ beside the caller it is a mistyped relative path
    import os
    import tempfile

    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.mkdir(os.path.join(d, "ckpts"))
        os.chdir(d)
        try:
            with pytest.raises(FileNotFoundError):
                OpenSoraPipeline._hub_fallback("ckpts/stdit3", "transformer", "synthetic:1234")
        finally:
            os.chdir(cwd)
