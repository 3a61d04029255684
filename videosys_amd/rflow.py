"""Rectified-flow sampler — host mirror of videosys/schedulers/scheduling_rflow_open_sora.py (RFLOW :164-257,
timestep_transform :47-70).  Same constructor kwargs and ``sample(model, z, model_args, y_null, device, ...)``
signature; the per-step CFG combine + Euler update is one HIP kernel (vsys_cfg_euler_step) and the timestep
schedule stays on the host so nothing in the loop syncs the device (the reference calls ``t.item()`` per step, :222).
"""
from __future__ import annotations

import torch

from . import ops
from .utils import progress_wrap


def timestep_transform(t, model_kwargs, base_resolution=512 * 512, base_num_frames=1, scale=1.0, num_timesteps=1):
    """Resolution / duration dependent time warp (scheduling_rflow_open_sora.py:47-70), on host tensors.

    With u = t / num_timesteps and r = scale * sqrt(H*W / base_resolution) * sqrt(F' / base_num_frames), where F' = 1 for an
    image and (num_frames // 17) * 5 latent frames for a video, the warped time is u' = r u / (1 + (r - 1) u)."""
    frames = model_kwargs["num_frames"]
    latent_frames = torch.ones_like(frames) if frames[0] == 1 else frames // 17 * 5
    r = scale * (model_kwargs["height"] * model_kwargs["width"] / base_resolution).sqrt() * (latent_frames / base_num_frames).sqrt()
    u = t / num_timesteps
    return num_timesteps * (r * u / (1 + (r - 1) * u))


class RFLOW:
    def __init__(self, num_sampling_steps=10, num_timesteps=1000, cfg_scale=4.0, use_discrete_timesteps=False,
                 use_timestep_transform=False, **kwargs):
        self.num_sampling_steps = num_sampling_steps
        self.num_timesteps = num_timesteps
        self.cfg_scale = cfg_scale
        self.use_discrete_timesteps = use_discrete_timesteps
        self.use_timestep_transform = use_timestep_transform

    def prepare_timesteps(self, batch, model_args):
        """:208-213 — host tensors [batch] per step."""
        n, full = self.num_sampling_steps, self.num_timesteps
        grid = [(1.0 - k / n) * full for k in range(n)]          # 1000, 1000 (1 - 1/n), ... (uniform in flow time)
        if self.use_discrete_timesteps:
            grid = [int(round(v)) for v in grid]
        steps = [torch.full((batch,), float(v), dtype=torch.float32) for v in grid]
        if not self.use_timestep_transform:
            return steps
        # the geometry tensors keep the dtype they arrive in (bf16 from the pipeline, as in the reference): the ratio is then
        # computed in that dtype and promoted by the fp32 timestep exactly as scheduling_rflow_open_sora.py:47-70 does
        geom = {k: model_args[k].detach().to("cpu") for k in ("height", "width", "num_frames")}
        return [timestep_transform(v, geom, num_timesteps=full) for v in steps]

    @torch.no_grad()
    def sample(self, model, z, model_args, y_null, device=None, mask=None, guidance_scale=None, progress=True,
               verbose=False, noise_fn=None):
        """``mask`` [B, T] (float; open_sora_condition.apply_mask_strategy) conditions the sampling on frames already present in
        ``z`` (:215-236,254-255): a frame is denoised only from the step on at which mask * num_timesteps >= t; until then the
        model sees it with the timestep-0 modulation (``x_mask``) and every update puts it back; at the step it joins it is
        noised to that step's level, once.  The mask and the per-step decisions live on the host (B*T values), only the boolean
        frame masks travel.  ``noise_fn(shape)`` draws that noise (fp32, host or device); default: torch.randn on the global
        CPU generator, one draw per step like the reference's randn_like, so a seed reproduces a run."""
        if guidance_scale is None:
            guidance_scale = self.cfg_scale
        model_args = dict(model_args)
        model_args["y"] = torch.cat([model_args["y"], y_null.to(model_args["y"].dtype).to(model_args["y"].device)], 0)
        B = z.shape[0]
        timesteps = self.prepare_timesteps(B, model_args)
        dtype = model.x_embedder.proj.weight.dtype
        model_args["all_timesteps"] = [int(t.to(dtype)[0]) for t in timesteps]
        fwd_args = {k: v for k, v in model_args.items() if k != "num_frames"}
        for k in ("height", "width", "fps"):  # model is called on the CFG-doubled batch
            if k in fwd_args and fwd_args[k] is not None and fwd_args[k].shape[0] == B:
                fwd_args[k] = torch.cat([fwd_args[k], fwd_args[k]], 0)
        z = z.to(device=model.device, dtype=torch.float32).contiguous().clone()
        if mask is not None:
            cond = mask.detach().to("cpu", torch.float32)
            if tuple(cond.shape) != (B, z.shape[2]):
                raise ValueError(f"mask must be [B, T] = [{B}, {z.shape[2]}], got {tuple(cond.shape)}")
            noise_added = cond == 1
            frames = lambda m: m.to(z.device)[:, None, :, None, None]
        for i, t in progress_wrap(list(enumerate(timesteps)), progress):   # (:219,224) tqdm on rank 0
            if mask is not None:
                x0 = z.clone()
                upper = (cond * self.num_timesteps) >= t.unsqueeze(1)            # frames that are being denoised at this step
                noise = noise_fn(x0.shape) if noise_fn is not None else torch.randn(x0.shape, dtype=torch.float32)
                joining = upper & ~noise_added
                if bool(joining.any()):
                    keep = (1 - t.float() / self.num_timesteps).to(z.device)[:, None, None, None, None]   # add_noise, :144-161
                    z = torch.where(frames(joining), keep * x0 + (1 - keep) * noise.to(z.device, torch.float32), x0)
                noise_added = upper
                # (an all-True mask selects the modulation of t for every frame — the plain step, which replays its launch program)
                fwd_args["x_mask"] = None if bool(upper.all()) else upper.repeat(2, 1)
            z_in = torch.cat([z, z], 0)
            tt = torch.cat([t, t], 0)
            out = model(z_in, tt, **fwd_args)
            dt = timesteps[i] - timesteps[i + 1] if i < len(timesteps) - 1 else timesteps[i]
            if not bool((dt == dt[0]).all()):   # the reference applies dt per sample (:250); one geometry per call here
                raise NotImplementedError("per-sample step sizes (mixed geometries in one batch) are not supported")
            dt = float(dt[0]) / self.num_timesteps
            ops.cfg_euler_step(z, out, guidance_scale, dt)
            if mask is not None and not bool(upper.all()):
                z = torch.where(frames(upper), z, x0)
        return z
