"""TEST INFRASTRUCTURE ONLY — mints tests/golden/opensora_vae_encode_small.pt from the reference's VideoAutoencoderPipeline.

    python oracle/make_golden_vae_encode.py       (build container only: needs /root/reference)

VideoAutoencoderPipeline.encode (autoencoder_kl_open_sora.py:653-670): the reference's own VAE_Temporal encoder (Encoder :177-272,
strided CausalConv3d :107-125, DiagonalGaussianDistribution :21-40) and VideoAutoencoderKL.encode micro-batching (:503-520) run on
CPU over the restated diffusers AutoencoderKL encoder of oracle/diffusers_stub.py (third-party leaf: parity unpinned), at the REAL
channel counts on a small clip: 21 frames of 32 x 48 pixels -> two temporal micro batches (17 + 4 frames -> 5 + 1 latent frames,
both with zero frames padded in front).  The posteriors' noise comes from the global CPU generator seeded with NOISE_SEED: six
draws [<=4, 4, 4, 6] for the 2-D micro batches, then [1, 4, 5, 4, 6] and [1, 4, 1, 4, 6] — a consumer that seeds the generator
and draws the same shapes in the same order sees the same noise.
"""
from __future__ import annotations

import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_loader  # noqa: E402
from oracle import vae_oracle as VO  # noqa: E402
from oracle.make_golden import OUT, bf16r, sd_checksum  # noqa: E402
from videosys_amd.vae_open_sora import synth_state_dict  # noqa: E402

VAE_SEED = 7
NOISE_SEED = 1234


def main():
    sd = synth_state_dict(VAE_SEED, encoder=True)
    g = torch.Generator().manual_seed(13)
    x = bf16r((torch.randn(1, 3, 21, 32, 48, generator=g) * 0.5).clamp(-1, 1))
    model = ref_loader.build_reference_opensora_vae(sd)
    with torch.no_grad():
        torch.manual_seed(NOISE_SEED)
        ref = model.encode(x)
        fr = x[0].permute(1, 0, 2, 3)
        m2d = model.spatial_vae.module.quant_conv(model.spatial_vae.module.encoder(fr[:4]))
        xz = bf16r(torch.randn(1, 4, 17, 4, 6, generator=g) * 0.2)
        mt = model.temporal_vae.quant_conv(model.temporal_vae.encoder(F.pad(xz, (0, 0, 0, 0, 3, 0))))
    torch.manual_seed(NOISE_SEED)
    mine = VO.encode(sd, x)
    err = (mine - ref).abs().max().item()
    print("reference fp32", tuple(ref.shape), "abs mean", ref.abs().mean().item(), "restatement max abs diff", err)
    assert err < 1e-4, err
    assert (VO.spatial_encode_moments(sd, fr[:4]) - m2d).abs().max().item() < 1e-4
    assert (VO.temporal_encode_moments(sd, xz) - mt).abs().max().item() < 1e-4
    model16 = ref_loader.build_reference_opensora_vae(sd, dtype=torch.bfloat16)
    with torch.no_grad():
        m2d16 = model16.spatial_vae.module.quant_conv(model16.spatial_vae.module.encoder(fr[:4].to(torch.bfloat16)))
        mt16 = model16.temporal_vae.quant_conv(model16.temporal_vae.encoder(F.pad(xz, (0, 0, 0, 0, 3, 0)).to(torch.bfloat16)))
    rel = lambda a, b: ((a.float() - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
    # (the sampled latents of a bf16 run are not comparable: bf16 posteriors draw their noise in bf16, another random stream —
    # the deterministic halves, the two encoders' moments, carry the bf16 floor)
    print("reference bf16 vs fp32 rel rms: 2-D moments", rel(m2d16, m2d), "temporal moments", rel(mt16, mt))
    torch.save({"x": x, "seed": VAE_SEED, "noise_seed": NOISE_SEED, "z_fp32": ref, "frames4_moments": m2d,
                "frames4_moments_bf16": m2d16, "xz17": xz, "xz17_moments": mt, "xz17_moments_bf16": mt16,
                "sd_checksum": sd_checksum(sd)}, os.path.join(OUT, "opensora_vae_encode_small.pt"))
    print("wrote", os.path.join(OUT, "opensora_vae_encode_small.pt"))


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
