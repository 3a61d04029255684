"""TEST INFRASTRUCTURE ONLY — mints tests/golden/opensora_vae_small.pt from the reference's VideoAutoencoderPipeline.

    python oracle/make_golden_vae.py       (build container only: needs /root/reference)

The reference's own autoencoder_kl_open_sora.py (VideoAutoencoderPipeline.decode, VAE_Temporal, Decoder, ResBlock,
CausalConv3d, VideoAutoencoderKL micro-batching) runs on CPU over the restated diffusers==0.30.0 AutoencoderKL decoder of
oracle/diffusers_stub.py, at the REAL architecture (VAE_Temporal_SD + SDXL-VAE channel counts) on a small latent:
z [1, 4, 7, 12, 8] -> 22 frames of 96 x 64 (two temporal micro-batches: 5 + 2 latent frames, 17 + 5 output frames, both with a
3-frame time padding to drop; 96 tokens in the mid-block attention, i.e. not a multiple of the 128-column GEMM tile).
Weights are videosys_amd.vae_open_sora.synth_state_dict(seed) (bf16-representable), so the fixture holds only tensors of a
few MB: the latent, the fp32 reference video, the reference's bf16 run (the noise floor a bf16 implementation is judged
against) and the temporal-VAE intermediate.
"""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_loader  # noqa: E402
from oracle import vae_oracle as VO  # noqa: E402
from oracle.make_golden import OUT, bf16r, sd_checksum  # noqa: E402
from videosys_amd.vae_open_sora import synth_state_dict  # noqa: E402

VAE_SEED = 7
NUM_FRAMES = 22


def inputs(seed=11):
    g = torch.Generator().manual_seed(seed)
    return bf16r(torch.randn(1, 4, 7, 12, 8, generator=g))


def main():
    sd = synth_state_dict(VAE_SEED)
    z = inputs()
    model = ref_loader.build_reference_opensora_vae(sd)
    with torch.no_grad():
        ref = model.decode(z, num_frames=NUM_FRAMES)
        x_z = model.temporal_vae.decode(z[:, :, :5] * model.scale + model.shift, num_frames=17)
    mine = VO.decode(sd, z, NUM_FRAMES)
    err = (mine - ref).abs().max().item()
    print("reference fp32", tuple(ref.shape), "abs mean", ref.abs().mean().item(), "restatement max abs diff", err)
    assert err < 1e-4, err
    model16 = ref_loader.build_reference_opensora_vae(sd, dtype=torch.bfloat16)
    with torch.no_grad():
        ref16 = model16.decode(z.to(torch.bfloat16), num_frames=NUM_FRAMES)
    d16 = (ref16.float() - ref)
    print("reference bf16 vs fp32: max abs", d16.abs().max().item(), "rel rms", (d16.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item())
    torch.save({"z": z, "num_frames": NUM_FRAMES, "seed": VAE_SEED, "out_fp32": ref.to(torch.float16), "out_bf16": ref16,
                "x_z_first": x_z, "sd_checksum": sd_checksum(sd)}, os.path.join(OUT, "opensora_vae_small.pt"))
    print("wrote", os.path.join(OUT, "opensora_vae_small.pt"))


if __name__ == "__main__":
    main()
