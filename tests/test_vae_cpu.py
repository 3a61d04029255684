"""CPU: the VAE-decode oracle (oracle/vae_oracle.py) against the golden minted from the reference's VideoAutoencoderPipeline
(oracle/make_golden_vae.py), the decode-side parameter inventory against the reference's own state_dict (when
/root/reference is present), and the host-side size helpers against the values SURVEY.md §8d derives from the reference."""
import os

import pytest
import torch

from conftest import load_golden


@pytest.fixture(scope="module")
def gold():
    return load_golden("opensora_vae_small.pt")


def test_vae_oracle_matches_reference_golden(gold):
    from oracle import vae_oracle as VO
    from oracle.make_golden import sd_checksum
    from videosys_amd.vae_open_sora import synth_state_dict

    sd = synth_state_dict(gold["seed"])
    assert sd_checksum(sd) == gold["sd_checksum"]
    out = VO.decode(sd, gold["z"], gold["num_frames"])
    ref = gold["out_fp32"].float()
    assert out.shape == ref.shape == (1, 3, 22, 96, 64)
    # the fixture stores the fp32 reference rounded to fp16 (|x| < 4: 2^-10 absolute)
    assert (out - ref).abs().max().item() < 3e-3
    xz = VO.temporal_decode(sd, gold["z"][:, :, :5] * torch.tensor(VO.SCALE)[None, :, None, None, None]
                            + torch.tensor(VO.SHIFT)[None, :, None, None, None], 17)
    assert xz.shape == gold["x_z_first"].shape == (1, 4, 17, 12, 8)
    assert (xz - gold["x_z_first"]).abs().max().item() < 1e-4


def test_decoder_param_inventory_matches_reference():
    from oracle import ref_loader

    if not ref_loader.reference_available():
        pytest.skip("reference tree not present on this box")
    from videosys_amd.vae_open_sora import decoder_param_shapes

    model = ref_loader.build_reference_opensora_vae()
    ref = {k: tuple(v.shape) for k, v in model.state_dict().items()
           if "encoder" not in k and ".quant_conv" not in k and k not in ("scale", "shift")}
    assert decoder_param_shapes() == ref


def test_latent_size_helpers():
    """get_latent_size of VideoAutoencoderPipeline (autoencoder_kl_open_sora.py:704-716): 64 x 512 x 512 -> [19, 64, 64];
    128 x 720 x 1280 -> [38, 90, 160]."""
    from videosys_amd.vae_open_sora import OpenSoraVAE

    v = OpenSoraVAE.__new__(OpenSoraVAE)
    v.micro_frame_size = 17
    assert v.get_latent_size((64, 512, 512)) == [19, 64, 64]
    assert v.get_latent_size((128, 720, 1280)) == [38, 90, 160]
    assert v.get_latent_size((17, 256, 256)) == [5, 32, 32]
    assert v.get_latent_size((1, 512, 512)) == [1, 64, 64]


def test_vae_needs_gpu():
    from videosys_amd.vae_open_sora import OpenSoraVAE

    with pytest.raises(RuntimeError):
        OpenSoraVAE({}, device="cpu")


def test_vae_oracle_matches_live_reference_class():
    """oracle/vae_oracle.py against the reference's VideoAutoencoderPipeline run here (another seed, another latent size, three
    micro-batches incl. a 1-latent-frame tail) when /root/reference is present."""
    from oracle import ref_loader

    if not ref_loader.reference_available():
        pytest.skip("reference tree not present on this box")
    from oracle import vae_oracle as VO
    from videosys_amd.vae_open_sora import synth_state_dict

    sd = synth_state_dict(21)
    model = ref_loader.build_reference_opensora_vae(sd)
    g = torch.Generator().manual_seed(4)
    z = torch.randn(1, 4, 11, 4, 6, generator=g)       # 5 + 5 + 1 latent frames -> 17 + 17 + 4 = 38 frames
    with torch.no_grad():
        ref = model.decode(z, num_frames=38)
    out = VO.decode(sd, z, 38)
    assert out.shape == ref.shape == (1, 3, 38, 32, 48)
    assert (out - ref).abs().max().item() < 1e-4
