"""CPU: the restatement of diffusers' AutoencoderKLTemporalDecoder decode path (oracle/svd_vae_oracle.py — an UNPINNED third-party
leaf: diffusers is absent here and the reference has no test for it) is at least self-consistent, and the product's parameter
table and its algebraic shortcuts agree with it."""
import torch

from oracle import svd_vae_oracle as SO


def test_param_tables_agree_and_decode_shapes():
    from videosys_amd.vae_svd_temporal import synth_state_dict, temporal_decoder_param_shapes

    assert temporal_decoder_param_shapes() == SO.param_shapes()
    a, b = synth_state_dict(3), SO.synth_state_dict(3)
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
    n_params = sum(v.numel() for v in a.values())
    assert 55e6 < n_params < 75e6, n_params      # decode side of the ~ 98 M-parameter checkpoint (its encoder is the 34 M SD encoder)


def test_alpha_blend_folding_and_symmetric_time_conv():
    """out = (1 - s) x_s + s (x_s + h) == x_s + s h (what the product folds into conv2), and a (3, 1, 1) Conv3d with padding
    (1, 0, 0) == the CAUSAL 3-tap convolution over [0, x_0 .. x_{F-1}, 0] read one plane later (how the product runs it)."""
    g = torch.Generator().manual_seed(0)
    xs, h = torch.randn(2, 8, 5, 4, 4, generator=g), torch.randn(2, 8, 5, 4, 4, generator=g)
    s = torch.sigmoid(torch.tensor(0.37))
    alpha = 1.0 - s
    torch.testing.assert_close(alpha * xs + (1.0 - alpha) * (xs + h), xs + s * h)
    w = torch.randn(8, 8, 3, 1, 1, generator=g)
    want = torch.nn.functional.conv3d(xs, w, padding=(1, 0, 0))
    padded = torch.cat([torch.zeros(2, 8, 1, 4, 4), xs, torch.zeros(2, 8, 1, 4, 4)], dim=2)   # planes 0 .. F+1
    got = sum(torch.einsum("oc,bcfhw->bofhw", w[:, :, k, 0, 0], padded[:, :, k:k + 5]) for k in range(3))
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


def test_oracle_decode_small_latent():
    sd = SO.synth_state_dict(1)
    lat = torch.randn(1, 4, 3, 4, 4, generator=torch.Generator().manual_seed(2)) * 0.18215
    v = SO.decode_latents_with_temporal_decoder(lat, sd, decode_chunk_size=2)
    assert v.dtype == torch.uint8 and tuple(v.shape) == (1, 3, 32, 32, 3)
    # chunking matters (GroupNorm / time convs see the frames of a chunk only): 2 + 1 frames differ from 3 at once
    a = SO.decode_latents_with_temporal_decoder(lat, sd, decode_chunk_size=2, as_uint8=False)
    b = SO.decode_latents_with_temporal_decoder(lat, sd, decode_chunk_size=14, as_uint8=False)
    assert (a - b).abs().max() > 1e-3
