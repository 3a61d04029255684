"""GPU, OPT-IN (VSYS_TEST_LAB=1): kernels that exist as lab variants but have not been validated on hardware yet.  They are
never dispatched by default, so these tests are skipped in the normal `-m gpu` run and must be green before a variant is adopted.

    VSYS_TEST_LAB=1 python -m pytest tests/test_gpu_lab.py -q
"""
import os

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("VSYS_TEST_LAB") != "1", reason="lab kernels: set VSYS_TEST_LAB=1")]


def test_temporal_attention_mfma_variant_matches_shipped_kernel_and_oracle():
    """flash variant 7 = attn_temporal_d72 on the matrix pipe (csrc/attention_t_mfma.hip): same result as the shipped kernel within
    bf16 rounding of the output (both are fp32-softmax formulations), with and without qk-norm / RoPE, T = 19, 5, 32, 1."""
    from oracle import stdit3_oracle as O
    from videosys_amd import _lib, ops

    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(33)
    for B, T, S, H, norm, rope in ((2, 19, 64, 16, True, True), (1, 5, 33, 3, True, False), (1, 32, 16, 4, False, True), (1, 1, 8, 2, True, True)):
        C = H * 72
        qkv = torch.randn(B * T * S, 3 * C, generator=g).to(torch.bfloat16).to(dev)
        qw = (1 + 0.1 * torch.randn(72, generator=g)).to(torch.bfloat16).to(dev) if norm else None
        kw = (1 + 0.1 * torch.randn(72, generator=g)).to(torch.bfloat16).to(dev) if norm else None
        cos = sin = None
        if rope:
            freqs = 1.0 / (10000 ** (torch.arange(0, 72, 2).float() / 72))
            cos, sin = (t.contiguous().to(dev) for t in O.rope_table(freqs, T))
        want = torch.empty(B * T * S, C, dtype=torch.bfloat16, device=dev)
        ops.attn_temporal(qkv, C, qw, kw, cos, sin, want, B, T, S, H)
        got = torch.full_like(want, 7.0)
        lib.vsys_tune_flash_variant(7)
        try:
            ops.attn_temporal(qkv, C, qw, kw, cos, sin, got, B, T, S, H)
        finally:
            lib.vsys_tune_flash_variant(0)
        torch.cuda.synchronize()
        err = (got.float() - want.float()).abs().max().item()
        assert err <= 2 ** -6 * want.float().abs().max().item(), (B, T, S, H, norm, rope, err)
