"""videosys/models/modules/normalization.py mirror — the normalisation operator of the hot path that the reference exposes (and
tests: tests/test_rms_norm.py) as a stand-alone module.  Inside the transformers of this build the q / k RMS norm is fused into the
attention prep kernels (vsys_attn_prep_kv); this class is the same arithmetic as one launch for code that calls it directly."""
from __future__ import annotations

import torch

from . import ops


class LlamaRMSNorm:
    """normalization.py:19-33 (``equivalent to T5LayerNorm``): ``weight * bf16(x * rsqrt(mean(x^2) + eps))`` over the last
    dimension, fp32 statistics.  bf16 tensors on the HIP device, last dimension = ``hidden_size`` (a multiple of 8, <= 8192);
    forward only."""

    def __init__(self, hidden_size, eps=1e-6, device="cuda"):
        self.weight = torch.ones(hidden_size, dtype=torch.bfloat16, device=device)
        self.variance_epsilon = eps

    def load_state_dict(self, sd, strict: bool = True):
        self.weight = sd["weight"].to(self.weight.device, torch.bfloat16).contiguous()
        return self

    def state_dict(self):
        return {"weight": self.weight}

    def to(self, device=None, dtype=None):
        if dtype not in (None, torch.bfloat16):
            raise NotImplementedError("LlamaRMSNorm computes in bf16 with fp32 statistics")
        if device is not None:
            self.weight = self.weight.to(device)
        return self

    def eval(self):
        return self

    def forward(self, hidden_states):
        if hidden_states.dtype != torch.bfloat16 or not hidden_states.is_cuda:
            raise RuntimeError("LlamaRMSNorm needs a bf16 tensor on the HIP device (videosys_amd has no CPU execution path)")
        C = hidden_states.shape[-1]
        if C != self.weight.numel():
            raise ValueError(f"last dimension {C} != hidden_size {self.weight.numel()}")
        y = ops.rms_norm_rows(hidden_states.reshape(-1, C).contiguous(), self.weight, self.variance_epsilon)
        return y.view(hidden_states.shape)

    __call__ = forward


def get_rms_norm():
    """normalization.py:9-16: the RMS norm class the models use (the reference prefers apex's fused kernel when installed; here
    the HIP kernel is the only one)."""
    return LlamaRMSNorm
