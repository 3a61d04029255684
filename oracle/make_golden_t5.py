"""TEST INFRASTRUCTURE ONLY — mints tests/golden/t5_small.pt from the REAL transformers.T5EncoderModel (the third-party class the
reference pipelines instantiate, pipeline_open_sora.py:211-214), random weights from videosys_amd.t5.synth_state_dict.

    python oracle/make_golden_t5.py
"""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import t5_oracle as TO  # noqa: E402
from oracle.make_golden import OUT  # noqa: E402
from videosys_amd.t5 import synth_state_dict  # noqa: E402

CFG = dict(d_model=256, d_ff=512, num_layers=3, num_heads=4, vocab_size=200)
SEED = 5


def hf_model(sd):
    from transformers import T5Config, T5EncoderModel

    cfg = T5Config(vocab_size=CFG["vocab_size"], d_model=CFG["d_model"], d_kv=64, d_ff=CFG["d_ff"], num_layers=CFG["num_layers"],
                   num_heads=CFG["num_heads"], relative_attention_num_buckets=32, relative_attention_max_distance=128,
                   feed_forward_proj="gated-gelu", layer_norm_epsilon=1e-6, dropout_rate=0.0)
    m = T5EncoderModel(cfg).eval()
    full = dict(sd)
    full["encoder.embed_tokens.weight"] = sd["shared.weight"]
    missing, unexpected = m.load_state_dict(full, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    return m


def inputs():
    g = torch.Generator().manual_seed(21)
    ids = torch.randint(0, CFG["vocab_size"], (2, 150), generator=g)
    mask = torch.ones(2, 150, dtype=torch.long)
    mask[1, 97:] = 0
    ids[1, 97:] = 0
    return ids, mask


def main():
    sd = synth_state_dict(seed=SEED, **CFG)
    ids, mask = inputs()
    m = hf_model(sd)
    with torch.no_grad():
        ref = m(input_ids=ids, attention_mask=mask)["last_hidden_state"]
    mine = TO.encode(sd, ids, mask, CFG["num_layers"], CFG["num_heads"])
    err = (mine - ref).abs().max().item()
    print("transformers T5EncoderModel fp32", tuple(ref.shape), "abs mean", ref.abs().mean().item(), "restatement max abs diff", err)
    assert err < 1e-4
    m16 = hf_model(sd).to(torch.bfloat16)
    with torch.no_grad():
        ref16 = m16(input_ids=ids, attention_mask=mask)["last_hidden_state"]
    d = ref16.float() - ref
    print("bf16 vs fp32: max", d.abs().max().item(), "rel rms", (d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item())
    torch.save({"cfg": CFG, "seed": SEED, "ids": ids, "mask": mask, "out_fp32": ref, "out_bf16": ref16},
               os.path.join(OUT, "t5_small.pt"))
    print("wrote t5_small.pt")


if __name__ == "__main__":
    main()
