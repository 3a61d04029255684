"""CPU: the T5-encoder oracle against the golden minted from the real transformers.T5EncoderModel (oracle/make_golden_t5.py) and
against the live class (transformers is part of the image, here and on the GPU box); the host-side bucket function."""
import pytest
import torch

from conftest import load_golden


def test_t5_oracle_matches_transformers_golden():
    from oracle import t5_oracle as TO
    from videosys_amd.t5 import synth_state_dict

    gold = load_golden("t5_small.pt")
    cfg = gold["cfg"]
    sd = synth_state_dict(seed=gold["seed"], **cfg)
    out = TO.encode(sd, gold["ids"], gold["mask"], cfg["num_layers"], cfg["num_heads"])
    assert (out - gold["out_fp32"]).abs().max().item() < 1e-4


def test_t5_oracle_matches_live_transformers():
    transformers = pytest.importorskip("transformers")
    from oracle import make_golden_t5 as MG
    from oracle import t5_oracle as TO
    from videosys_amd.t5 import synth_state_dict

    sd = synth_state_dict(seed=9, **MG.CFG)
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, MG.CFG["vocab_size"], (1, 40), generator=g)
    mask = torch.ones(1, 40, dtype=torch.long)
    mask[0, 33:] = 0
    with torch.no_grad():
        ref = MG.hf_model(sd)(input_ids=ids, attention_mask=mask)["last_hidden_state"]
    out = TO.encode(sd, ids, mask, MG.CFG["num_layers"], MG.CFG["num_heads"])
    assert (out - ref).abs().max().item() < 1e-4


def test_relative_position_bucket_matches_transformers():
    pytest.importorskip("transformers")
    from transformers.models.t5.modeling_t5 import T5Attention

    from videosys_amd.t5 import relative_position_bucket

    rel = torch.arange(-511, 512, dtype=torch.long)
    want = T5Attention._relative_position_bucket(rel, bidirectional=True, num_buckets=32, max_distance=128)
    assert torch.equal(relative_position_bucket(rel, 32, 128), want)


def test_t5_needs_gpu():
    from videosys_amd.t5 import T5Encoder

    with pytest.raises(RuntimeError):
        T5Encoder(device="cpu")


def test_relative_bias_table_matches_transformers_compute_bias():
    """The [heads, 2L-1] table the attention kernel indexes with j - i holds exactly T5Attention.compute_bias's [1, H, L, L]."""
    pytest.importorskip("transformers")
    from oracle import make_golden_t5 as MG
    from videosys_amd.t5 import relative_bias_table, synth_state_dict

    sd = synth_state_dict(seed=2, **MG.CFG)
    m = MG.hf_model(sd)
    att = m.encoder.block[0].layer[0].SelfAttention
    L = 77
    with torch.no_grad():
        want = att.compute_bias(L, L)[0]                                    # [H, L, L]
    tab = relative_bias_table(sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"], L)
    idx = (torch.arange(L)[None, :] - torch.arange(L)[:, None]) + L - 1      # j - i + L - 1
    assert torch.equal(tab[:, idx], want)


def test_skinny_split_rule_and_padded_bias_table():
    """Host side of the weight-streaming T5 path: the K-slice rule (256 x 384 tile form: panels x slices within one round of 256
    workgroups, slices of >= 512 columns, K need not divide; 128-column form: the measured rule) and the padded relative-position
    table of the matrix-pipe attention (every index a tile can form stays inside; the real entries equal the plain table)."""
    import math

    from videosys_amd.ops import skinny_split
    from videosys_amd.t5 import padded_bias_table, relative_bias_table

    # T5-XXL at 300 tokens (rows padded to 384): qkv, o, wi, wo
    assert [skinny_split(n, 384, k) for n, k in ((12288, 4096), (4096, 4096), (20480, 4096), (4096, 10240))] == [5, 8, 3, 16]
    for n, k in ((12288, 4096), (4096, 4096), (20480, 4096), (4096, 10240), (1024, 512), (256, 64)):
        s = skinny_split(n, 384, k)
        ks = -(-(k // 32) // s) * 32
        assert s >= 1 and ks * (s - 1) < k <= ks * s and (s == 1 or ks >= 512) and ((n + 255) // 256) * s <= max(256, (n + 255) // 256)
    assert [skinny_split(n, 384, k, wide=False) for n, k in ((12288, 4096), (4096, 4096), (20480, 4096), (4096, 10240))] == [1, 4, 1, 4]
    g = torch.Generator().manual_seed(0)
    for L, H in ((300, 4), (77, 2), (128, 3), (1, 2)):
        rel = torch.randn(32, H, generator=g)
        t, center = padded_bias_table(rel, L)
        qpad, kpad = (L + 127) // 128 * 128, (L + 63) // 64 * 64
        assert center >= qpad - 1 and t.shape == (H, center + kpad)        # min index center - (qpad - 1) >= 0, max center + kpad - 1
        plain = relative_bias_table(rel, L)
        assert torch.allclose(t[:, center - (L - 1):center + L], plain * math.log2(math.e))
        assert float(t[:, :center - (L - 1)].abs().sum()) == 0 and float(t[:, center + L:].abs().sum()) == 0
