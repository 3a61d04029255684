"""GPU: the T5 encoder kernels (embedding gather, T5LayerNorm, gated GELU, relative-bias attention) against torch fp32, and the
whole T5Encoder against the golden minted from transformers.T5EncoderModel (rel-rms error vs fp32 <= 1.5x the error of the
class's own bf16 run; cosine >= 0.999 on the unmasked positions)."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def bfr(t):
    return t.to(torch.bfloat16).float()


def check(out, ref, rel=2 ** -7, what=""):
    out, ref = out.float().cpu(), ref.float().cpu()
    err = (out - ref).abs().max().item()
    assert err <= rel * ref.abs().max().item(), f"{what}: max|err| {err:.4e} vs max|ref| {ref.abs().max().item():.3f}"


def test_t5_elementwise_kernels():
    from videosys_amd import ops

    g = torch.Generator().manual_seed(3)
    table = bfr(torch.randn(50, 256, generator=g))
    ids = torch.randint(0, 50, (77,), generator=g)
    out = ops.gather_rows(table.to(torch.bfloat16).to(dev()), ids.to(dev()))
    assert torch.equal(out.float().cpu(), table[ids])
    for C in (256, 4096, 8192):
        x = bfr(torch.randn(37, C, generator=g) * 3)
        w = bfr(1 + 0.1 * torch.randn(C, generator=g))
        ref = w * bfr(x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6))
        check(ops.rms_norm_rows(x.to(torch.bfloat16).to(dev()), w.to(torch.bfloat16).to(dev())), ref, what=f"rms norm C={C}")
    # the stand-alone module under the reference's name (models/modules/normalization.py LlamaRMSNorm; its own test is
    # tests/test_rms_norm.py): any leading shape, q / k head widths
    from videosys_amd.modules import LlamaRMSNorm

    for shape in ((2, 50, 16, 72), (3, 7, 128)):
        x = bfr(torch.randn(*shape, generator=g) * 2)
        w = bfr(1 + 0.1 * torch.randn(shape[-1], generator=g))
        norm = LlamaRMSNorm(shape[-1], device=dev()).load_state_dict({"weight": w})
        ref = w * bfr(x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6))
        out = norm(x.to(torch.bfloat16).to(dev()))
        assert out.shape == x.shape
        check(out.reshape(-1, shape[-1]), ref.reshape(-1, shape[-1]), what=f"LlamaRMSNorm {shape}")
    h = bfr(torch.randn(33, 2 * 512, generator=g) * 2)
    ref = bfr(F.gelu(h[:, :512], approximate="tanh")) * h[:, 512:]
    check(ops.geglu(h.to(torch.bfloat16).to(dev())), ref, what="gated gelu")


@pytest.mark.parametrize("B,L,H,lens", [(2, 150, 4, (150, 97)), (1, 300, 8, (120,)), (2, 77, 2, (1, 77))])
def test_t5_attention_matches_torch(B, L, H, lens):
    from videosys_amd import ops

    g = torch.Generator().manual_seed(L + H)
    inner = H * 64
    qkv = bfr(torch.randn(B * L, 3 * inner, generator=g))
    rel = torch.randn(H, 2 * L - 1, generator=g)
    klen = torch.tensor(lens, dtype=torch.int32)
    q, k, v = (qkv[:, i * inner:(i + 1) * inner].view(B, L, H, 64).transpose(1, 2) for i in range(3))
    idx = (torch.arange(L)[None, :] - torch.arange(L)[:, None]) + L - 1
    s = q @ k.transpose(-1, -2) + rel[:, idx][None]
    for b in range(B):
        s[b, :, :, lens[b]:] = float("-inf")
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * L, inner)
    out = ops.t5_attention(qkv.to(torch.bfloat16).to(dev()), rel.to(dev()), klen.to(dev()), B, L, H)
    check(out, ref, what="t5 attention")
    # the matrix-pipe form (d64 flash kernel + relative-position bias hook): same reference, valid query rows
    L1 = L - 1
    center = (L + 127) // 128 * 128 - 1
    tab = torch.zeros(H, center + (L + 63) // 64 * 64)
    tab[:, center - L1:center + L] = rel * math.log2(math.e)
    out2 = ops.t5_attention_mfma(qkv.to(torch.bfloat16).to(dev()), tab.to(dev()), center, lens, B, L, H)
    check(out2, ref, what="t5 attention (mfma)")


@pytest.mark.parametrize("M,N,K,nsplit,res,pad", [(300, 512, 2048, None, True, 384), (300, 768, 1024, 1, False, 384), (77, 256, 4096, 8, True, 128),
                                                   (600, 1024, 512, 2, False, 128), (128, 4096, 4096, None, True, 128),
                                                   (600, 1000, 1024, 2, True, 384), (300, 4096, 10240, None, True, 384),
                                                   (300, 1024, 4096, 3, True, 384), (200, 512, 1056, 5, False, 384)])
def test_linear_skinny_matches_torch(M, N, K, nsplit, res, pad):
    """The weight-streaming linear (transposed split-K GEMM + vsys_splitk_reduce_t) against torch on bf16-rounded operands; rows
    past M of the padded activation buffer are poisoned with NaN: they must not reach any result row."""
    from videosys_amd import ops

    g = torch.Generator().manual_seed(M + N + K)
    Mp = (M + pad - 1) // pad * pad     # 384: the 256 x 384 tile (one workgroup per weight panel); 128: the 128-column kernel
    x = torch.full((Mp, K), float("nan"))
    x[:M] = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    r = torch.randn(Mp, N, generator=g) if res else None
    xb, wb = x.to(torch.bfloat16), w.to(torch.bfloat16)
    ref = xb[:M].float() @ wb.float().t()
    if res:
        ref = ref.to(torch.bfloat16).float() + r[:M].to(torch.bfloat16).float()
    out = ops.linear_skinny(xb.to(dev()), M, wb.to(dev()), res=None if r is None else r.to(torch.bfloat16).to(dev()), nsplit=nsplit)
    assert out.shape == (Mp, N)
    check(out[:M], ref, what=f"linear_skinny {M}x{N}x{K} split {nsplit}")
    if res:   # in place on the residual stream, as the encoder uses it
        rr = r.to(torch.bfloat16).to(dev())
        out2 = ops.linear_skinny(xb.to(dev()), M, wb.to(dev()), res=rr, out=rr, nsplit=nsplit)
        assert out2.data_ptr() == rr.data_ptr() and torch.equal(out2[:M], out[:M])


@pytest.mark.parametrize("skinny,mfma", [(True, True), (False, True), (True, False)])
def test_t5_encoder_matches_transformers_golden(skinny, mfma):
    from videosys_amd.t5 import T5Encoder, synth_state_dict

    gold = load_golden("t5_small.pt")
    cfg = gold["cfg"]
    enc = T5Encoder(device=dev(), **cfg).load_state_dict(synth_state_dict(seed=gold["seed"], **cfg))
    enc.mfma_attention = mfma         # False: the VALU attention kernel
    if not skinny:
        enc.skinny_rows = 0           # the many-rows path (direct 128-column GEMMs)
    out = enc(gold["ids"], gold["mask"]).last_hidden_state.float().cpu()
    ref = gold["out_fp32"]
    keep = gold["mask"].bool()                        # padded query positions are junk in every implementation
    rms = lambda t: t.pow(2).mean().sqrt().item()
    floor = rms((gold["out_bf16"].float() - ref)[keep]) / rms(ref[keep])
    mine = rms((out - ref)[keep]) / rms(ref[keep])
    cos = F.cosine_similarity(out[keep].flatten(), ref[keep].flatten(), dim=0).item()
    assert mine <= 1.5 * floor + 1e-3, f"rel rms {mine:.4f} vs transformers-bf16 floor {floor:.4f}"
    assert cos >= 0.999, cos
