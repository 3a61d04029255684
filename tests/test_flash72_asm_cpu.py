"""CPU (-m "not gpu"): the hand-allocated instruction stream of flash_attn_d72_w64 (csrc/attention_w64.hip).

  * the committed csrc/flash72_w64_asm.inc is what csrc/gen/flash72_gen.py generates (the build does not run the generator);
  * tools/gcn_emu.py executes that text for all four waves of a workgroup — LDS-DMA pieces and ds_read results landing as LATE and
    as EARLY as the counters allow, both stepping orders of the waves — around the numpy restatement of the kernel's C++ prologue /
    epilogue (tools/flash72_emu_case.py), against float64 attention on the same bf16 inputs: even / odd tile counts, a ragged last
    tile, inputs that force the deferred-rescale branch; tolerance 2^-8 of max|ref| (P is rounded to bf16);
  * no software-visible hazard in any executed path (MFMA result -> use, VALU -> MFMA operand, VALU -> permlane, transcendental -> use,
    M0 -> LDS-DMA, MFMA C read -> overwrite), and every MFMA gap of the steady-state loop carries at most 5 other instructions."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "videosys_amd", "csrc", "gen"))


def test_committed_include_is_what_the_generator_emits(tmp_path):
    import flash72_gen as G

    out = str(tmp_path / "f.inc")
    G.write_inc(out)
    with open(out) as a, open(os.path.join(ROOT, "videosys_amd", "csrc", "flash72_w64_asm.inc")) as b:
        assert a.read() == b.read(), "regenerate: python videosys_amd/csrc/gen/flash72_gen.py"


def test_steady_state_gaps_hold_at_most_five_fillers():
    import flash72_gen as G

    lines = G.generate(1)       # the shipped placement variant (attention.hip W64_DEFAULT_VAR)
    top, tail = lines.index("TOP_%=:"), lines.index("TAIL0_%=:")
    loop = [ln for ln in lines[top:tail] if not ln.endswith(":")]
    gaps, cur, seen = [], 0, False
    for ln in loop:
        if ln.startswith("v_mfma"):
            if seen:
                gaps.append(cur)
            cur, seen = 0, True
        else:
            cur += 1
    n_mfma = sum(1 for ln in loop if ln.startswith("v_mfma"))
    assert n_mfma == 88                      # two tiles of 44 per trip of the unrolled loop
    # phase boundaries (wait + barrier, loop control, the branch to the rare rescale block) are the only longer gaps
    assert sorted(gaps)[-9] <= 5 and max(gaps) <= 12, sorted(gaps)[-12:]
    assert sum(gaps) / len(gaps) <= 4.6


@pytest.mark.parametrize("kv_len,spike,qscale,late_vm,late_ds,order", [
    (256, False, 1.0, True, True, None),            # 4 tiles: no trip of the main loop
    (320, True, 3.0, False, False, [3, 2, 1, 0]),   # odd tile count, rescale branch taken repeatedly
    (300, True, 1.0, True, False, None),            # ragged last tile (the text length of the cross attention)
    (448, False, 1.0, False, True, [3, 2, 1, 0]),
    (384, False, 6.0, True, True, [3, 2, 1, 0]),    # large logits: the branch fires on most tiles
])
@pytest.mark.parametrize("variant", [0, 1, 3])
def test_emulated_workgroup_matches_numpy_attention(kv_len, spike, qscale, late_vm, late_ds, order, variant):
    import flash72_emu_case as C

    with np.errstate(all="ignore"):
        err, viol, stats = C.run(kv_len, spike=spike, qscale=qscale, late_vm=late_vm, late_ds=late_ds, order=order, variant=variant)
    assert not viol, viol[:5]
    assert err <= 2.0**-8, err
    if spike or qscale > 1.0:
        assert stats["counts"].get("v_accvgpr_read_b32", 0) >= 96, "the rescale branch was never taken"
    assert stats["barriers"] == 2 + (kv_len + 63) // 64 - 1


@pytest.mark.parametrize("kv_len,nitems,late_vm,late_ds,order", [
    (256, 3, True, True, None),               # 4 tiles per item: the descriptor switch precedes the first tile
    (512, 3, False, False, [3, 2, 1, 0]),
    (1024, 2, True, False, None),             # the config-2 spatial length
    (256, 4, False, True, [3, 2, 1, 0]),
])
def test_emulated_persistent_walk_matches_numpy_attention(kv_len, nitems, late_vm, late_ds, order):
    """The persistent form (FLASH72_W64P_ASM): consecutive items on one workgroup state — tiles 0..3 and the Q rows of item k+1 are
    fetched by the tail of item k, K / Vt change between items, the output stores of item k are still in flight when item k+1
    starts, the last item's tail issues zero-length pieces."""
    import flash72_emu_case as C

    with np.errstate(all="ignore"):
        errs, viol, stats = C.run_persist(kv_len, nitems=nitems, late_vm=late_vm, late_ds=late_ds, order=order, spike=True, qscale=2.0)
    assert not viol, viol[:5]
    assert max(errs) <= 2.0**-8, errs


@pytest.mark.parametrize("kv_len,spike,qscale,late_vm,late_ds,order", [
    (256, False, 1.0, True, True, None),
    (320, True, 3.0, False, False, [3, 2, 1, 0]),
    (300, True, 1.0, True, False, None),            # ragged last tile
    (448, False, 6.0, False, True, [3, 2, 1, 0]),
])
@pytest.mark.parametrize("variant", [1, 4])
def test_emulated_workgroup_head_dim_64(kv_len, spike, qscale, late_vm, late_ds, order, variant):
    """The stream generated for head_dim 64 (FLASH64_W64_ASM, csrc/attention64_w64.hip): swizzled 128-byte K rows, four LDS-DMA
    pieces per wave and tile, the ones rows in the constant third block of the Vt image."""
    import flash72_emu_case as C

    with np.errstate(all="ignore"):
        err, viol, stats = C.run64(kv_len, spike=spike, qscale=qscale, late_vm=late_vm, late_ds=late_ds, order=order, variant=variant)
    assert not viol, viol[:5]
    assert err <= 2.0 ** -8, err
    assert stats["counts"].get("buffer_load_dwordx4", 0) >= 16


@pytest.mark.parametrize("kv_len,late_vm,late_ds,order", [(256, True, True, None), (300, False, False, [3, 2, 1, 0]), (448, True, False, None),
                                                          (320, False, True, [3, 2, 1, 0])])
@pytest.mark.parametrize("d64", [False, True])
def test_emulated_workgroup_without_running_max(kv_len, late_vm, late_ds, order, d64):
    """Variant 5 of the stream (both head sizes): the caller's per-row bound |q_i| max_j |k_j| as the subtracted constant — no row
    max, no rescale block, four / five of the eight P units of a tile in the PV phase.  Same tolerance as the running-max stream."""
    import flash72_emu_case as C

    run = C.run64 if d64 else C.run
    with np.errstate(all="ignore"):
        err, viol, stats = run(kv_len, late_vm=late_vm, late_ds=late_ds, order=order, variant=5)
    assert not viol, viol[:5]
    assert err <= 2.0 ** -8, err
    assert stats["counts"].get("v_max3_f32", 0) == 0 and stats["counts"].get("v_exp_f32_e32", 0) >= 64 * ((kv_len + 63) // 64)


def test_stream_without_running_max_is_lighter():
    import flash72_gen as G

    for d64, limit in ((False, 3.6), (True, 3.9)):
        lines = G.generate(5, d64=d64)
        top, tail = lines.index("TOP_%=:"), lines.index("TAIL0_%=:")
        loop = [ln for ln in lines[top:tail] if not ln.endswith(":")]
        n_mfma = sum(1 for ln in loop if ln.startswith("v_mfma"))
        assert n_mfma == (80 if d64 else 88)
        assert (len(loop) - n_mfma) / n_mfma <= limit      # 4.3 with the row max (test_steady_state_gaps...)
        assert not any("RESC" in ln for ln in lines)


def test_persistent_walk_without_running_max():
    import flash72_emu_case as C

    with np.errstate(all="ignore"):
        errs, viol, _ = C.run_persist(512, nitems=3, late_vm=False, late_ds=True, order=[3, 2, 1, 0], static=True)
    assert not viol, viol[:5]
    assert max(errs) <= 2.0 ** -8, errs


def test_every_scc_reader_sits_behind_its_compare():
    """An s_cselect / s_cbranch_scc* reads SCC; every SALU add / sub / shift overwrites it.  The placement spreads independent
    streams over the MFMA gaps, so the pair must be emitted as one item — in EVERY generated statement the instruction in front of
    an SCC reader is its s_cmp (the hardware failure this guards: the LDS ring of variant 5 never wrapped; the emulator models SCC
    since)."""
    import flash72_gen as G

    texts = [G.generate(v) for v in G.VARIANTS] + [G.generate(1, persist=True), G.generate(5, persist=True),
                                                    G.generate(1, d64=True), G.generate(4, d64=True), G.generate(5, d64=True)]
    for lines in texts:
        code = [ln for ln in lines if not ln.endswith(":")]
        for i, ln in enumerate(code):
            if ln.startswith(("s_cselect", "s_cbranch_scc")):
                j = i - 1
                while not code[j].startswith("s_cmp"):     # only instructions that leave SCC alone may sit in between
                    assert not code[j].startswith(("s_add", "s_sub", "s_lsh", "s_and", "s_or", "s_xor", "s_min", "s_max", "s_bf", "s_ash")), (code[j], ln)
                    j -= 1
                    assert j >= 0 and i - j < 8, ln


@pytest.mark.parametrize("kv_len,nitems,static", [(320, 4, False), (300, 3, True), (448, 3, True), (576, 2, False)])
def test_persistent_walk_any_tile_count(kv_len, nitems, static):
    """The ring continues from item to item: tile 0 of the k-th item sits in stage (k ntiles) mod 4 (5, 7 and 9 tiles here), a
    ragged key count is walked as whole tiles of zero-padded keys — with and without the running max."""
    import flash72_emu_case as C

    with np.errstate(all="ignore"):
        errs, viol, _ = C.run_persist(kv_len, nitems=nitems, late_vm=kv_len % 128 == 0, late_ds=True,
                                      order=[3, 2, 1, 0] if nitems == 3 else None, static=static)
    assert not viol, viol[:5]
    assert max(errs) <= 2.0 ** -8, errs
