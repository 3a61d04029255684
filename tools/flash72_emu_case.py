#!/usr/bin/env python
"""One workgroup of flash_attn_d72_w64_kernel on tools/gcn_emu.py (TEST INFRASTRUCTURE): the C++ prologue / epilogue of
csrc/attention_w64.hip restated in numpy around the generated instruction stream (csrc/gen/flash72_gen.py), against plain
numpy attention on the same bf16 inputs.

    python tools/flash72_emu_case.py [kv_len] [--spike]      prints the error and the hazard report"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "videosys_amd", "csrc", "gen"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import flash72_gen as G  # noqa: E402
import gcn_emu as E  # noqa: E402

HD, KROW, VROW, K_TILE, STAGE = 72, 144, 128, 9216, 21504


def make_case(kv_len, seed=0, spike=False, qscale=1.0):
    """Q [256][72] bf16 (already normed), Kp [kv_pad][72] bf16 (normed, scaled: exp2 domain), Vt [96][kv_pad] bf16 + ones rows"""
    rng = np.random.default_rng(seed)
    kv_pad = (kv_len + 63) // 64 * 64
    q = E.bf16_to_f32(E.bf16_round(rng.standard_normal((256, HD)).astype(np.float32) * qscale))
    k = E.bf16_to_f32(E.bf16_round(rng.standard_normal((kv_pad, HD)).astype(np.float32) * 0.35))
    vv = E.bf16_to_f32(E.bf16_round(rng.standard_normal((kv_pad, HD)).astype(np.float32)))
    k[kv_len:] = 0
    vv[kv_len:] = 0
    if spike:   # one key far above the rest for some queries, late in the sequence: forces the rescale branch
        k[kv_len - 70] = q[5] * 3.0
        k[kv_len - 70] = E.bf16_to_f32(E.bf16_round(k[kv_len - 70]))
        k[130] = E.bf16_to_f32(E.bf16_round(q[200] * 2.0))
    vt = np.zeros((96, kv_pad), dtype=np.float32)
    vt[:HD] = vv.T
    vt[72, :kv_len] = 1.0
    vt[76, :kv_len] = 1.0
    return q, k, vt, kv_pad


def reference(q, k, vt, kv_len):
    s = q.astype(np.float64) @ k[:kv_len].astype(np.float64).T          # exp2 domain
    p = np.exp2(s - s.max(1, keepdims=True))
    return (p @ vt[:HD, :kv_len].astype(np.float64).T) / p.sum(1, keepdims=True)


def to_bytes_bf16(x):
    return E.bf16_round(x).astype(np.uint16).view(np.uint8).reshape(-1)


def run(kv_len, seed=0, spike=False, late_vm=True, late_ds=True, order=None, qscale=1.0, lb=0, variant=1):
    q, k, vt, kv_pad = make_case(kv_len, seed, spike, qscale)
    ntiles = (kv_len + 63) // 64
    lines = G.generate(variant)
    kbytes = to_bytes_bf16(k)
    vbytes = to_bytes_bf16(vt)
    rs_k, rs_v = E.Rsrc(kbytes, kv_pad * HD * 2), E.Rsrc(vbytes, 96 * kv_pad * 2)
    lane = np.arange(64)
    l31, hi = lane & 31, lane >> 5
    binds = []
    for w in range(4):
        s4k = w == 0
        s4j = 8 if w == 1 else 9
        binds.append({"rk": "s[4:7]", "rv": "s[8:11]", "r4": "s[12:15]", "wl": "s16", "sv0": "s17", "sv1": "s18", "s4": "s19",
                      "st4": "s20", "l4": "s21", "lb": "s22", "nt": "s23", "lim": "v210", "kvo": "v211", "vvo": "v212", "v4o": "v213",
                      "kfa": "v214", "vfa0": "v215", "vfa1": "v216", "vfa2": "v217", "vfa3": "v218"})
    wg = E.Workgroup(lines, binds, late_vm=late_vm, late_ds=late_ds)
    wg.lds[:] = 0xAB   # garbage: anything the kernel relies on must have been written
    # the C++ prologue zeroes rows 80..95 of every stage's Vt image
    for st in range(4):
        o = lb + st * STAGE + K_TILE + 80 * VROW
        wg.lds[o:o + 16 * VROW] = 0
    for w, wave in enumerate(wg.waves):
        s4k = w == 0
        s4j = 8 if w == 1 else 9
        wg.rsrc[4], wg.rsrc[8] = rs_k, rs_v
        wave.s.update({16: w * 1024, 17: w * 8 * kv_pad * 2, 18: (w + 4) * 8 * kv_pad * 2, 19: 8 * 1024 if s4k else s4j * 8 * kv_pad * 2,
                       20: K_TILE if s4k else 128, 21: 8 * 1024 if s4k else K_TILE + s4j * 1024, 22: lb, 23: ntiles})
        k_voff = lane * 16
        v_voff0 = (lane >> 3) * kv_pad * 2 + (((lane & 7) ^ (lane >> 4)) << 4)
        v_voff = v_voff0 ^ ((w & 1) << 6)
        voff_4 = k_voff if s4k else (v_voff0 ^ ((s4j & 1) << 6))
        krow = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3)
        kfa = krow * KROW + 16 * hi
        v_roff = K_TILE + l31 * VROW + (((2 * hi) ^ ((l31 >> 1) & 7)) << 4)
        for reg, val in ((210, kv_len - (ntiles - 1) * 64 - 16 * hi), (211, k_voff), (212, v_voff), (213, voff_4), (214, kfa),
                         (215, v_roff ^ 0), (216, v_roff ^ 16), (217, v_roff ^ 64), (218, v_roff ^ 80)):
            wave.v[reg] = np.asarray(val).astype(np.int64).astype(np.uint32)
        # Q fragments -> a[96:135]: block blk, chunk c, word j holds d = 16c + 8hi + 2j, +1 of row 64w + 32blk + l31
        for blk in range(2):
            rows = 64 * w + 32 * blk + l31
            for c in range(5):
                for j in range(4):
                    d = 16 * c + 8 * hi + 2 * j
                    lo = np.where(d < HD, E.bf16_round(q[rows, np.minimum(d, HD - 1)]), 0)
                    hi_ = np.where(d + 1 < HD, E.bf16_round(q[rows, np.minimum(d + 1, HD - 1)]), 0)
                    wave.a[96 + 20 * blk + 4 * c + j] = (lo | (hi_ << 16)).astype(np.uint32)
    # slot-4 descriptor differs per wave: the emulator keys descriptors by their first SGPR, so step waves with their own table
    viol = []
    wg_r4 = {0: rs_k, 1: rs_v, 2: rs_v, 3: rs_v}

    class RsrcView(dict):
        pass

    # run with a per-wave view of s[12:15]
    orig_run = E.Wave.run_until_barrier

    def run_wave(self):
        self.wg.rsrc[12] = wg_r4[self.wid]
        return orig_run(self)

    E.Wave.run_until_barrier = run_wave
    try:
        viol = wg.run(order=order)
    finally:
        E.Wave.run_until_barrier = orig_run
    # epilogue: O^T[d][q] / l
    out = np.zeros((256, HD), dtype=np.float32)
    for w, wave in enumerate(wg.waves):
        for blk in range(2):
            base = 48 * blk
            l_ = E.f32(wave.a[base + 32 + 4])        # d = 72 (hi = 0) / 76 (hi = 1)
            for dt in range(3):
                for r in range(16):
                    d = 32 * dt + (r & 3) + 8 * (r >> 2) + 4 * hi
                    val = E.f32(wave.a[base + 16 * dt + r]) / l_
                    ok = d < HD
                    out[(64 * w + 32 * blk + l31)[ok], d[ok]] = val[ok]
    ref = reference(q, k, vt, kv_len)
    err = np.abs(out - ref).max() / np.abs(ref).max()
    stats = {"barriers": wg.waves[0].nbarrier, "counts": wg.waves[0].count}
    return err, viol, stats


if __name__ == "__main__":
    kv = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 384
    err, viol, stats = run(kv, spike="--spike" in sys.argv)
    print("rel err", err, "violations", len(viol))
    for x in viol[:20]:
        print("  ", x)
    c = stats["counts"]
    print({k: c[k] for k in sorted(c) if c[k] > 20})
