#!/usr/bin/env python
"""Generator of the hand-allocated gfx950 instruction stream of flash_attn_d72_w64 (csrc/attention_w64.hip).

    python videosys_amd/csrc/gen/flash72_gen.py            -> videosys_amd/csrc/flash72_w64_asm.inc  (committed; build() does not run this)

One wave = 64 query rows (two 32-row blocks A, B) of one (batch, head), ONE wave per SIMD (512 registers), 4 waves per workgroup.
Replaces the tile loop of flash_attn_d72_kernel (attention.hip; reference: modules/attentions.py:80-120 SDPA of OpenSoraAttention)
with the same arithmetic: S^T = K Q^T - m on v_mfma_f32_32x32x16_bf16 (running max as the C operand), deferred rescale
(threshold 8 in the exp2 domain), P = exp2(S) rounded to bf16, O^T += Vt P^T with the row sum in the ones rows of Vt.

Register file (fixed, listed as clobbers of the asm statement):
  a[0:47] O_A   a[48:95] O_B   a[96:115] Q_A   a[116:135] Q_B   a[136:175] K fragments   a[176:223] Vt fragments
  v[0:63] S buffer 0 (A | B)   v[64:127] S buffer 1   v[128:143] P_A   v[144:159] P_B   v[160:175] -m_A splat   v[176:191] -m_B splat
  v[192:209] temporaries
Software pipeline per 64-key tile i (two phases, one s_barrier):
  X_i  20 MFMA  S(i+1) = K(i+1) Q^T - m      beside   P(i) = exp2(S(i)) (A, and keys 0..31 of B), Vt(i) fragment reads
  Y_i  24 MFMA  O += Vt(i) P(i)^T            beside   exp2 of B's keys 32..63, row max of S(i+1), K(i+2) fragment reads, LDS-DMA of tile i+4
with every non-MFMA instruction PLACED in a gap behind an MFMA (<= 5 per gap where the counts allow: MI355X_MICROARCH.md "one wave
per SIMD" row).  K/V tiles travel HBM -> LDS by LDS-DMA into a ring of 4 stages, two tiles ahead of their first reader.
tools/gcn_emu.py executes the text this file emits (all four waves, LDS-DMA / ds_read completion as late and as early as the
counters allow) and checks the software-visible hazards; tests/test_flash72_asm_cpu.py runs it against numpy attention."""
import os
import sys

K_TILE, STAGE, NSTAGE = 9216, 21504, 4
# head_dim 64 (flash_attn_d64_w64, csrc/attention64_w64.hip; set by generate(d64=True)): 4 QK^T chunks, K rows of 128 bytes whose 16-byte
# chunks are XOR-swizzled (four fragment address registers instead of one + immediates), 8 + 8 LDS-DMA pieces per tile (four per
# wave), the third 32-row block of the Vt image (ones rows 72 / 76: the row sum) constant in LDS instead of fetched
HD64 = False
NCC = 5            # QK^T chunks of 16 dims
NPIECE = 5         # LDS-DMA pieces per wave and tile
KADDR64 = [196, 210, 211, 212]
O = {"A": 0, "B": 48}
Q = {"A": 96, "B": 116}
KF, VF = 136, 176
SBUF = [{"A": 0, "B": 32}, {"A": 64, "B": 96}]
P = {"A": 128, "B": 144}
NM = {"A": 160, "B": 176}
MX = {"A": 192, "B": 193}
TT = {"A": 194, "B": 195}
KADDR = 196
VADDR = [197, 198, 199, 200]
DL = {"A": 201, "B": 202}
AL = {"A": 203, "B": 204}
TMP = [205, 206, 207, 208]
NEGBIG = 209
NV = 210        # v0..v209 are the asm's
NA = 224        # a0..a223
# scalar registers of the asm (clobbered): s40..s63
S_CNT, S_ST, S_ST2, S_T, S_KA, S_KB, S_VC, S_VD, S_E, S_DST, S_WRAP, S_T2 = range(40, 52)
NS_LO, NS_HI = 40, 82


def v(i, n=1):
    return f"v{i}" if n == 1 else f"v[{i}:{i + n - 1}]"


def a(i, n=1):
    return f"a{i}" if n == 1 else f"a[{i}:{i + n - 1}]"


def s(i):
    return f"s{i}"


def kf(kt, cc):
    return KF + kt * 20 + cc * 4


def vf(kt, cc, dt):
    return VF + ((kt * 2 + cc) * 3 + dt) * 4


def sreg(buf, blk, kt, r=0):
    return SBUF[buf][blk] + kt * 16 + r


def preg(blk, kt, cc):
    return P[blk] + (kt * 2 + cc) * 4


# STATIC (variant 5): no running max.  The caller hands in, per query row, an upper bound m_i >= max_j s_ij (Cauchy-Schwarz on the
# normed rows: |q_i| * max_j |k_j|) as the -m splat; P = exp2(s - m_i) <= 1 needs no row max, no rescale branch and no adoption of the
# first tile — softmax is invariant under the choice of m as long as nothing under- or overflows (the caller checks 2 m_i < 126).
STATIC = False
NLATE = 2         # P units (of the 8 of a tile) computed in the PV phase
PHASE = False     # lab (variant 10): s_memtime around the two phases of every loop tile, summed per wave: X, wait + barrier, Y
PSTAMP = False    # lab: s_memtime behind the opening wait + barrier and at the loop head of the persistent form
PERSIST = False   # set by generate(): the persistent (workgroup walks several work items) form of the statement
S_QP, S_QM, S_QO = 52, 53, 54
DK, DV, D4 = 60, 64, 68           # LDS-DMA descriptors of the tiles being fetched (persistent form: a copy the tail re-points to the
S_KNR, S_VNR = 72, 73             # next item's K / Vt), byte sizes of one (batch, head)'s Kp / Vt


def sq(i):
    return f"s[{i}:{i + 3}]"


class Emit:
    def __init__(self):
        self.lines = []
        self.uid = 0

    def __call__(self, text):
        self.lines.extend(text.split("\n"))

    def label(self, name):
        self.lines.append(name + ":")


def schedule(e, mfmas, fillers, cap=5, front=False):
    """Emit MFMAs in order; behind MFMA g up to ``cap`` fillers whose ``after`` index is <= g (fillers keep their order).
    fillers: list of (text, after) — ``after`` = the index of the last MFMA that must have ISSUED before it."""
    q = list(fillers)
    n = len(mfmas)
    for g, m in enumerate(mfmas):
        e(m)
        left = n - g
        # even spreading: what is left over what is left, but never above cap and always respecting ``after``
        want = (cap if front else min(cap, -(-len(q) // left))) if q else 0
        k = 0
        while q and k < want and q[0][1] <= g:
            e(q.pop(0)[0])
            k += 1
    for t, _ in q:
        e(t)


def place(e, mfmas, streams):
    """Emit the MFMAs in order; every stream (items, first_gap, last_gap) is spread evenly over the gaps of its window (an item lands
    behind the MFMA of its gap), streams in the order given inside a gap.  Returns the number of fillers per gap."""
    n = len(mfmas)
    per = [[] for _ in range(n)]
    for items, g0, g1 in streams:
        g0, g1 = max(0, g0), min(n - 1, g1)
        w = g1 - g0 + 1
        for k, t in enumerate(items):
            per[g0 + (k * w) // len(items)].append(t)
    for g, m in enumerate(mfmas):
        e(m)
        for t in per[g]:
            e(t)
    return [len(x) for x in per]


def qk_mfmas(nxt):
    out = []
    for c in range(NCC):
        for blk in ("A", "B"):
            for kt in range(2):
                d = v(sreg(nxt, blk, kt), 16)
                src_c = v(NM[blk], 16) if c == 0 else d
                out.append(f"v_mfma_f32_32x32x16_bf16 {d}, {a(kf(kt, c), 4)}, {a(Q[blk] + 4 * c, 4)}, {src_c}")
    # order inside a chunk: A.kt0, A.kt1, B.kt0, B.kt1
    return out


def pv_mfmas():
    out = []
    for kt in range(2):
        for blk in ("A", "B"):
            for cc in range(2):
                for dt in range(3):
                    d = a(O[blk] + 16 * dt, 16)
                    out.append((f"v_mfma_f32_32x32x16_bf16 {d}, {a(vf(kt, cc, dt), 4)}, {v(preg(blk, kt, cc), 4)}, {d}", blk, kt))
    return out   # order: A.kt0 (6), B.kt0 (6), A.kt1 (6), B.kt1 (6)


def p_unit(cur, blk, kt, cc):
    """8 exps in place + 4 converts: S[8cc .. 8cc+7] of (blk, kt) -> P fragment (blk, kt, cc)"""
    base = sreg(cur, blk, kt, 8 * cc)
    out = [f"v_exp_f32_e32 {v(base + r)}, {v(base + r)}" for r in range(8)]
    out += [f"v_cvt_pk_bf16_f32 {v(preg(blk, kt, cc) + w)}, {v(base + 2 * w)}, {v(base + 2 * w + 1)}" for w in range(4)]
    return out


def v_reads():
    out = []
    for kt in range(2):
        for cc in range(2):
            for dt in range(3):
                out.append(f"ds_read_b128 {a(vf(kt, cc, dt), 4)}, {v(VADDR[kt * 2 + cc])} offset:{dt * 4096}")
    return out


def k_reads():
    if HD64:
        return [f"ds_read_b128 {a(kf(kt, cc), 4)}, {v(KADDR64[cc])} offset:{kt * 4096}" for kt in range(2) for cc in range(4)]
    return [f"ds_read_b128 {a(kf(kt, cc), 4)}, {v(KADDR)} offset:{kt * 4608 + cc * 32}" for kt in range(2) for cc in range(5)]


def k_addr(stage_sreg, plus=0):
    """the K fragment read address(es) of the tile in the stage at s[stage_sreg] (+ ``plus`` bytes)"""
    if HD64:
        out = [f"v_add_u32_e32 {v(KADDR64[cc])}, {s(stage_sreg)}, %[kfa{cc}]" for cc in range(4)]
        if plus:
            out += [f"v_add_u32_e32 {v(KADDR64[cc])}, {plus}, {v(KADDR64[cc])}" for cc in range(4)]
        return out
    out = [f"v_add_u32_e32 {v(KADDR)}, {s(stage_sreg)}, %[kfa]"]
    if plus:
        out.append(f"v_add_u32_e32 {v(KADDR)}, {plus}, {v(KADDR)}")
    return out


def max_chain(buf, blk):
    r0 = SBUF[buf][blk]
    out = [f"v_max3_f32 {v(MX[blk])}, {v(r0)}, {v(r0 + 1)}, {v(r0 + 2)}"]
    for i in range(3, 32, 2):
        b = v(r0 + i + 1) if i + 1 < 32 else v(r0 + i)
        out.append(f"v_max3_f32 {v(MX[blk])}, {v(MX[blk])}, {v(r0 + i)}, {b}")
    return out   # 16 instructions: 3 + 15 * 2 = 33 >= 32 values


def mask_ops(buf):
    """keys 64 (ntiles - 1) + 32 kt + 16 hi + r >= kv_len of the LAST tile -> -1e30 (lim = kv_len - 64 (ntiles - 1) - 16 hi)"""
    out = []
    for blk in ("A", "B"):
        for kt in range(2):
            for r in range(16):
                reg = v(sreg(buf, blk, kt, r))
                out.append(f"v_cmp_lt_i32_e32 vcc, {kt * 32 + r}, %[lim]")
                out.append(f"v_cndmask_b32_e32 {reg}, {v(NEGBIG)}, {reg}, vcc")
    return out


def dma_tile_static(e, stage_expr_k, t):
    """prologue: the five pieces of tile t (this wave's) into stage t; soff registers are advanced afterwards"""
    raise NotImplementedError


def dma_ops():
    """the wave's five LDS-DMA pieces of the next tile into the stage at s[S_ST] (+ soff advance); (text, after) tuples spread by
    the caller.  An independent instruction sits between every M0 write and the load that reads it."""
    rk, rv, r4 = (sq(DK), sq(DV), sq(D4)) if PERSIST else ("%[rk]", "%[rv]", "%[r4]")
    ops = [
        f"s_add_i32 {s(S_DST)}, {s(S_ST)}, %[wl]",
        f"s_mov_b32 m0, {s(S_DST)}",
        f"s_add_i32 {s(S_T2)}, {s(S_DST)}, 4096",
        f"buffer_load_dwordx4 %[kvo], {rk}, {s(S_KA)} offen lds",
        f"s_mov_b32 m0, {s(S_T2)}",
        f"s_add_i32 {s(S_KA)}, {s(S_KA)}, {K_TILE}",
        f"buffer_load_dwordx4 %[kvo], {rk}, {s(S_KB)} offen lds",
        f"s_add_i32 m0, {s(S_DST)}, {K_TILE}",
        f"s_add_i32 {s(S_KB)}, {s(S_KB)}, {K_TILE}",
        f"buffer_load_dwordx4 %[vvo], {rv}, {s(S_VC)} offen lds",
        f"s_add_i32 m0, {s(S_DST)}, {K_TILE + 4096}",
        f"s_add_i32 {s(S_VC)}, {s(S_VC)}, 128",
        f"buffer_load_dwordx4 %[vvo], {rv}, {s(S_VD)} offen lds",
        f"s_add_i32 m0, {s(S_ST)}, %[l4]",
        f"s_add_i32 {s(S_VD)}, {s(S_VD)}, 128",
        f"buffer_load_dwordx4 %[v4o], {r4}, {s(S_E)} offen lds",
        f"s_add_i32 {s(S_E)}, {s(S_E)}, %[st4]",
    ]
    return ops[:13] + [ops[14]] if HD64 else ops   # (d64: four pieces — K w, K w + 4, Vt w, Vt w + 4)


def soff_reset(tile):
    """the five LDS-DMA source offsets of this wave for tile ``tile`` of a (batch, head) (persistent form: from wl, kvp2, s4, st4)"""
    out = [f"s_add_u32 {s(S_KA)}, %[wl], {tile * K_TILE}",
           f"s_add_u32 {s(S_KB)}, {s(S_KA)}, 4096",
           f"s_lshr_b32 {s(S_T)}, %[wl], 7",                       # 8 w
           f"s_mul_i32 {s(S_VC)}, {s(S_T)}, %[kvp2]",               # Vt piece w: row 8 w
           f"s_lshl_b32 {s(S_T)}, %[kvp2], 5",
           f"s_add_u32 {s(S_VD)}, {s(S_VC)}, {s(S_T)}",             # Vt piece w + 4: 32 rows further
           f"s_mov_b32 {s(S_E)}, %[s4]"]
    if tile:
        out += [f"s_add_u32 {s(S_VC)}, {s(S_VC)}, {tile * 128}", f"s_add_u32 {s(S_VD)}, {s(S_VD)}, {tile * 128}",
                f"s_mul_i32 {s(S_T)}, %[st4], {tile}", f"s_add_u32 {s(S_E)}, {s(S_E)}, {s(S_T)}"]
    return out


def q_prefetch_hook(e_uid, npieces):
    """``npieces`` of the nine LDS-DMA pieces of the NEXT item's Q rows (this wave's 64 rows, image [chunk][row]); contiguous (a
    branch may not jump over MFMAs).  Issued in the LAST two tiles of an item, behind their tile pieces: counted waits are in issue
    order, so a Q piece (a cold HBM fetch) in front of a tile piece makes the next tile's wait sit out its whole latency; behind the
    last tile pieces of the item nothing waits on it before the next item's opening vmcnt(0)."""
    lines = [f"s_cmp_eq_u32 {s(S_QP)}, 0", f"s_cbranch_scc1 QSKIP_{e_uid}_%="]
    for _ in range(npieces):
        lines += [f"s_mov_b32 m0, {s(S_QM)}", f"s_add_u32 {s(S_QM)}, {s(S_QM)}, 1024",
                  f"buffer_load_dwordx4 %[qvo], %[rqn], {s(S_QO)} offen lds", f"s_add_u32 {s(S_QO)}, {s(S_QO)}, 16"]
    lines += [f"QSKIP_{e_uid}_%=:"]
    return "\n".join(lines)


def switch_block(e, tag):
    """out of line, once per item (four tiles before its end): from here on the LDS-DMA stream fetches the NEXT item's tiles 0..3"""
    e.label(f"SWITCH_{tag}_%=")
    e("s_cmp_eq_u32 %[hn], 0")
    e(f"s_cbranch_scc1 NONEXT_{tag}_%=")
    e(f"s_mov_b64 s[{DK}:{DK + 1}], %[kbn]")
    e(f"s_mov_b64 s[{DV}:{DV + 1}], %[vbn]")
    e("s_cmp_eq_u32 %[wl], 0")
    e(f"s_cselect_b64 s[{D4}:{D4 + 1}], %[kbn], %[vbn]")
    e(f"s_branch SOFF_{tag}_%=")
    e.label(f"NONEXT_{tag}_%=")       # no next item: zero-length descriptors, the pieces return zeros (same count of operations)
    for d in (DK, DV, D4):
        e(f"s_mov_b32 {s(d + 2)}, 0")
    e.label(f"SOFF_{tag}_%=")
    for ln in soff_reset(0):
        e(ln)
    e(f"s_branch BACKSW_{tag}_%=")


def stage_advance():
    """s[S_ST] = stage of tile i -> tile i + 1;  s[S_ST2] = stage of tile (i + 1) + 2"""
    return [
        f"s_add_u32 {s(S_ST)}, {s(S_ST)}, {STAGE}",
        f"s_cmp_eq_u32 {s(S_ST)}, {s(S_WRAP)}",
        f"s_cselect_b32 {s(S_ST)}, %[lb], {s(S_ST)}",
        f"s_add_u32 {s(S_ST2)}, {s(S_ST)}, {2 * STAGE}",
        f"s_sub_u32 {s(S_T)}, {s(S_ST2)}, {NSTAGE * STAGE}",
        f"s_cmp_ge_u32 {s(S_ST2)}, {s(S_WRAP)}",
        f"s_cselect_b32 {s(S_ST2)}, {s(S_T)}, {s(S_ST2)}",
    ]


def rescale_block(e, tag, buf):
    """out of line: raise the running max of both blocks by delta = max(mx, 0), rescale O, shift S(next) and -m"""
    e.label(f"RESC_{tag}_%=")
    e("s_nop 15")   # the last PV MFMA's result -> v_accvgpr_read
    for blk in ("A", "B"):
        e(f"v_max_f32_e32 {v(DL[blk])}, 0, {v(MX[blk])}")
    for blk in ("A", "B"):
        e(f"v_exp_f32_e64 {v(AL[blk])}, -{v(DL[blk])}")
    for blk in ("A", "B"):
        for i in range(16):
            e(f"v_sub_f32_e32 {v(NM[blk] + i)}, {v(NM[blk] + i)}, {v(DL[blk])}")
        for r in range(32):
            e(f"v_sub_f32_e32 {v(SBUF[buf][blk] + r)}, {v(SBUF[buf][blk] + r)}, {v(DL[blk])}")
    for blk in ("A", "B"):
        for r0 in range(0, 48, 4):
            for j in range(4):
                e(f"v_accvgpr_read_b32 {v(TMP[j])}, {a(O[blk] + r0 + j)}")
            for j in range(4):
                e(f"v_mul_f32_e32 {v(TMP[j])}, {v(TMP[j])}, {v(AL[blk])}")
            for j in range(4):
                e(f"v_accvgpr_write_b32 {a(O[blk] + r0 + j)}, {v(TMP[j])}")
    e("s_nop 7")
    e(f"s_branch BACK_{tag}_%=")


def body(e, tag, p, masked, resc, variant=0):
    """iteration i with S(i) in buffer p: X_i, barrier, Y_i.  ``resc`` collects the out-of-line rescale blocks to emit later.
    ``variant``: where the five LDS-DMA pieces of tile i+4 are issued (an LDS-DMA instruction holds its wave 25..185 cycles at issue
    depending on what else is in flight, MI355X_MICROARCH.md; with one wave per SIMD nothing covers that):
      0  behind the K(i+2) fragment reads, early in Y;   1  in the second half of Y, one per ~2.5 gaps, among the row-max VALU;
      3  the two K pieces in the second half of X (the K rows of stage(i) were last read two barriers ago), the three Vt / mixed
         pieces in Y.  (All five in X would race: other waves still read Vt(i) from that stage until the barrier — the emulator's
         early-landing mode catches exactly that.)
    Lab only (results NOT valid): 8 = no LDS-DMA in the loop, 9 = neither LDS-DMA nor the softmax VALU work."""
    cur, nxt = p, 1 - p
    nodma, novalu = variant in (8, 9), variant == 9
    dma = [] if nodma else dma_ops()
    dma_groups = [] if nodma else ([dma[0:4], dma[4:7], dma[7:10], dma[10:14]] if HD64 else [dma[0:4], dma[4:7], dma[7:10], dma[10:13], dma[13:17]])   # (M0 setup, load, soff advance) per piece
    flat = lambda gs: [t for g in gs for t in g]
    # ---- X
    if PHASE:
        e("s_memtime s[76:77]")
    e("s_waitcnt lgkmcnt(0)")        # K(i+1) fragments (read during Y_{i-1})
    units = [("A", 0, 0), ("A", 0, 1), ("A", 1, 0), ("A", 1, 1), ("B", 0, 0), ("B", 0, 1)]
    late = [("B", 1, 0), ("B", 1, 1)]
    if STATIC:   # in the order the PV MFMAs need them: unit j is read by MFMA 3 j of Y
        order = [("A", 0, 0), ("A", 0, 1), ("B", 0, 0), ("B", 0, 1), ("A", 1, 0), ("A", 1, 1), ("B", 1, 0), ("B", 1, 1)]
        units, late = order[:8 - NLATE], order[8 - NLATE:]
    if variant == 4:   # (d64: X has 16 MFMAs for the same softmax work) one more P unit moves to Y: A's keys 48..63, needed by its 16th MFMA
        units, late = units[:3] + units[4:], [("A", 1, 1)] + late
        variant = 1
    pu = []
    for blk, kt, cc in units:
        pu += p_unit(cur, blk, kt, cc)
    if novalu:
        pu = []
    kaddr = k_addr(S_ST2)
    nq = 4 * NCC     # MFMAs of X
    streams = [(v_reads(), 0, 7), (pu[:24], 0, 7), (pu[24:], 8, nq - 1), (kaddr, 10, 10 + len(kaddr) - 1)]
    if STATIC:
        streams = [(v_reads(), 0, 7), (pu, 0, nq - 1), (kaddr, 10, 10 + len(kaddr) - 1)]
    if variant == 3:
        streams.append((flat(dma_groups[:2]), 11, nq - 2))
    place(e, qk_mfmas(nxt), [st for st in streams if st[0]])
    if PHASE:
        e("s_memtime s[78:79]")
    e(f"s_waitcnt vmcnt({NPIECE}) lgkmcnt(0)" if not nodma else "s_waitcnt lgkmcnt(0)")   # Vt(i) fragments; this wave's pieces of tile i+2
    e("s_barrier")                       # ... and everybody else's; every wave is done reading stage(i)
    if PHASE:
        e("s_memtime s[80:81]")
    # ---- Y
    pv = [m for m, _, _ in pv_mfmas()]
    pb = [] if novalu else [t for u in late for t in p_unit(cur, *u)]
    ca, cb = max_chain(nxt, "A"), max_chain(nxt, "B")
    adv = stage_advance()
    chains = ca + [f"v_mov_b32_e32 {v(TT['A'])}, {v(MX['A'])}"] + cb
    tail = [
        f"v_permlane32_swap_b32_e32 {v(MX['A'])}, {v(TT['A'])}",
        f"v_mov_b32_e32 {v(TT['B'])}, {v(MX['B'])}",
        f"v_max_f32_e32 {v(MX['A'])}, {v(MX['A'])}, {v(TT['A'])}",
        adv[0],
        f"v_permlane32_swap_b32_e32 {v(MX['B'])}, {v(TT['B'])}",
        adv[1], adv[2],
        f"v_max_f32_e32 {v(MX['B'])}, {v(MX['B'])}, {v(TT['B'])}",
        adv[3], adv[4], adv[5], adv[6],
        f"v_max_f32_e32 {v(TMP[0])}, {v(MX['A'])}, {v(MX['B'])}",
        f"v_cmp_lt_f32_e32 vcc, 8.0, {v(TMP[0])}"]
    tail += [f"v_add_u32_e32 {v(VADDR[k])}, {s(S_ST)}, %[vfa{k}]" for k in range(4)]   # Vt read addresses of tile i+1 (stage advanced)
    if novalu:
        chains = []
        tail = [t for t in tail if t.startswith("s_") or t.startswith("v_add_u32") or t.startswith("v_cmp")]
        tail = [f"v_mov_b32_e32 {v(TMP[0])}, 0"] + tail
    streams = [(k_reads(), 0, 4), (pb, 0, 7 if len(late) == 2 else 11)]
    if STATIC:
        chains = []
        # (an s_cmp and the s_cselect that reads its SCC travel as ONE item: the LDS-DMA stream's s_add_i32 of the same gap would
        # otherwise land between them and overwrite SCC — the stage would never wrap)
        tail = [adv[0], adv[1] + "\n" + adv[2], adv[3], adv[4], adv[5] + "\n" + adv[6]] + [t for t in tail if t.startswith("v_add_u32")]
        streams = [(k_reads(), 0, 4), (pb, 0, 19)]
    if masked:
        streams.append((mask_ops(nxt), 2, 10))
    streams.append((chains, 5, 17))
    if PERSIST:
        assert variant == 1
    if variant == 0:
        streams.append((dma, 5, 9))
    elif variant == 1:
        streams.append((flat(dma_groups), 9, 21))
    elif variant == 3:
        streams.append((flat(dma_groups[2:]), 9, 20))
    if PERSIST and tag.startswith("m"):       # (the tail body of an item = its second-to-last tile)
        streams.append(([q_prefetch_hook(tag, 5)], 22, 22))
    streams.append((tail, 18, 23))
    place(e, pv, [st for st in streams if st[0]])
    if PHASE:
        def acc(dst, hi_pair, lo_pair):
            e(f"s_sub_u32 s90, s{hi_pair}, s{lo_pair}")
            e(f"s_subb_u32 s91, s{hi_pair + 1}, s{lo_pair + 1}")
            e(f"s_add_u32 s{dst}, s{dst}, s90")
            e(f"s_addc_u32 s{dst + 1}, s{dst + 1}, s91")
        e("s_memtime s[82:83]")
        e("s_waitcnt lgkmcnt(0)")
        acc(84, 78, 76)
        acc(86, 80, 78)
        acc(88, 82, 80)
    if not STATIC:
        e(f"s_cbranch_vccnz RESC_{tag}_%=")
        e.label(f"BACK_{tag}_%=")
        resc.append((tag, nxt))


def final(e, p):
    """last tile: P = exp2(S), O += Vt P^T, nothing for a next tile"""
    cur = p
    vr = v_reads()
    for t in vr:
        e(t)
    units = [("A", 0, 0), ("A", 0, 1), ("B", 0, 0), ("B", 0, 1), ("A", 1, 0), ("A", 1, 1), ("B", 1, 0), ("B", 1, 1)]
    pre = []
    for blk, kt, cc in units[:4]:
        pre += p_unit(cur, blk, kt, cc)
    for t in pre:
        e(t)
    e("s_waitcnt lgkmcnt(0)")
    if PERSIST:
        e("s_barrier")   # every wave has its Vt(n-1) fragments: the stage may now receive the next item's tile 3
    fill = []
    for blk, kt, cc in units[4:]:
        fill += [(t, 0) for t in p_unit(cur, blk, kt, cc)]
    if PERSIST:   # the next item's tile 3 into the stage of the tile being consumed (every wave has its Vt fragments: barrier above)
        fill += [(t, 0) for t in dma_ops()]
        fill.append((q_prefetch_hook(f"f{p}", 4), 0))
    schedule(e, [m for m, _, _ in pv_mfmas()], fill, cap=5, front=True)   # P of keys 32..63 is needed from the 13th MFMA on


def prologue_persist(e):
    """start of an item in the persistent form: tiles 0..3 are already in flight (issued by the previous item's tail or by the
    kernel's entry code), and so may be the previous item's output stores — one full drain, then the ring is at tile 0"""
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e("s_barrier")
    if PSTAMP:
        e("s_memtime s[76:77]")
    e(f"v_mov_b32_e32 {v(NEGBIG)}, 0xf149f2ca")
    for i in range(96):
        e(f"v_accvgpr_write_b32 {a(i)}, 0")
    for blk in ("A", "B"):
        for i in range(16):
            e(f"v_mov_b32_e32 {v(NM[blk] + i)}, " + (f"%[nm{blk.lower()}]" if STATIC else "0"))
    # descriptors of the tiles this item still has to fetch (4 .. n-1): its own K / Vt
    e(f"s_mul_i32 {s(S_KNR)}, %[kvp2], 72")
    e(f"s_mul_i32 {s(S_VNR)}, %[kvp2], 96")
    e(f"s_mov_b64 s[{DK}:{DK + 1}], %[kb]")
    e(f"s_mov_b32 {s(DK + 2)}, {s(S_KNR)}")
    e(f"s_mov_b32 {s(DK + 3)}, 0x20000")
    e(f"s_mov_b64 s[{DV}:{DV + 1}], %[vb]")
    e(f"s_mov_b32 {s(DV + 2)}, {s(S_VNR)}")
    e(f"s_mov_b32 {s(DV + 3)}, 0x20000")
    e("s_cmp_eq_u32 %[wl], 0")
    e(f"s_cselect_b64 s[{D4}:{D4 + 1}], %[kb], %[vb]")
    e(f"s_cselect_b32 {s(D4 + 2)}, {s(S_KNR)}, {s(S_VNR)}")
    e(f"s_mov_b32 {s(D4 + 3)}, 0x20000")
    for ln in soff_reset(4):
        e(ln)
    e(f"s_mul_i32 {s(S_QP)}, %[hn], 9")
    e(f"s_mov_b32 {s(S_QM)}, %[qlds]")
    e(f"s_mov_b32 {s(S_QO)}, 0")
    # The ring simply continues from item to item (the tail of the previous item put this item's tiles 0..3 into the four stages that
    # follow ITS last tile): tile 0 of the k-th item of a workgroup sits in stage (k ntiles) mod 4 — the C++ hands the three stage
    # addresses the prologue needs (tiles 0, 1, 2); any tile count >= 4 works, not only multiples of four.
    e(f"s_add_u32 {s(S_WRAP)}, %[lb], {NSTAGE * STAGE}")
    e(f"s_mov_b32 {s(S_ST)}, %[st0]")
    e(f"s_mov_b32 {s(S_ST2)}, %[st2]")
    e(f"s_sub_u32 {s(S_CNT)}, %[nt], 2")
    e(f"v_add_u32_e32 {v(KADDR)}, {s(S_ST)}, %[kfa]")
    for t in k_reads():
        e(t)
    e("s_waitcnt lgkmcnt(0)")
    for m in qk_mfmas(0):
        e(m)
    e(f"v_add_u32_e32 {v(KADDR)}, %[st1], %[kfa]")
    for t in k_reads():
        e(t)
    adopt(e)


def adopt(e):
    if STATIC:     # S(0) was computed against the caller's bound: nothing to adopt
        e("s_nop 15")
        for k in range(4):
            e(f"v_add_u32_e32 {v(VADDR[k])}, {s(S_ST)}, %[vfa{k}]")
        return
    e("s_nop 7")
    for t in max_chain(0, "A"):
        e(t)
    e(f"v_mov_b32_e32 {v(TT['A'])}, {v(MX['A'])}")
    for t in max_chain(0, "B"):
        e(t)
    e(f"v_permlane32_swap_b32_e32 {v(MX['A'])}, {v(TT['A'])}")
    e(f"v_mov_b32_e32 {v(TT['B'])}, {v(MX['B'])}")
    e(f"v_max_f32_e32 {v(MX['A'])}, {v(MX['A'])}, {v(TT['A'])}")
    e("s_nop 1")
    e(f"v_permlane32_swap_b32_e32 {v(MX['B'])}, {v(TT['B'])}")
    e(f"v_max_f32_e32 {v(MX['B'])}, {v(MX['B'])}, {v(TT['B'])}")
    for blk in ("A", "B"):
        for r in range(32):
            e(f"v_sub_f32_e32 {v(SBUF[0][blk] + r)}, {v(SBUF[0][blk] + r)}, {v(MX[blk])}")
        for i in range(16):
            e(f"v_sub_f32_e32 {v(NM[blk] + i)}, 0, {v(MX[blk])}")
    for k in range(4):
        e(f"v_add_u32_e32 {v(VADDR[k])}, {s(S_ST)}, %[vfa{k}]")


def prologue(e):
    if PERSIST:
        return prologue_persist(e)
    e(f"v_mov_b32_e32 {v(NEGBIG)}, 0xf149f2ca")
    for i in range(96):
        e(f"v_accvgpr_write_b32 {a(i)}, 0")
    for blk in ("A", "B"):
        for i in range(16):
            e(f"v_mov_b32_e32 {v(NM[blk] + i)}, " + (f"%[nm{blk.lower()}]" if STATIC else "0"))
    # DMA soff registers of tile 0
    e(f"s_mov_b32 {s(S_KA)}, %[wl]")
    e(f"s_add_i32 {s(S_KB)}, %[wl], 4096")
    e(f"s_mov_b32 {s(S_VC)}, %[sv0]")
    e(f"s_mov_b32 {s(S_VD)}, %[sv1]")
    if not HD64:
        e(f"s_mov_b32 {s(S_E)}, %[s4]")
    e(f"s_add_u32 {s(S_WRAP)}, %[lb], {NSTAGE * STAGE}")
    e(f"s_mov_b32 {s(S_ST)}, %[lb]")
    for t in range(4):    # tiles 0..3 into stages 0..3
        for ln in dma_ops():
            e(ln)
        e(f"s_add_u32 {s(S_ST)}, {s(S_ST)}, {STAGE}")
    e(f"s_mov_b32 {s(S_ST)}, %[lb]")
    e(f"s_add_u32 {s(S_ST2)}, %[lb], {2 * STAGE}")
    e(f"s_sub_u32 {s(S_CNT)}, %[nt], 2")
    # tile 0: K fragments, S(0) = K(0) Q^T (-m = 0), adopt its row max
    e(f"s_waitcnt vmcnt({3 * NPIECE})")
    e("s_barrier")
    for t in k_addr(S_ST):
        e(t)
    for t in k_reads():
        e(t)
    e("s_waitcnt lgkmcnt(0)")
    for m in qk_mfmas(0):
        e(m)
    e(f"s_waitcnt vmcnt({2 * NPIECE})")
    e("s_barrier")
    for t in k_addr(S_ST, STAGE):
        e(t)
    for t in k_reads():
        e(t)
    adopt(e)


def generate(variant=0, persist=False, pstamp=False, d64=False):
    global PERSIST, PSTAMP, HD64, NCC, NPIECE, K_TILE, STAGE, NV, PHASE, STATIC, NLATE
    assert not (d64 and persist)
    PERSIST, PSTAMP = persist, pstamp
    keep = (HD64, NCC, NPIECE, K_TILE, STAGE, NV)
    if d64:
        HD64, NCC, NPIECE, K_TILE, STAGE, NV = True, 4, 4, 8192, 8192 + 96 * 128, 213
    try:
        if variant == 10:
            PHASE = True
            return _generate(1)
        if variant == 5:
            STATIC, NLATE = True, (5 if d64 else 4)
            return _generate(1)
        return _generate(variant)
    finally:
        PERSIST = PSTAMP = PHASE = STATIC = False
        NLATE = 2
        HD64, NCC, NPIECE, K_TILE, STAGE, NV = keep


def _generate(variant):
    e = Emit()
    resc = []
    stamps = variant == 7      # lab: s_memtime at the start of the statement, at the loop head and at the end (variant 1 otherwise)
    if stamps:
        variant = 1
        e("s_memtime s[76:77]")
    prologue(e)
    if PHASE:
        for r in range(84, 90):
            e(f"s_mov_b32 s{r}, 0")
    if stamps or PSTAMP:
        e("s_memtime s[78:79]")

    def switch_check(tag):
        if PERSIST:   # four tiles before the end of the item the LDS-DMA stream turns to the next item
            e(f"s_cmp_eq_u32 {s(S_CNT)}, 2")
            e(f"s_cbranch_scc1 SWITCH_{tag}_%=")
            e.label(f"BACKSW_{tag}_%=")

    e.label("TOP_%=")
    e(f"s_cmp_eq_u32 {s(S_CNT)}, 0")
    e("s_cbranch_scc1 TAIL0_%=")
    switch_check("a")
    e(f"s_sub_u32 {s(S_CNT)}, {s(S_CNT)}, 1")
    body(e, "r0", 0, False, resc, variant)
    e(f"s_cmp_eq_u32 {s(S_CNT)}, 0")
    e("s_cbranch_scc1 TAIL1_%=")
    switch_check("b")
    e(f"s_sub_u32 {s(S_CNT)}, {s(S_CNT)}, 1")
    body(e, "r1", 1, False, resc, variant)
    e("s_branch TOP_%=")
    e.label("TAIL0_%=")
    body(e, "m0", 0, not PERSIST, resc, variant)   # (persistent form: whole tiles only, nothing to mask)
    final(e, 1)
    e("s_branch END_%=")
    e.label("TAIL1_%=")
    body(e, "m1", 1, not PERSIST, resc, variant)
    final(e, 0)
    e("s_branch END_%=")
    for tag, buf in resc:
        rescale_block(e, tag, buf)
    if PERSIST:
        switch_block(e, "a")
        switch_block(e, "b")
    e.label("END_%=")
    if not PERSIST:
        e("s_waitcnt vmcnt(0)")
    e("s_nop 15")    # the last PV MFMAs -> the v_accvgpr_read of the epilogue (a separate asm statement)
    if PSTAMP:
        e("s_waitcnt lgkmcnt(0)")
        e("s_mov_b64 %[t0], s[76:77]")
        e("s_mov_b64 %[t1], s[78:79]")
    if PHASE:
        e("s_mov_b64 %[t0], s[84:85]")
        e("s_mov_b64 %[t1], s[86:87]")
        e("s_mov_b64 %[t2], s[88:89]")
    if stamps:
        e("s_memtime s[80:81]")
        e("s_waitcnt lgkmcnt(0)")
        e("s_mov_b64 %[t0], s[76:77]")
        e("s_mov_b64 %[t1], s[78:79]")
        e("s_mov_b64 %[t2], s[80:81]")
    return e.lines


OPERANDS = ["rk", "rv", "r4", "wl", "sv0", "sv1", "s4", "st4", "l4", "lb", "nt", "lim", "kvo", "vvo", "v4o", "kfa", "vfa0", "vfa1",
            "vfa2", "vfa3"]


def clobbers(d64=False, phase=False):
    # (76..81: the stamp variants' s_memtime pairs; 82..91: the sums of the per-phase stamp variant 10 only — every SGPR listed here
    # is one the compiler cannot keep a value in across the statement)
    sregs = list(range(40, 55)) + list(range(60, 74)) + list(range(76, 92 if phase else 82))
    return [f"v{i}" for i in range(213 if d64 else NV)] + [f"a{i}" for i in range(NA)] + [f"s{i}" for i in sregs] + ["vcc", "memory"]


VARIANTS = (0, 1, 3, 5, 7, 8, 9, 10)   # 5: no running max (the caller's row bound); 7, 10: lab stamps; 8, 9: lab ablations (results not valid), compiled under VSYS_LAB only


def write_inc(path):
    with open(path, "w") as f:
        f.write("// GENERATED by csrc/gen/flash72_gen.py — do not edit (tests/test_flash72_asm_cpu.py regenerates and compares).\n")
        f.write("// The tile loop of flash_attn_d72_w64_kernel as ONE asm statement; register map and schedule: see the generator.\n")
        f.write("// FLASH72_W64_ASM_V<n>: placement variant n of the LDS-DMA pieces (same arithmetic, same bits).\n")
        for var in VARIANTS:
            lines = generate(var)
            f.write(f"#define FLASH72_W64_ASM_V{var} \\\n")
            for ln in lines:
                f.write('  "' + ln + '\\n\\t" \\\n')
            f.write('  ""\n')
        f.write("// FLASH72_W64P_ASM: the persistent form (a workgroup walks several (batch, head, query block) items: no LDS-DMA issue at the\n")
        f.write("// start of an item, the tail fetches the next item's first four tiles and its Q rows), placement variant 1.\n")
        f.write("#define FLASH72_W64P_ASM \\\n")
        for ln in generate(1, persist=True):
            f.write('  "' + ln + '\\n\\t" \\\n')
        f.write('  ""\n')
        f.write("#define FLASH72_W64P_ASM_S \\\n")     # the persistent form without the running max (variant 5)
        for ln in generate(5, persist=True):
            f.write('  "' + ln + '\\n\\t" \\\n')
        f.write('  ""\n')
        f.write("#define FLASH72_W64P_ASM_STAMP \\\n")
        for ln in generate(1, persist=True, pstamp=True):
            f.write('  "' + ln + '\\n\\t" \\\n')
        f.write('  ""\n')
        f.write("#define FLASH72_W64_CLOBBERS " + ", ".join('"' + c + '"' for c in clobbers()) + "\n")
        f.write("#define FLASH72_W64_CLOBBERS_PHASE " + ", ".join('"' + c + '"' for c in clobbers(phase=True)) + "\n")
        f.write("// FLASH64_W64_ASM: the head_dim 64 form (flash_attn_d64_w64_kernel, csrc/attention64_w64.hip), placement variant 1.\n")
        for var in (1, 4, 5):   # 4: one more P unit in the PV phase; 5: no running max (the QK^T phase has 16 MFMAs here for the same softmax work)
            f.write(f"#define FLASH64_W64_ASM_V{var} \\\n")
            for ln in generate(var, d64=True):
                f.write('  "' + ln + '\\n\\t" \\\n')
            f.write('  ""\n')
        f.write("#define FLASH64_W64_CLOBBERS " + ", ".join('"' + c + '"' for c in clobbers(True)) + "\n")
    return generate(0)


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(here), "flash72_w64_asm.inc")
    lines = write_inc(out)
    n_mfma = sum(1 for ln in lines if ln.startswith("v_mfma"))
    print(f"{out}: {len(lines)} lines, {n_mfma} MFMA")
