#!/usr/bin/env python
"""A small gfx950 wave emulator for the hand-written instruction streams of this repo (TEST INFRASTRUCTURE, CPU only).

It executes the TEXT an asm generator emits (csrc/gen/flash72_gen.py) for all waves of one workgroup — 64 lanes x 512 registers
per wave in numpy, one shared LDS image, s_barrier rendezvous — and checks what a GPU run cannot show reliably:

  * data flow and fragment layouts (v_mfma_f32_32x32x16_bf16 operand / result maps, v_permlane32_swap, v_cvt_pk_bf16_f32, LDS-DMA
    lane-linear destination, buffer bounds);
  * counted waits: an LDS-DMA piece becomes visible in LDS either AT ISSUE ("early") or only when the issuing wave executes an
    s_waitcnt vmcnt(N) that covers it ("late"); a ds_read result lands at issue or at the covering lgkmcnt wait, and a register
    with a read still in flight is POISON (NaN pattern) — a schedule is accepted only if both extremes give the right answer,
    for both orders in which the waves of the workgroup are stepped between barriers;
  * the software-visible hazards of the stream (wait states the assembler does not insert: MI355X guide §5.7 item 2; LLVM
    GCNHazardRecognizer gfx940 / gfx950 rules): MFMA result -> any non-accumulate use (12), VALU write -> MFMA operand (2),
    VALU write -> v_permlane32_swap (2), transcendental -> VALU use (1), M0 write -> LDS-DMA (1), MFMA C read -> VALU overwrite (13).

Only the instructions the generators emit are implemented; anything else raises."""
import re

import numpy as np

POISON = np.uint32(0x7FC0DEAD)


def f32(u):
    return u.view(np.float32)


def u32(f):
    return np.asarray(f, dtype=np.float32).view(np.uint32)


def bf16_round(x):
    """fp32 array -> bf16 bits (round to nearest even), as uint32 in the low 16 bits"""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    lsb = (u >> 16) & 1
    r = (u + 0x7FFF + lsb) >> 16
    nan = np.isnan(np.asarray(x, dtype=np.float32))
    r = np.where(nan, 0x7FC0, r)
    return (r & 0xFFFF).astype(np.uint32)


def bf16_to_f32(h):
    return (np.asarray(h, dtype=np.uint32) << 16).view(np.float32)


class Rsrc:
    def __init__(self, data_u8, num_records):
        self.data = data_u8
        self.n = num_records


class Hazard(Exception):
    pass


REG = re.compile(r"^([vas])(\d+)$|^([vas])\[(\d+):(\d+)\]$")


def parse_reg(tok):
    m = REG.match(tok)
    if not m:
        return None
    if m.group(1):
        return m.group(1), int(m.group(2)), 1
    return m.group(3), int(m.group(4)), int(m.group(5)) - int(m.group(4)) + 1


INLINE_F = {"0.5": 0.5, "1.0": 1.0, "2.0": 2.0, "4.0": 4.0, "8.0": 8.0, "-1.0": -1.0, "-0.5": -0.5, "-2.0": -2.0, "-4.0": -4.0}


class Wave:
    def __init__(self, wg, wid, binds):
        self.wg, self.wid = wg, wid
        self.v = np.full((256, 64), POISON, dtype=np.uint32)
        self.a = np.full((256, 64), POISON, dtype=np.uint32)
        self.s = {}
        self.m0 = 0
        self.vcc = np.zeros(64, dtype=bool)
        self.scc = False
        self.pc = 0
        self.t = 0                      # wait-state clock
        self.pend_ds = []               # (file, reg0, data[4][64]) in issue order
        self.pend_vm = []               # (lds byte addresses [64], data [64][16] uint8) in issue order
        self.binds = binds
        self.done = False
        self.nbarrier = 0
        # hazard bookkeeping: (file, idx) -> time
        self.w_mfma, self.w_valu, self.w_trans, self.r_mfmac = {}, {}, {}, {}
        self.m0_t = -100
        self.violations = []
        self.count = {}

    # ------------------------------------------------------------------ operand access
    def file(self, f):
        return self.v if f == "v" else self.a

    def src_u32(self, tok, as_float=False):
        """vector source operand -> uint32[64]"""
        neg = tok.startswith("-") and parse_reg(tok[1:]) is not None
        if neg:
            tok = tok[1:]
        r = parse_reg(tok)
        if r is not None:
            f, i, n = r
            assert n == 1
            if f == "s":
                val = np.full(64, self.s[i], dtype=np.uint32)
            else:
                self.check_read(f, i, 1)
                val = self.file(f)[i].copy()
        elif tok in INLINE_F and as_float:
            val = np.full(64, u32(np.float32(INLINE_F[tok])), dtype=np.uint32)
        elif tok.startswith("0x"):
            val = np.full(64, int(tok, 16), dtype=np.uint32)
        else:
            iv = int(tok)
            if as_float:
                assert iv == 0, f"integer {iv} as a float operand"
            val = np.full(64, iv & 0xFFFFFFFF, dtype=np.uint32)
        if neg:
            val = val ^ np.uint32(0x80000000)
        return val

    def ssrc(self, tok):
        r = parse_reg(tok)
        if r is not None:
            assert r[0] == "s" and r[2] == 1, tok
            return int(self.s[r[1]])
        if tok == "m0":
            return self.m0
        if tok.startswith("0x"):
            return int(tok, 16)
        return int(tok)

    # ------------------------------------------------------------------ hazards
    def viol(self, msg):
        self.violations.append(f"wave {self.wid} pc {self.pc}: {msg}: {self.wg.prog[self.pc][0]}")

    def check_read(self, f, i, n, kind="valu"):
        for k in range(i, i + n):
            key = (f, k)
            if kind != "mfma_c_same":
                tw = self.w_mfma.get(key)
                if tw is not None and self.t - tw - 1 < 12:
                    self.viol(f"{f}{k} read {self.t - tw - 1} wait states after the MFMA that writes it (need 12)")
            if kind in ("mfma", "mfma_c_same"):
                tv = self.w_valu.get(key)
                if tv is not None and self.t - tv - 1 < 2:
                    self.viol(f"{f}{k} is an MFMA operand {self.t - tv - 1} wait states after a VALU write (need 2)")
            if kind == "permlane":
                tv = self.w_valu.get(key)
                if tv is not None and self.t - tv - 1 < 2:
                    self.viol(f"{f}{k} read by v_permlane32_swap {self.t - tv - 1} wait states after a VALU write (need 2)")
            if kind in ("valu", "permlane"):
                tt = self.w_trans.get(key)
                if tt is not None and self.t - tt - 1 < 1:
                    self.viol(f"{f}{k} (transcendental result) used by the next VALU instruction (need 1 wait state)")
            if self.reg_pending(f, k):
                self.viol(f"{f}{k} read while a ds_read into it is in flight")

    def note_write(self, f, i, n, kind):
        for k in range(i, i + n):
            key = (f, k)
            if kind != "mfma":
                tw = self.w_mfma.get(key)
                if tw is not None and self.t - tw - 1 < 12:
                    self.viol(f"{f}{k} overwritten {self.t - tw - 1} wait states after the MFMA that writes it (need 12)")
                tc = self.r_mfmac.get(key)
                if tc is not None and self.t - tc - 1 < 13:
                    self.viol(f"{f}{k} overwritten {self.t - tc - 1} wait states after an MFMA read it as C (need 13)")
            self.w_mfma.pop(key, None)
            self.w_valu.pop(key, None)
            self.w_trans.pop(key, None)
            if kind == "mfma":
                self.w_mfma[key] = self.t
            elif kind == "valu":
                self.w_valu[key] = self.t
            elif kind == "trans":
                self.w_valu[key] = self.t
                self.w_trans[key] = self.t

    def reg_pending(self, f, k):
        for pf, r0, _ in self.pend_ds:
            if pf == f and r0 <= k < r0 + 4:
                return True
        return False

    # ------------------------------------------------------------------ waits
    def wait_lgkm(self, n):
        while len(self.pend_ds) > n:
            f, r0, data = self.pend_ds.pop(0)
            self.file(f)[r0:r0 + 4] = data

    def wait_vm(self, n):
        while len(self.pend_vm) > n:
            addr, data = self.pend_vm.pop(0)
            self.wg.lds_write(addr, data)

    # ------------------------------------------------------------------ execution
    def run_until_barrier(self):
        prog = self.wg.prog
        while True:
            if self.pc >= len(prog):
                if getattr(self, "drain_at_end", True):
                    self.wait_vm(0)
                    self.wait_lgkm(0)
                self.done = True
                return "done"
            text, op, args = prog[self.pc]
            self.count[op] = self.count.get(op, 0) + 1
            r = self.step(op, args)
            self.t += 1
            if r == "barrier":
                self.pc += 1
                self.nbarrier += 1
                return "barrier"
            if r is None:
                self.pc += 1

    def step(self, op, A):
        v, wg = self.v, self.wg
        if op == "s_nop":
            self.t += int(A[0])
            return
        if op == "s_waitcnt":
            for part in A:
                m = re.match(r"(vmcnt|lgkmcnt)\((\d+)\)", part)
                assert m, part
                (self.wait_vm if m.group(1) == "vmcnt" else self.wait_lgkm)(int(m.group(2)))
            return
        if op == "s_barrier":
            return "barrier"
        if op in ("s_mov_b32", "s_add_i32", "s_add_u32", "s_sub_u32"):
            if op == "s_mov_b32":
                val = self.ssrc(A[1])
            elif op == "s_sub_u32":
                val = self.ssrc(A[1]) - self.ssrc(A[2])
            else:
                val = self.ssrc(A[1]) + self.ssrc(A[2])
            # SCC as the hardware sets it (carry / borrow / signed overflow): an s_cselect separated from its s_cmp by one of these
            # reads THIS, which is what a placement that splits the pair gets on the GPU
            if op == "s_add_u32":
                self.scc = val > 0xFFFFFFFF
            elif op == "s_sub_u32":
                self.scc = val < 0
            elif op == "s_add_i32":
                x_, y_ = self.ssrc(A[1]) & 0xFFFFFFFF, self.ssrc(A[2]) & 0xFFFFFFFF
                sx, sy, sr = x_ >> 31, y_ >> 31, ((x_ + y_) & 0xFFFFFFFF) >> 31
                self.scc = bool(sx == sy and sr != sx)
            val &= 0xFFFFFFFF
            if A[0] == "m0":
                self.m0 = val
                self.m0_t = self.t
            else:
                self.s[parse_reg(A[0])[1]] = val
            return
        if op in ("s_mul_i32", "s_lshr_b32", "s_lshl_b32"):
            x, y = self.ssrc(A[1]), self.ssrc(A[2])
            val = {"s_mul_i32": x * y, "s_lshr_b32": x >> (y & 31), "s_lshl_b32": x << (y & 31)}[op] & 0xFFFFFFFF
            if op != "s_mul_i32":
                self.scc = val != 0
            self.s[parse_reg(A[0])[1]] = val
            return
        if op in ("s_mov_b64", "s_cselect_b64"):
            _, d0, n = parse_reg(A[0])
            assert n == 2
            src = A[1] if (op == "s_mov_b64" or self.scc) else A[2]
            _, s0, n2 = parse_reg(src)
            assert n2 == 2
            self.s[d0], self.s[d0 + 1] = self.s[s0], self.s[s0 + 1]
            return
        if op in ("s_cmp_eq_u32", "s_cmp_ge_u32", "s_cmp_lt_u32"):
            x, y = self.ssrc(A[0]), self.ssrc(A[1])
            self.scc = {"s_cmp_eq_u32": x == y, "s_cmp_ge_u32": x >= y, "s_cmp_lt_u32": x < y}[op]
            return
        if op == "s_cselect_b32":
            self.s[parse_reg(A[0])[1]] = self.ssrc(A[1]) if self.scc else self.ssrc(A[2])
            return
        if op in ("s_branch", "s_cbranch_scc1", "s_cbranch_scc0", "s_cbranch_vccnz", "s_cbranch_vccz"):
            take = {"s_branch": True, "s_cbranch_scc1": self.scc, "s_cbranch_scc0": not self.scc,
                    "s_cbranch_vccnz": bool(self.vcc.any()), "s_cbranch_vccz": not self.vcc.any()}[op]
            if take:
                self.pc = wg.labels[A[0]]
                return "jump"
            return
        # ---- VALU
        if op in ("v_mov_b32_e32", "v_accvgpr_write_b32", "v_accvgpr_read_b32"):
            f, i, _ = parse_reg(A[0])
            val = self.src_u32(A[1])
            self.note_write(f, i, 1, "valu")
            self.file(f)[i] = val
            return
        if op == "v_add_u32_e32":
            f, i, _ = parse_reg(A[0])
            val = (self.src_u32(A[1]).astype(np.uint64) + self.src_u32(A[2])).astype(np.uint32)
            self.note_write(f, i, 1, "valu")
            v[i] = val
            return
        if op in ("v_sub_f32_e32", "v_mul_f32_e32", "v_max_f32_e32", "v_add_f32_e32"):
            f, i, _ = parse_reg(A[0])
            x, y = f32(self.src_u32(A[1], True)), f32(self.src_u32(A[2], True))
            with np.errstate(all="ignore"):
                r = {"v_sub_f32_e32": x - y, "v_mul_f32_e32": x * y, "v_add_f32_e32": x + y,
                     "v_max_f32_e32": np.fmax(x, y)}[op]
            self.note_write(f, i, 1, "valu")
            v[i] = u32(r.astype(np.float32))
            return
        if op == "v_max3_f32":
            f, i, _ = parse_reg(A[0])
            x = np.fmax(np.fmax(f32(self.src_u32(A[1], True)), f32(self.src_u32(A[2], True))), f32(self.src_u32(A[3], True)))
            self.note_write(f, i, 1, "valu")
            v[i] = u32(x)
            return
        if op in ("v_exp_f32_e32", "v_exp_f32_e64", "v_rcp_f32_e32"):
            f, i, _ = parse_reg(A[0])
            x = f32(self.src_u32(A[1], True))
            with np.errstate(all="ignore"):
                r = np.exp2(x.astype(np.float64)).astype(np.float32) if op.startswith("v_exp") else (1.0 / x).astype(np.float32)
            self.note_write(f, i, 1, "trans")
            v[i] = u32(r)
            return
        if op == "v_cvt_pk_bf16_f32":
            f, i, _ = parse_reg(A[0])
            lo, hi = bf16_round(f32(self.src_u32(A[1], True))), bf16_round(f32(self.src_u32(A[2], True)))
            self.note_write(f, i, 1, "valu")
            v[i] = lo | (hi << 16)
            return
        if op == "v_permlane32_swap_b32_e32":
            fd, d, _ = parse_reg(A[0])
            fs, s_, _ = parse_reg(A[1])
            self.check_read("v", d, 1, "permlane")
            self.check_read("v", s_, 1, "permlane")
            dv, sv = v[d].copy(), v[s_].copy()
            nd, ns = dv.copy(), sv.copy()
            nd[32:] = sv[:32]
            ns[:32] = dv[32:]
            self.note_write("v", d, 1, "valu")
            self.note_write("v", s_, 1, "valu")
            v[d], v[s_] = nd, ns
            return
        if op in ("v_cmp_lt_f32_e32", "v_cmp_lt_i32_e32", "v_cmp_gt_f32_e32"):
            assert A[0] == "vcc"
            if op.endswith("i32_e32"):
                x, y = self.src_u32(A[1]).view(np.int32), self.src_u32(A[2]).view(np.int32)
            else:
                x, y = f32(self.src_u32(A[1], True)), f32(self.src_u32(A[2], True))
            with np.errstate(all="ignore"):
                self.vcc = (x < y) if "_lt_" in op else (x > y)
            return
        if op == "v_cndmask_b32_e32":
            f, i, _ = parse_reg(A[0])
            assert A[3] == "vcc"
            x, y = self.src_u32(A[1]), self.src_u32(A[2])
            self.note_write(f, i, 1, "valu")
            v[i] = np.where(self.vcc, y, x)
            return
        if op == "v_mfma_f32_32x32x16_bf16":
            return self.mfma(A)
        if op == "ds_read_b128":
            f, r0, n = parse_reg(A[0])
            assert n == 4
            _, ai, _ = parse_reg(A[1])
            off = 0
            for extra in A[2:]:
                m = re.match(r"offset:(\d+)", extra)
                assert m, extra
                off = int(m.group(1))
                assert off < 65536
            self.check_read("v", ai, 1, "addr")
            addr = v[ai].astype(np.int64) + off
            assert (addr % 16 == 0).all(), "ds_read_b128 off its natural alignment"
            data = wg.lds_read16(addr)                   # [64][16] uint8 -> 4 dwords per lane
            words = data.view(np.uint32).reshape(64, 4).T.copy()
            self.note_write(f, r0, 4, "load")
            if wg.late_ds:
                self.file(f)[r0:r0 + 4] = POISON
                self.pend_ds.append((f, r0, words))
            else:
                self.file(f)[r0:r0 + 4] = words
            return
        if op == "buffer_load_dwordx4":
            assert A[-1] == "lds" and A[-2] == "offen", A
            _, vo, _ = parse_reg(A[0])
            _, rs, n = parse_reg(A[1])
            assert n == 4
            if self.t - self.m0_t - 1 < 1:
                self.viol("LDS-DMA issued right behind the M0 write (need 1 wait state)")
            # descriptor = 4 SGPRs: words 0..1 name a buffer of the table (a fake address), word 2 = num_records
            buf = wg.bufs[(int(self.s[rs]), int(self.s[rs + 1]))]
            nrec = int(self.s[rs + 2])
            soff = self.ssrc(A[2])
            goff = v[vo].astype(np.int64) + soff
            data = np.zeros((64, 16), dtype=np.uint8)
            for l in range(64):
                if 0 <= goff[l] and goff[l] + 16 <= nrec:
                    data[l] = buf[goff[l]:goff[l] + 16]
            addr = self.m0 + 16 * np.arange(64, dtype=np.int64)   # (gfx950: M0 carries the full LDS byte address; the GEMM kernels of this repo DMA beyond 64 KiB)
            if wg.late_vm:
                self.pend_vm.append((addr, data))
            else:
                wg.lds_write(addr, data)
                self.pend_vm.append((addr, None))
            return
        raise NotImplementedError(op + " " + ", ".join(A))

    def frag_ab(self, f, r0):
        """4 registers of 2 bf16 each per lane -> matrix [32 (row or col = lane & 31)][16 (k = 8 (lane >> 5) + j)] fp32"""
        regs = self.file(f)[r0:r0 + 4]                   # [4][64]
        out = np.zeros((32, 16), dtype=np.float32)
        lanes = np.arange(64)
        for w in range(4):
            lo = bf16_to_f32(regs[w] & 0xFFFF)
            hi = bf16_to_f32(regs[w] >> 16)
            out[lanes & 31, 8 * (lanes >> 5) + 2 * w] = lo
            out[lanes & 31, 8 * (lanes >> 5) + 2 * w + 1] = hi
        return out

    def mfma(self, A):
        fd, d0, nd = parse_reg(A[0])
        fa, a0, na = parse_reg(A[1])
        fb, b0, nb = parse_reg(A[2])
        fc, c0, nc = parse_reg(A[3])
        assert nd == nc == 16 and na == nb == 4
        assert fd == fc, "MFMA C and D must both be VGPRs or both AGPRs (one acc_cd bit)"
        self.check_read(fa, a0, 4, "mfma")
        self.check_read(fb, b0, 4, "mfma")
        same = (fc, c0) == (fd, d0)
        self.check_read(fc, c0, 16, "mfma_c_same" if same else "mfma")
        if not same:
            for k in range(c0, c0 + 16):
                self.r_mfmac[(fc, k)] = self.t
            assert c0 + 16 <= d0 or d0 + 16 <= c0 or fc != fd, "partially overlapping C / D"
        Am, Bm = self.frag_ab(fa, a0), self.frag_ab(fb, b0)      # A[row][k], B[col][k]
        lanes = np.arange(64)
        C = np.zeros((32, 32), dtype=np.float32)
        cregs = self.file(fc)[c0:c0 + 16]
        for r in range(16):
            rows = (r & 3) + 8 * (r >> 2) + 4 * (lanes >> 5)
            C[rows, lanes & 31] = f32(cregs[r])
        with np.errstate(all="ignore"):
            D = (Am.astype(np.float64) @ Bm.astype(np.float64).T + C).astype(np.float32)
        self.note_write(fd, d0, 16, "mfma")
        dregs = self.file(fd)
        for r in range(16):
            rows = (r & 3) + 8 * (r >> 2) + 4 * (lanes >> 5)
            dregs[d0 + r] = u32(D[rows, lanes & 31])
        return


class Workgroup:
    """``lines``: asm text lines (labels end with ':'), ``binds``: operand name -> register text for the %[name] placeholders."""

    def __init__(self, lines, binds, nwaves=4, lds_bytes=160 * 1024, late_vm=True, late_ds=True):
        self.late_vm, self.late_ds = late_vm, late_ds
        self.lds = np.zeros(lds_bytes, dtype=np.uint8)
        self.bufs = {}                  # (address word 0, word 1) -> uint8 array (buffer descriptors name them)
        self.labels = {}
        self.prog = []
        for ln in lines:
            ln = ln.strip()
            if not ln or ln.startswith(";"):
                continue
            ln = ln.replace("_%=", "")
            if ln.endswith(":"):
                self.labels[ln[:-1]] = len(self.prog)
                continue
            self.prog.append(ln)
        self.raw = self.prog
        self.waves = [Wave(self, w, binds[w]) for w in range(nwaves)]
        self.decoded = {}

    def decode(self, wave):
        out = []
        for ln in self.raw:
            t = ln
            for name, reg in wave.binds.items():
                t = t.replace("%[" + name + "]", reg)
            assert "%[" not in t, t
            op, _, rest = t.partition(" ")
            args = [x.strip() for x in rest.split(",")] if rest.strip() else []
            # trailing modifiers separated by spaces ("offen lds", "offset:32")
            if args:
                tail = args[-1].split()
                args = args[:-1] + tail
            out.append((ln, op, args))
        return out

    def lds_write(self, addr, data):
        if data is None or addr is None:
            return
        for l in range(64):
            self.lds[addr[l]:addr[l] + 16] = data[l]

    def lds_read16(self, addr):
        out = np.zeros((64, 16), dtype=np.uint8)
        for l in range(64):
            out[l] = self.lds[addr[l]:addr[l] + 16]
        return out

    def load(self, lines):
        """another asm statement on the same workgroup state (registers, LDS, operations in flight carry over)"""
        self.labels = {}
        self.raw = []
        for ln in lines:
            ln = ln.strip()
            if not ln or ln.startswith(";"):
                continue
            ln = ln.replace("_%=", "")
            if ln.endswith(":"):
                self.labels[ln[:-1]] = len(self.raw)
                continue
            self.raw.append(ln)
        for w in self.waves:
            w.pc, w.done, w.drain_at_end = 0, False, False

    def run(self, order=None, max_rounds=100000):
        order = order or list(range(len(self.waves)))
        progs = {w.wid: self.decode(w) for w in self.waves}
        for _ in range(max_rounds):
            states = []
            for wid in order:
                w = self.waves[wid]
                if w.done:
                    states.append("done")
                    continue
                self.prog = progs[wid]
                states.append(w.run_until_barrier())
            if all(s == "done" for s in states):
                break
            assert all(s == "barrier" for s in states), f"waves disagree at a barrier: {states}"
        else:
            raise RuntimeError("emulation did not terminate")
        viol = [x for w in self.waves for x in w.violations]
        return viol
