"""CPU emulation of the index math of csrc/attention_t_mfma.hip (the lab MFMA temporal-attention kernel): lanes, fragments and
accumulators are simulated with the v_mfma_f32_32x32x16_bf16 layout conventions the shipped kernels are tested with
(A[m][k]: lane m = l%32 holds k = 8*(l//32)+e; B[k][n]: lane n = l%32, same k; D[m][n]: lane (n = l%32, hi = l//32), register
4g+r <-> m = 8g + 4hi + r).  Checks the key permutation, the P^T -> B-fragment identity, the V^T column reads and the output map
against plain attention.   python tools/emulate_t_mfma.py"""
import numpy as np


def mfma(A_frag, B_frag, acc):
    """A_frag/B_frag: [64 lanes][8]; acc: [64][16].  One 32x32x16 step."""
    A = np.zeros((32, 16)); Bm = np.zeros((16, 32))
    for l in range(64):
        m, hi = l % 32, l // 32
        A[m, 8 * hi:8 * hi + 8] = A_frag[l]
        Bm[8 * hi:8 * hi + 8, m] = B_frag[l]
    D = A @ Bm
    out = acc.copy()
    for l in range(64):
        n, hi = l % 32, l // 32
        for g in range(4):
            for r in range(4):
                out[l, 4 * g + r] += D[8 * g + 4 * hi + r, n]
    return out


def run(T, seed=0):
    rng = np.random.default_rng(seed)
    HD = 72
    q, k, v = (rng.standard_normal((T, HD)) for _ in range(3))
    scale = HD ** -0.5
    s = (q @ k.T) * scale
    p = np.exp(s - s.max(1, keepdims=True)); p /= p.sum(1, keepdims=True)
    ref = p @ v
    sigma = lambda m: 16 * (m >> 4) + 8 * ((m >> 2) & 1) + 4 * ((m >> 3) & 1) + (m & 3)
    assert sorted(sigma(m) for m in range(32)) == list(range(32))
    # per-lane loads (5 chunks of 8 dims at 16c + 8hi), zero beyond T / beyond 72
    def frag(mat, frame_of_lane):
        f = np.zeros((5, 64, 8))
        for l in range(64):
            l31, hi = l % 32, l // 32
            fr = frame_of_lane(l31)
            for c in range(5):
                d0 = 16 * c + 8 * hi
                if d0 < HD and fr < T:
                    f[c, l] = mat[fr, d0:d0 + 8]
        return f
    qf, kf = frag(q, lambda m: m), frag(k, sigma)
    vt = np.zeros((32, 80))
    vt[:T, :HD] = v                                    # the LDS tile, row-major
    sacc = np.zeros((64, 16))
    for c in range(5):
        sacc = mfma(kf[c], qf[c], sacc)
    out = np.zeros((T, HD))
    P = np.zeros((64, 16)); L = np.zeros(64); M = np.full(64, -1e30)
    for l in range(64):
        hi = l // 32
        for g in range(4):
            for r in range(4):
                key = 16 * (g >> 1) + 8 * hi + 4 * (g & 1) + r
                sacc[l, 4 * g + r] = sacc[l, 4 * g + r] * scale if key < T else -1e30
        M[l] = sacc[l].max()
    for l in range(64):
        m = max(M[l], M[l ^ 32])
        P[l] = np.exp(sacc[l] - m)
        L[l] = P[l].sum()
    Ltot = np.array([L[l] + L[l ^ 32] for l in range(64)])
    oacc = np.zeros((3, 64, 16))
    for blk in range(3):
        for s2 in range(2):
            A_frag = np.zeros((64, 8)); B_frag = np.zeros((64, 8))
            for l in range(64):
                l31, hi = l % 32, l // 32
                d = 32 * blk + l31
                B_frag[l] = P[l, 8 * s2:8 * s2 + 8]
                if d < 80:
                    for e in range(8):
                        A_frag[l, e] = vt[16 * s2 + 8 * hi + e, d]
            oacc[blk] = mfma(A_frag, B_frag, oacc[blk])
    for l in range(64):
        i, hi = l % 32, l // 32
        if i >= T:
            continue
        for blk in range(3):
            for g in range(4):
                d0 = 32 * blk + 8 * g + 4 * hi
                if d0 < HD:
                    out[i, d0:d0 + 4] = oacc[blk, l, 4 * g:4 * g + 4] / Ltot[l]
    err = np.abs(out - ref).max()
    return err


if __name__ == "__main__":
    for T in (1, 5, 19, 32):
        e = run(T, seed=T)
        print(f"T = {T:2d}: max |emulated - reference| = {e:.3e}")
        assert e < 1e-10
    print("index math of attention_t_mfma.hip is consistent")
