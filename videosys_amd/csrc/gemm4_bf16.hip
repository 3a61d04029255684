// bf16 MFMA GEMM, "ping-pong" schedule: the two waves of every SIMD alternate between a LOAD segment and an MFMA segment.
//
// Same contract, block tile (256 x 192 x 64), wave tile (64 x 96 = 2 x 3 v_mfma_f32_32x32x16_bf16, issued "swapped"), LDS
// image (128-byte rows, 16-byte slot XOR-swizzled by (row>>1)&7 on the SOURCE side of the LDS-DMA) and k order as
// gemm_bf16.hip — results are bit-identical to it.  What changes is WHEN each wave does what.
//
// In gemm_bf16.hip both waves of a SIMD belong to the same phase: they read fragments together, multiply together and sit in
// the tile barrier together, and the matrix pipe idles through every read burst and barrier (cycle stamps, DESIGN.md §3.1:
// 2354 cycles per stage for 1536 cycles of MFMA).  Here the 8 waves form two GROUPS — G0 = waves 0-3 = token rows 0..127 of
// the tile, G1 = waves 4-7 = rows 128..255; a SIMD hosts one wave of each — and the workgroup barrier sequence makes them
// alternate (MI355X_MICROARCH.md "Two waves per SIMD"): in every interval between two barriers one group issues its 12 MFMAs
// of a half K-tile at s_setprio 1 while the other group reads the fragments of ITS next half K-tile and issues LDS-DMA:
//
//     interval   4t          4t+1         4t+2         4t+3
//     G0         LOAD(2t)    MFMA(2t)     LOAD(2t+1)   MFMA(2t+1)          (h = 2t, 2t+1: the halves of K-tile t)
//     G1         MFMA(2t-1)  LOAD(2t)     MFMA(2t)     LOAD(2t+1)
//
// LOAD(2t) reads ALL twelve W fragments of K-tile t (48 VGPRs) and the A fragments of its first half; LOAD(2t+1) reads the
// second half's A fragments.  Reading W early frees its slot after the first half of the tile, which is what gives the
// two-slot W ring a lead of >= 3 intervals.  The A operand is PRIVATE to a group (rows of the tile), so its three-slot ring
// needs no cross-group hand-off.  LDS: 3 x 32 KiB (A) + 2 x 24 KiB (W) = 144 KiB, one workgroup per CU.
//
// LDS-DMA issue points and counted waits (a piece = 1 KiB = 8 rows x 128 B; per K-tile a wave stages 4 A + 3 W pieces):
//   G0: LOAD(2t)   issues W(t+1) -> W slot (t+1)%2     [last read by G1 in interval 4t-3]
//       LOAD(2t+1) issues A(t+2) -> A slot (t+2)%3     [its own rows, last read in interval 4t-2]
//       end of MFMA(2t+1): s_waitcnt vmcnt(4)  = "everything but A(t+2) has landed" -> K-tile t+1 complete
//   G1: LOAD(2t+1) issues A(t+2), W(t+2) -> W slot t%2 [last read (by G1 itself) in interval 4t+1, retired before barrier 4t+3]
//       end of LOAD(2t+1): s_waitcnt vmcnt(7)  = "everything but A(t+2), W(t+2)" -> K-tile t+1 complete
//   Both waits sit in front of barrier 4t+4; the first read of K-tile t+1 (G0, LOAD(2t+2)) comes after it.  Nothing else orders
//   a ds_read behind an LDS-DMA (MI355X_MICROARCH.md item 7), and raw s_barrier is used throughout: __syncthreads() would drain
//   the DMA queue.
//
// Epilogue without LDS: after v_permlane32_swap of column-group pairs a lane holds 16 contiguous bytes of ONE output row
// (cdna_hip_programming.md T21), so bias / GELU / gate / residual / PAB copy all happen in the accumulator layout and the
// tile leaves in 12 dwordx4 stores per lane.  G0's epilogue overlaps G1's last MFMA segment.
#include "common.h"
#include "vsys_internal.h"

namespace vsys {
namespace {

constexpr int BM = 256, BN = 192, BK = 64;
constexpr int A_BYTES = BM * BK * 2;   // 32768
constexpr int B_BYTES = BN * BK * 2;   // 24576
constexpr int W_BASE = 3 * A_BYTES;    // 98304
constexpr int LDS_BYTES = W_BASE + 2 * B_BYTES;  // 147456
constexpr int NA = 4, NB = 3;

#define G4_SB() __builtin_amdgcn_sched_barrier(0)
#define G4_BAR()                        \
  do {                                  \
    G4_SB();                            \
    __builtin_amdgcn_s_barrier();       \
    G4_SB();                            \
  } while (0)

// ---- epilogue in the accumulator layout (no LDS).  Lane (l31, hi) of block (i, j) holds row i*32 + l31, columns
// j*32 + 8g + 4hi .. +3 for g = 0..3.  permlane32_swap of groups (g, g+1), g even: lanes 0-31 then hold columns 8g .. 8g+7,
// lanes 32-63 columns 8g+8 .. 8g+15 of their row: one 16-byte store each (T21).  Same arithmetic and rounding points as
// gemm_bf16.hip: bf16(gate * act(acc + bias)) is what the PAB slab receives, the residual is added to THAT in fp32.
// Bias and gate vectors are read through the SCALAR cache (wave-uniform address -> s_load, lgkmcnt): a vector load would be a
// vmcnt event, and hipcc's wait in front of its first use drains the whole VMEM queue — the prefetched LDS-DMA pieces of the next
// tile and the stores just issued — in the one interval per tile in which the matrix pipe has nothing else to do.  With them on
// the scalar path the store-only epilogues contain no vmcnt wait at all.
typedef const __attribute__((address_space(4))) uint32_t* sptr_t;

// The finished 16-byte pieces are not stored here: they are handed back in ob[12] (piece (i*3 + j)*2 + k of row orow[i]) and the
// K loop of the NEXT tile stores two of them per K-tile from its LOAD segments.  Every workgroup finishes its tiles at the same
// moment, so an in-place store phase is a chip-wide 25 MB burst that runs at the HBM write rate (~7 k cycles per tile with the
// matrix pipe idle, measured); spread over six K-tiles the same bytes overlap the MFMAs and never queue.
template <int EPI, int ABL>
__device__ __forceinline__ void g4_epilogue(const GemmParams& p, f32x16 (&acc)[2][3], int row0, int col0, int wm, int wn, int l31,
                                            int hi, uint4 (&ob)[12], bf16_t* (&orow)[2]) {
  const int ncol0 = col0 + wn * 96;   // wave-uniform
  int grow_[2];
  bool ok_[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int gr = row0 + wm * 64 + i * 32 + l31;
    ok_[i] = gr < p.M;
    grow_[i] = ok_[i] ? gr : p.M - 1;
  }
  uint4 rres[2][3][2];
  if (EPI == EPI_GATE_RES) {
    if (p.res != nullptr) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int k = 0; k < 2; ++k)
            rres[i][j][k] = *reinterpret_cast<const uint4*>(p.res + (int64_t)grow_[i] * p.ldr + ncol0 + j * 32 + 16 * k + 8 * hi);
    }
  }
  // 96 columns = 48 dwords of bias; lane half hi uses dwords j*16 + 4g + 2hi, +1 (columns j*32 + 8g + 4hi .. +3)
  const bool has_bias = p.bias != nullptr;
  const sptr_t sb = (sptr_t)(p.bias != nullptr ? p.bias + ncol0 : p.out);   // never dereferenced without has_bias
  // gate: one vector per sample (and row segment); the launcher only takes this kernel when a wave's 64 rows always lie in ONE
  // sample / segment (rows_per_sample and seg_split multiples of 64: every denoise-path call), so the vector is wave-uniform
  const bool has_gate = EPI == EPI_GATE_RES && p.gate != nullptr;
  sptr_t sg = (sptr_t)p.out;
  if (has_gate) {
    const int r_first = row0 + wm * 64;
    const int s_first = r_first / p.rows_per_sample;
    const bool seg_first = p.seg_split > 0 && r_first - s_first * p.rows_per_sample < p.seg_split;
    sg = (sptr_t)(p.gate + (int64_t)s_first * p.gate_stride + (seg_first ? p.gate_alt : 0) + ncol0);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      // the 16 dwords (32 columns) of this block, pinned to SGPRs: without the pin hipcc folds the half-wave select into the
      // ADDRESS and emits per-lane global loads again
      uint32_t sbj[16], sgj[16];
      if (has_bias) {
#pragma unroll
        for (int d = 0; d < 16; ++d) sbj[d] = sb[j * 16 + d];
#pragma unroll
        for (int d = 0; d < 16; ++d) asm volatile("" : "+s"(sbj[d]));
      }
      if (has_gate) {
#pragma unroll
        for (int d = 0; d < 16; ++d) sgj[d] = sg[j * 16 + d];
#pragma unroll
        for (int d = 0; d < 16; ++d) asm volatile("" : "+s"(sgj[d]));
      }
      uint2 o[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][4 * g + r];
        if (has_bias) {
          const uint32_t bx = hi ? sbj[4 * g + 2] : sbj[4 * g], by = hi ? sbj[4 * g + 3] : sbj[4 * g + 1];
          v[0] += bflo(bx); v[1] += bfhi(bx); v[2] += bflo(by); v[3] += bfhi(by);
        }
        if (EPI == EPI_BIAS_GELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = gelu_tanh(v[r]);
        }
        if (has_gate) {
          const uint32_t gx = hi ? sgj[4 * g + 2] : sgj[4 * g], gy = hi ? sgj[4 * g + 3] : sgj[4 * g + 1];
          v[0] *= bflo(gx); v[1] *= bfhi(gx); v[2] *= bflo(gy); v[3] *= bfhi(gy);
        }
        o[g].x = pack2bf(v[0], v[1]);
        o[g].y = pack2bf(v[2], v[3]);
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        uint2 lo = o[2 * k], up = o[2 * k + 1];
        auto sx = __builtin_amdgcn_permlane32_swap(lo.x, up.x, false, false);
        auto sy = __builtin_amdgcn_permlane32_swap(lo.y, up.y, false, false);
        uint4 v = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        const int64_t coff = ncol0 + j * 32 + 16 * k + 8 * hi;
        if (EPI == EPI_GATE_RES) {
          if (p.aux != nullptr && ok_[i]) *reinterpret_cast<uint4*>(p.aux + (int64_t)grow_[i] * p.ldaux + coff) = v;
          if (p.res != nullptr) {
            float a[8], b[8];
            unpack8(v, a);
            unpack8(rres[i][j][k], b);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += b[e];
            v = pack8(a);
          }
        }
        ob[(i * 3 + j) * 2 + k] = v;
      }
    }
    orow[i] = ok_[i] ? p.out + (int64_t)grow_[i] * p.ldo + ncol0 + 8 * hi : nullptr;
  }
}

// store piece idx_ (compile-time) of the pending tile: columns j*32 + 16k (+ 8 hi, already in orow) of row block i
template <int ABL>
__device__ __forceinline__ void g4_store_piece(const uint4& v, bf16_t* row, int col) {
  if constexpr (ABL & 8) {   // lab: everything but the HBM stores
    asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
  } else {
    if (row != nullptr) *reinterpret_cast<uint4*>(row + col) = v;
  }
}
#define G4_STORE(idx_) g4_store_piece<ABL>(ob[idx_], orow[(idx_) / 6], (((idx_) % 6) / 2) * 32 + ((idx_) % 2) * 16)
// pieces 2 t_ and 2 t_ + 1 of the pending tile, from LOAD(2 t_) of the running one
#define G4_FLUSH_STEP(t_)                                             \
  do {                                                                \
    if (pend) {                                                       \
      if ((t_) == 0) { G4_STORE(0); G4_STORE(1); }                    \
      else if ((t_) == 1) { G4_STORE(2); G4_STORE(3); }               \
      else if ((t_) == 2) { G4_STORE(4); G4_STORE(5); }               \
      else if ((t_) == 3) { G4_STORE(6); G4_STORE(7); }               \
      else if ((t_) == 4) { G4_STORE(8); G4_STORE(9); }               \
      else if ((t_) == 5) { G4_STORE(10); G4_STORE(11); pend = false; } \
    }                                                                 \
  } while (0)
// whatever is still pending from piece first_ on (tiles with fewer than six K-tiles; the last tile of a workgroup)
#define G4_FLUSH_REST(first_)                                         \
  do {                                                                \
    if (pend) {                                                       \
      if ((first_) <= 0) G4_STORE(0);                                 \
      if ((first_) <= 1) G4_STORE(1);                                 \
      if ((first_) <= 2) G4_STORE(2);                                 \
      if ((first_) <= 3) G4_STORE(3);                                 \
      if ((first_) <= 4) G4_STORE(4);                                 \
      if ((first_) <= 5) G4_STORE(5);                                 \
      if ((first_) <= 6) G4_STORE(6);                                 \
      if ((first_) <= 7) G4_STORE(7);                                 \
      if ((first_) <= 8) G4_STORE(8);                                 \
      if ((first_) <= 9) G4_STORE(9);                                 \
      if ((first_) <= 10) G4_STORE(10);                               \
      if ((first_) <= 11) G4_STORE(11);                               \
      pend = false;                                                   \
    }                                                                 \
  } while (0)

// PERSISTENT form: the grid is min(#tiles, #CUs) workgroups; workgroup b computes tiles b, b + grid, b + 2 grid, ... and the
// K-tile stream (and with it the LDS-DMA ring, its slot counters and the counted waits) simply runs on across tile borders:
// during the last two K-tiles of a tile the pieces issued are the first K-tiles of the NEXT tile, so no tile but the first pays
// a prologue.  Both groups run their epilogue in the SAME interval (G0 right after its last barrier of the tile, in front of
// LOAD(0) of the next tile; G1 right after its last MFMA segment, in front of that barrier), so a tile border costs one epilogue
// of matrix-pipe idle time instead of prologue + two epilogues.  With grid == #tiles this is the one-tile-per-workgroup kernel.
//
// ABL (lab builds only, bit mask): 1 = no LDS-DMA inside the K loop (the ring keeps the prologue's bytes: wrong results, same
// instruction stream otherwise), 2 = no fragment reads inside the K loop, 4 = s_memtime stamps per segment, summed per wave into
// the lab debug buffer as int64[blocks][8 waves][8] = {load, barrier after load, mfma, barrier after mfma, loop, epilogue, start, nt}.
template <int EPI, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm4_kernel(GemmParams p, long long* stamps) {
#if __HIP_DEVICE_COMPILE__
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int grp = wave_u >> 2;           // 0: rows 0..127 (leads), 1: rows 128..255 (one interval behind)
  const int wm = wave_u >> 1, wn = wave_u & 1;
  const int l31 = lane & 31, hi = lane >> 5;

  // tile order: the W-resident raster of gemm_bf16.hip (column groups of 6 tiles inside 8 row-panel groups, one per XCD chunk
  // of the remap).  Workgroup b runs on XCD b % 8 and the grid is a multiple of 8 (or == #tiles), so b + k grid stays on it.
  const int nbn = p.N / BN, nbm = (p.M + BM - 1) / BM, ntiles = nbm * nbn;
  auto decode = [&](int lin, int& row0_, int& col0_) {
    const int tile = xcd_remap(lin, ntiles);
    int bm, bn;
    if (nbn > 6) {
      const int q = nbm / 8, r = nbm - q * 8;
      const int big = r * (q + 1) * nbn;
      int off, np, p0;
      if (tile < big) {
        const int xg = tile / ((q + 1) * nbn);
        off = tile - xg * (q + 1) * nbn; np = q + 1; p0 = xg * (q + 1);
      } else {
        const int t2 = tile - big;
        const int xg = t2 / (q * nbn);
        off = t2 - xg * q * nbn; np = q; p0 = r * (q + 1) + xg * q;
      }
      constexpr int GW = 6;
      const int ng = (nbn + GW - 1) / GW;
      int g = off / (np * GW);
      g = g < ng - 1 ? g : ng - 1;
      const int off2 = off - g * np * GW;
      const int width = g < ng - 1 ? GW : nbn - (ng - 1) * GW;
      const int pm = off2 / width;
      bm = p0 + pm;
      bn = g * GW + (off2 - pm * width);
    } else {
      bm = tile / nbn;
      bn = tile - bm * nbn;
    }
    row0_ = bm * BM;
    col0_ = bn * BN;
  };

  // ---- LDS-DMA assignment (as gemm_bf16.hip): wave w stages A rows [32w, 32w+32) — so G0 stages exactly the rows G0 reads —
  // and W rows [24w, 24w+24); the lane that lands in physical 16-byte slot s of row r fetches logical chunk s ^ ((r>>1)&7).
  int b_off[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int r = wave_u * (NB * 8) + i * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    b_off[i] = (r * (int)p.ldw + c * 8) * 2;
  }
  auto a_offsets = [&](int row0_, int (&o)[NA]) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int r = wave_u * (NA * 8) + i * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      const int rl = row0_ + r < p.M ? r : p.M - 1 - row0_;  // rows past M re-read the last row (never stored)
      o[i] = (rl * (int)p.lda + c * 8) * 2;
    }
  };
  const int a_lds0 = wave_u * (NA * 8) * 128, b_lds0 = W_BASE + wave_u * (NB * 8) * 128;
  // descriptors start at the tile's first row and end with the matrix: reads past the end return 0 instead of faulting
  auto mk_a = [&](int row0_) {
    const int64_t nb = ((int64_t)(p.M - 1 - row0_) * p.lda + p.K) * 2;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)row0_ * p.lda), 0, (int)(nb < 0x7fffffff ? nb : 0x7fffffff), 0x00020000);
  };
  auto mk_b = [&](int col0_) {
    const int64_t nb = ((int64_t)(p.N - 1 - col0_) * p.ldw + p.K) * 2;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)col0_ * p.ldw), 0, (int)(nb < 0x7fffffff ? nb : 0x7fffffff), 0x00020000);
  };
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define G4_DMA_A(rs_, off_, kt_, slot_)                                                                                        \
  do {                                                                                                                         \
    _Pragma("unroll") for (int i_ = 0; i_ < NA; ++i_)                                                                          \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_ptr_t)(smem + (slot_) * A_BYTES + a_lds0 + i_ * 1024), 16, off_[i_], \
                                                 (kt_) * (BK * 2), 0, 0);                                                      \
  } while (0)
#define G4_DMA_W(rs_, kt_, slot_)                                                                                              \
  do {                                                                                                                         \
    _Pragma("unroll") for (int i_ = 0; i_ < NB; ++i_)                                                                          \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_ptr_t)(smem + (slot_) * B_BYTES + b_lds0 + i_ * 1024), 16, b_off[i_], \
                                                 (kt_) * (BK * 2), 0, 0);                                                      \
  } while (0)

  // ---- fragment read offsets: swz(row, chunk = 2 ks + hi) = (row*128 + ((hi ^ ((row>>1)&7)) << 4)) ^ (ks << 5)
  int xo[2], wo[3];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = wm * 64 + i * 32 + l31;
    xo[i] = (r << 7) + ((hi ^ ((r >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int r = wn * 96 + j * 32 + l31;
    wo[j] = W_BASE + (r << 7) + ((hi ^ ((r >> 1) & 7)) << 4);
  }

  f32x16 acc[2][3];
  // fragment registers: W of the whole K-tile (4 k-steps x 3 column blocks), A of one half K-tile (2 k-steps x 2 row blocks)
  bf16x8 w00, w01, w02, w10, w11, w12, w20, w21, w22, w30, w31, w32, a00, a01, a10, a11;
#define G4_READ_W(sw_)                                                          \
  do {                                                                          \
    const char* wb_ = smem + (sw_) * B_BYTES;                                   \
    w00 = *reinterpret_cast<const bf16x8*>(wb_ + wo[0]);                        \
    w01 = *reinterpret_cast<const bf16x8*>(wb_ + wo[1]);                        \
    w02 = *reinterpret_cast<const bf16x8*>(wb_ + wo[2]);                        \
    w10 = *reinterpret_cast<const bf16x8*>(wb_ + (wo[0] ^ 32));                 \
    w11 = *reinterpret_cast<const bf16x8*>(wb_ + (wo[1] ^ 32));                 \
    w12 = *reinterpret_cast<const bf16x8*>(wb_ + (wo[2] ^ 32));                 \
    w20 = *reinterpret_cast<const bf16x8*>(wb_ + (wo[0] ^ 64));                 \
    w21 = *reinterpret_cast<const bf16x8*>(wb_ + (wo[1] ^ 64));                 \
    w22 = *reinterpret_cast<const bf16x8*>(wb_ + (wo[2] ^ 64));                 \
    w30 = *reinterpret_cast<const bf16x8*>(wb_ + (wo[0] ^ 96));                 \
    w31 = *reinterpret_cast<const bf16x8*>(wb_ + (wo[1] ^ 96));                 \
    w32 = *reinterpret_cast<const bf16x8*>(wb_ + (wo[2] ^ 96));                 \
  } while (0)
  // A fragments of k-steps ks0_, ks0_+1 of the tile in slot sa_
#define G4_READ_A(sa_, ks0_)                                                    \
  do {                                                                          \
    const char* ab_ = smem + (sa_) * A_BYTES;                                   \
    a00 = *reinterpret_cast<const bf16x8*>(ab_ + (xo[0] ^ ((ks0_) << 5)));      \
    a01 = *reinterpret_cast<const bf16x8*>(ab_ + (xo[1] ^ ((ks0_) << 5)));      \
    a10 = *reinterpret_cast<const bf16x8*>(ab_ + (xo[0] ^ (((ks0_) + 1) << 5))); \
    a11 = *reinterpret_cast<const bf16x8*>(ab_ + (xo[1] ^ (((ks0_) + 1) << 5))); \
  } while (0)
  // one k-step: 6 MFMAs in the k order of gemm_bf16.hip (each accumulator sees ks = 0, 1, 2, 3 of every K-tile in order)
#define G4_KSTEP(W0_, W1_, W2_, X0_, X1_)                                                       \
  do {                                                                                          \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W0_, X0_, acc[0][0], 0, 0, 0);          \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W1_, X0_, acc[0][1], 0, 0, 0);          \
    acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2_, X0_, acc[0][2], 0, 0, 0);          \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W0_, X1_, acc[1][0], 0, 0, 0);          \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W1_, X1_, acc[1][1], 0, 0, 0);          \
    acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2_, X1_, acc[1][2], 0, 0, 0);          \
  } while (0)
  // MFMA segment of half h_ (0: k-steps 0,1 — 1: k-steps 2,3)
#define G4_MFMA(h_)                                                 \
  do {                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              \
    G4_SB();                                                        \
    __builtin_amdgcn_s_setprio(1);                                  \
    if ((h_) == 0) {                                                \
      G4_KSTEP(w00, w01, w02, a00, a01);                            \
      G4_KSTEP(w10, w11, w12, a10, a11);                            \
    } else {                                                        \
      G4_KSTEP(w20, w21, w22, a00, a01);                            \
      G4_KSTEP(w30, w31, w32, a10, a11);                            \
    }                                                               \
    __builtin_amdgcn_s_setprio(0);                                  \
    G4_SB();                                                        \
  } while (0)

  const int nt = p.K / BK;   // >= 2 (launch check)
  constexpr bool DMA_ON = !(ABL & 1), READ_ON = !(ABL & 2), STAMP = (ABL & 4) != 0;
  unsigned long long st_l = 0, st_b1 = 0, st_m = 0, st_b2 = 0, st_e = 0, st_t0 = 0, st_begin = 0;
#define G4_STAMP(acc_)                                                        \
  do {                                                                        \
    if constexpr (STAMP) {                                                    \
      G4_SB();                                                                \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();           \
      acc_ += now_ - st_t0;                                                   \
      st_t0 = now_;                                                           \
      G4_SB();                                                                \
    }                                                                         \
  } while (0)

  // ---- first tile + prologue: its K-tile 0 complete before the first barrier, K-tile 1 on its way
  int lin = blockIdx.x, row0, col0;
  decode(lin, row0, col0);
  auto ra_c = mk_a(row0), ra_n = ra_c;
  auto rb_c = mk_b(col0), rb_n = rb_c;
  int a_off_c[NA], a_off_n[NA];
  a_offsets(row0, a_off_c);
#pragma unroll
  for (int i = 0; i < NA; ++i) a_off_n[i] = a_off_c[i];
  G4_DMA_A(ra_c, a_off_c, 0, 0);
  G4_DMA_W(rb_c, 0, 0);
  G4_DMA_A(ra_c, a_off_c, 1, 1);
  if (grp == 1) {
    G4_DMA_W(rb_c, 1, 1);
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  }
  G4_BAR();
  if (grp == 1) G4_BAR();   // G1 runs one interval behind G0
  if constexpr (!READ_ON) { G4_READ_W(0); G4_READ_A(0, 0); }
  if constexpr (STAMP) { st_begin = __builtin_amdgcn_s_memtime(); st_t0 = st_begin; }

  // K-tile tt_ of the stream seen from the current tile: tt_ < nt is this tile's, otherwise the next tile's tt_ - nt
#define G4_ISSUE_A(tt_, slot_)                                              \
  do {                                                                      \
    if ((tt_) < nt) G4_DMA_A(ra_c, a_off_c, (tt_), (slot_));                \
    else G4_DMA_A(ra_n, a_off_n, (tt_) - nt, (slot_));                      \
  } while (0)
#define G4_ISSUE_W(tt_, slot_)                                              \
  do {                                                                      \
    if ((tt_) < nt) G4_DMA_W(rb_c, (tt_), (slot_));                         \
    else G4_DMA_W(rb_n, (tt_) - nt, (slot_));                               \
  } while (0)

  int sa = 0, sw = 0;   // ring slots of the current K-tile (carried across tiles)
  uint4 ob[12];         // finished pieces of the previous tile, stored during this tile's K loop
  bf16_t* orow[2] = {nullptr, nullptr};
  bool pend = false;
  for (;;) {
    const int nlin = lin + (int)gridDim.x;
    const bool has_next = nlin < ntiles;
    int nrow0 = row0, ncol0 = col0;
    if (has_next) {
      decode(nlin, nrow0, ncol0);
      ra_n = mk_a(nrow0);
      rb_n = mk_b(ncol0);
      a_offsets(nrow0, a_off_n);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (grp == 0) {
      for (int t = 0; t < nt; ++t) {
        const int sa1 = sa == 2 ? 0 : sa + 1, sa2 = sa1 == 2 ? 0 : sa1 + 1;
        const bool i1 = DMA_ON && (t + 1 < nt || has_next), i2 = DMA_ON && (t + 2 < nt || has_next);
        // LOAD(2t)
        if constexpr (READ_ON) { G4_READ_W(sw); G4_READ_A(sa, 0); }
        G4_SB();
        G4_FLUSH_STEP(t);
        if (i1) G4_ISSUE_W(t + 1, sw ^ 1);
        G4_STAMP(st_l);
        G4_BAR();
        G4_STAMP(st_b1);
        G4_MFMA(0);
        G4_STAMP(st_m);
        G4_BAR();
        G4_STAMP(st_b2);
        // LOAD(2t+1)
        if constexpr (READ_ON) G4_READ_A(sa, 2);
        G4_SB();
        if (i2) G4_ISSUE_A(t + 2, sa2);
        G4_STAMP(st_l);
        G4_BAR();
        G4_STAMP(st_b1);
        G4_MFMA(1);
        if (i2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        G4_STAMP(st_m);
        G4_BAR();
        G4_STAMP(st_b2);
        sa = sa1;
        sw ^= 1;
      }
      G4_FLUSH_REST(2 * nt);   // only when nt < 6
      g4_epilogue<EPI, ABL>(p, acc, row0, col0, wm, wn, l31, hi, ob, orow);   // same interval as LOAD(0) of the next tile
      pend = true;
      G4_STAMP(st_e);
    } else {
      for (int t = 0; t < nt; ++t) {
        const int sa1 = sa == 2 ? 0 : sa + 1, sa2 = sa1 == 2 ? 0 : sa1 + 1;
        const bool i2 = DMA_ON && (t + 2 < nt || has_next);
        // LOAD(2t)
        if constexpr (READ_ON) { G4_READ_W(sw); G4_READ_A(sa, 0); }
        G4_SB();
        G4_FLUSH_STEP(t);
        G4_STAMP(st_l);
        G4_BAR();
        G4_STAMP(st_b1);
        G4_MFMA(0);
        G4_STAMP(st_m);
        G4_BAR();
        G4_STAMP(st_b2);
        // LOAD(2t+1)
        if constexpr (READ_ON) G4_READ_A(sa, 2);
        G4_SB();
        if (i2) {
          G4_ISSUE_A(t + 2, sa2);
          G4_ISSUE_W(t + 2, sw);
          asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        G4_STAMP(st_l);
        G4_BAR();
        G4_STAMP(st_b1);
        G4_MFMA(1);
        G4_STAMP(st_m);
        if (t + 1 < nt) {
          G4_BAR();
          G4_STAMP(st_b2);
        }
        sa = sa1;
        sw ^= 1;
      }
      G4_FLUSH_REST(2 * nt);   // only when nt < 6
      g4_epilogue<EPI, ABL>(p, acc, row0, col0, wm, wn, l31, hi, ob, orow);   // right behind the last MFMA segment, in front of its barrier
      pend = true;
      G4_STAMP(st_e);
      if (has_next) {
        G4_BAR();
        G4_STAMP(st_b2);
      }
    }
    if (!has_next) break;   // (the pieces of the last tile are stored below)
    lin = nlin;
    row0 = nrow0;
    col0 = ncol0;
    ra_c = ra_n;
    rb_c = rb_n;
#pragma unroll
    for (int i = 0; i < NA; ++i) a_off_c[i] = a_off_n[i];
  }
  G4_FLUSH_REST(0);
  if constexpr (STAMP) {
    if (lane == 0 && stamps != nullptr) {
      const unsigned long long end = __builtin_amdgcn_s_memtime();
      long long* d = stamps + ((long long)blockIdx.x * 8 + wave_u) * 8;
      d[0] = (long long)st_l; d[1] = (long long)st_b1; d[2] = (long long)st_m; d[3] = (long long)st_b2;
      d[4] = (long long)(end - st_begin); d[5] = (long long)st_e; d[6] = (long long)st_begin; d[7] = nt;
    }
  }
#endif
}

}  // namespace

static int g4_grid(const GemmParams& p, int persistent) {
  const int ntiles = ((p.M + BM - 1) / BM) * (p.N / BN);
  if (!persistent) return ntiles;
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
    ncu -= ncu % 8;   // a multiple of the XCD count keeps every workgroup's tiles on its XCD chunk of the raster
    if (ncu <= 0) ncu = 8;
  }
  return ntiles < ncu ? ntiles : ncu;
}

#ifdef VSYS_LAB
// lab: ablations / stamps of the ping-pong loop (EPI_BIAS only): abl = 1 no in-loop DMA, 2 no in-loop reads, 3 both, 4 stamps
int launch_gemm4_lab(const GemmParams& p, int abl, int persistent, hipStream_t stream) {
  if (p.N % BN != 0 || p.K % BK != 0 || p.K < 2 * BK) return VSYS_ERR_SHAPE;
  const int grid = g4_grid(p, persistent);
  long long* st = reinterpret_cast<long long*>(get_lab_debug_buffer());
#define G4_LAB(A_)                                                                                                       \
  case A_:                                                                                                               \
    (void)hipFuncSetAttribute((const void*)gemm4_kernel<EPI_BIAS, A_>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); \
    hipLaunchKernelGGL((gemm4_kernel<EPI_BIAS, A_>), dim3(grid), dim3(512), LDS_BYTES, stream, p, st);                   \
    break
  switch (abl) {
    G4_LAB(1);
    G4_LAB(2);
    G4_LAB(3);
    G4_LAB(4);
    G4_LAB(8);
    default: return VSYS_ERR_ARG;
  }
#undef G4_LAB
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

#endif  // VSYS_LAB

// true when launch_gemm4 can take the problem (otherwise the caller stays on gemm_bf16.hip): at least two K-tiles, and a gate vector
// that is uniform over each wave's 64 rows (see g4_epilogue)
bool gemm4_supports(const GemmParams& p, int epi) {
  if (p.N % BN != 0 || p.K % BK != 0 || p.K < 2 * BK) return false;
  if (epi == EPI_GATE_RES && p.gate != nullptr) {
    if (p.rows_per_sample <= 0 || p.rows_per_sample % 64 != 0 || p.seg_split % 64 != 0) return false;
    if ((p.gate_stride & 1) || (p.gate_alt & 1) || (reinterpret_cast<uintptr_t>(p.gate) & 3)) return false;
  }
  if (p.bias != nullptr && (reinterpret_cast<uintptr_t>(p.bias) & 3)) return false;
  return true;
}

int launch_gemm4(const GemmParams& p, int epi, int persistent, hipStream_t stream) {
  if (!gemm4_supports(p, epi)) return VSYS_ERR_SHAPE;
  const int grid = g4_grid(p, persistent);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm4_kernel<EPI_BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm4_kernel<EPI_BIAS_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm4_kernel<EPI_GATE_RES>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  switch (epi) {
    case EPI_BIAS: hipLaunchKernelGGL((gemm4_kernel<EPI_BIAS>), dim3(grid), dim3(512), LDS_BYTES, stream, p, (long long*)nullptr); break;
    case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm4_kernel<EPI_BIAS_GELU>), dim3(grid), dim3(512), LDS_BYTES, stream, p, (long long*)nullptr); break;
    case EPI_GATE_RES: hipLaunchKernelGGL((gemm4_kernel<EPI_GATE_RES>), dim3(grid), dim3(512), LDS_BYTES, stream, p, (long long*)nullptr); break;
    default: return VSYS_ERR_ARG;
  }
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

}  // namespace vsys
