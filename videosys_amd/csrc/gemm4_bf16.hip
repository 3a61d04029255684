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

#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

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
// Row blocks [I0, I1) of the wave tile only: the stream-K body runs the two halves one after the other (half the residual and
// output registers live at a time — it has no deferred stores to hide behind and no registers to spare).
template <int EPI, int ABL, int I0 = 0, int I1 = 2>
__device__ __forceinline__ void g4_epilogue(const GemmParams& p, f32x16 (&acc)[2][3], int row0, int col0, int wm, int wn, int l31,
                                            int hi, uint4 (&ob)[12], bf16_t* (&orow)[2]) {
  const int ncol0 = col0 + wn * 96;   // wave-uniform
  int grow_[2];
  bool ok_[2];
#pragma unroll
  for (int i = I0; i < I1; ++i) {
    const int gr = row0 + wm * 64 + i * 32 + l31;
    ok_[i] = gr < p.M;
    grow_[i] = ok_[i] ? gr : p.M - 1;
  }
  uint4 rres[2][3][2];
  if (EPI == EPI_GATE_RES) {
    if (p.res != nullptr) {
#pragma unroll
      for (int i = I0; i < I1; ++i)
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
  for (int i = I0; i < I1; ++i) {
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
    if (!SK && pend) {                                                \
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
    if (!SK && pend) {                                                \
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

// ---- stream-K tail (SK = true).  A persistent grid of G workgroups leaves ceil(tiles / G) - tiles / G of the last round idle
// (N = 1152 at 38 912 rows: 912 tiles on 256 CUs = 3.56 rounds, 11 % lost).  The launcher therefore runs the tiles of the FULL
// rounds with the persistent body (SkArgs::tile_limit) in workgroups [0, G) and the remaining tiles with the SK = true body in
// workgroups [G, 2 G) of the same grid (gemm4_sk_kernel), in which every workgroup gets a SEGMENT list (sk_plan below): the
// K-tiles of the remaining tiles cut into G equal ranges.  (One body for both parts was built first: with the deferred-store
// registers live across the hand-off code it needs more than 256 VGPRs and hipcc spills inside the K loop; the SK body stores
// its tiles in place instead.)
// A range that does not end with its tile is a DUMP segment: the fp32 accumulators go to the workgroup's slot of a workspace in
// the accumulator layout (the consumer is the same wave / lane of another workgroup, so no layout change) and a per-wave flag is
// released at agent scope.  The range that ends a tile is its FINAL segment: it waits for the flags of the
// workgroups that hold the earlier K ranges of that tile, adds their partial sums in a fixed order and runs the normal epilogue.
// A workgroup runs its DUMP segment FIRST and its FINAL segment LAST, and a FINAL only depends on DUMPs of lower-numbered
// workgroups of its own XCD (b - 8, b - 16, ...): nothing a running workgroup waits for can be undispatched, and the partial sums were
// written long before they are read (no spinning in practice).  Results differ from the unsplit kernels by fp32 summation order only (deterministic).
struct SkArgs {
  const int4* segs;   // [grid][nseg_max]: x = linear tile id (-1 ends the list), y = kb | ke << 16 (K-tile range), z = kind | nsrc << 8
  float* ws;          // [grid][8 waves][24][64 lanes] float4
  int* flags;         // [grid][8 waves]: epoch of the launch whose partial sums the slot holds
  int nseg_max, epoch;
  int tile_limit;     // SK = false: tiles [0, tile_limit) only (0 = all); the rest belongs to the SK role
  int abl;            // lab builds: 1 = no DUMP stores (flag only), 2 = no FINAL poll / gather, 4 = SK role does nothing (wrong results)
};
constexpr int SK_FULL = 0, SK_DUMP = 1, SK_FINAL = 2;
#ifdef VSYS_LAB
#define SK_LAB_ON(bit_) (!(sk.abl & (bit_)))
#else
#define SK_LAB_ON(bit_) true
#endif

// Hand-off (cdna_hip_programming.md §6 Guideline 16, recipe R1): the partial sums leave as 16-byte WRITE-THROUGH (sc1) buffer
// stores, every storing wave drains its own stores and publishes its own flag (a wave's slot is read by the same wave index of
// the consumer, so no workgroup barrier is needed — the barrier sequence of the K loop stays untouched); the consumer polls that
// one word relaxed and reads the slot with sc1 loads (L1-bypassing; valid because the producer stored sc1).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// row i of this workgroup's segment list through the scalar cache (the list is constant for the launch; a plain load of memory the
// kernel cannot prove read-only becomes a VECTOR load and drags tile coordinates and loop bounds into VGPRs)
__device__ __forceinline__ int4 sk_seg(const SkArgs& sk, int bid, int i) {
  const sptr_t q = (sptr_t)(sk.segs + bid * sk.nseg_max + i);
  uint32_t x = q[0], y = q[1], z = q[2];
  asm volatile("" : "+s"(x), "+s"(y), "+s"(z));
  return make_int4((int)x, (int)y, (int)z, 0);
}
constexpr int SK_SC1 = 16;
constexpr int SK_WAVE_BYTES = 24 * 64 * 16;   // one wave's accumulators: 24 x 16 bytes per lane

__device__ __forceinline__ void g4_sk_dump(const SkArgs& sk, f32x16 (&acc)[2][3], int bid, int wave_u, int lane) {
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<char*>(sk.ws) + (int64_t)(bid * 8 + wave_u) * SK_WAVE_BYTES),
                                                    0, SK_WAVE_BYTES, 0x00020000);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x4 v;
        v.x = __float_as_uint(acc[i][j][4 * g]); v.y = __float_as_uint(acc[i][j][4 * g + 1]);
        v.z = __float_as_uint(acc[i][j][4 * g + 2]); v.w = __float_as_uint(acc[i][j][4 * g + 3]);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16, ((i * 3 + j) * 4 + g) * 1024, SK_SC1);
      }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_store(sk.flags + bid * 8 + wave_u, sk.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void g4_sk_gather(const SkArgs& sk, f32x16 (&acc)[2][3], int bid, int wave_u, int lane, int nsrc) {
  for (int s = 1; s <= nsrc; ++s) {
    const int src = bid - 8 * s;
    const int* f = sk.flags + src * 8 + wave_u;
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sk.epoch) __builtin_amdgcn_s_sleep(8);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<char*>(sk.ws) + (int64_t)(src * 8 + wave_u) * SK_WAVE_BYTES), 0,
                                                      SK_WAVE_BYTES, 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        u32x4 v[4];   // one accumulator block (16 registers) in flight at a time: more makes hipcc spill inside the K loop
#pragma unroll
        for (int g = 0; g < 4; ++g) v[g] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, ((i * 3 + j) * 4 + g) * 1024, SK_SC1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          acc[i][j][4 * g] += __uint_as_float(v[g].x); acc[i][j][4 * g + 1] += __uint_as_float(v[g].y);
          acc[i][j][4 * g + 2] += __uint_as_float(v[g].z); acc[i][j][4 * g + 3] += __uint_as_float(v[g].w);
        }
        G4_SB();
      }
  }
}

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
// bid / nblk: this workgroup's index and the workgroup count of ITS role (gemm4_sk_kernel runs two roles in one grid)
template <int EPI, int ABL, bool SK>
__device__ __forceinline__ void gemm4_body(const GemmParams& p, long long* stamps, const SkArgs& sk, char* smem, const int bid, const int nblk) {
#if __HIP_DEVICE_COMPILE__
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
  // The slot base (a multiple of 8 KiB) is added BEFORE the k-step XOR (bits 5, 6: same address either way): the XOR then depends
  // on the ring slot and cannot be hoisted out of the K loop into nine more loop-invariant address registers.
#define G4_READ_W(sw_)                                                          \
  do {                                                                          \
    const int w0_ = wo[0] + (sw_) * B_BYTES, w1_ = wo[1] + (sw_) * B_BYTES, w2_ = wo[2] + (sw_) * B_BYTES; \
    w00 = *reinterpret_cast<const bf16x8*>(smem + w0_);                         \
    w01 = *reinterpret_cast<const bf16x8*>(smem + w1_);                         \
    w02 = *reinterpret_cast<const bf16x8*>(smem + w2_);                         \
    w10 = *reinterpret_cast<const bf16x8*>(smem + (w0_ ^ 32));                  \
    w11 = *reinterpret_cast<const bf16x8*>(smem + (w1_ ^ 32));                  \
    w12 = *reinterpret_cast<const bf16x8*>(smem + (w2_ ^ 32));                  \
    w20 = *reinterpret_cast<const bf16x8*>(smem + (w0_ ^ 64));                  \
    w21 = *reinterpret_cast<const bf16x8*>(smem + (w1_ ^ 64));                  \
    w22 = *reinterpret_cast<const bf16x8*>(smem + (w2_ ^ 64));                  \
    w30 = *reinterpret_cast<const bf16x8*>(smem + (w0_ ^ 96));                  \
    w31 = *reinterpret_cast<const bf16x8*>(smem + (w1_ ^ 96));                  \
    w32 = *reinterpret_cast<const bf16x8*>(smem + (w2_ ^ 96));                  \
  } while (0)
  // A fragments of k-steps ks0_, ks0_+1 of the tile in slot sa_
#define G4_READ_A(sa_, ks0_)                                                    \
  do {                                                                          \
    const int x0_ = xo[0] + (sa_) * A_BYTES, x1_ = xo[1] + (sa_) * A_BYTES;     \
    a00 = *reinterpret_cast<const bf16x8*>(smem + (x0_ ^ ((ks0_) << 5)));       \
    a01 = *reinterpret_cast<const bf16x8*>(smem + (x1_ ^ ((ks0_) << 5)));       \
    a10 = *reinterpret_cast<const bf16x8*>(smem + (x0_ ^ (((ks0_) + 1) << 5))); \
    a11 = *reinterpret_cast<const bf16x8*>(smem + (x1_ ^ (((ks0_) + 1) << 5))); \
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

  int nt = p.K / BK;   // K-tiles of the current segment: >= 2 (launch check / sk_plan)
  int kb_c = 0, kb_n = 0, kind_c = SK_FULL, nsrc_c = 0, seg_i = 0;   // SK: first K-tile of the current / next segment, ...
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
  int lin = bid, row0, col0;
  if constexpr (SK) {
    const int4 s0 = sk_seg(sk, bid, 0);
    if (s0.x < 0 || !SK_LAB_ON(4)) return;   // no segment for this workgroup (uniform: before the first barrier)
    lin = s0.x; kb_c = s0.y & 0xffff; nt = (s0.y >> 16) - kb_c; kind_c = s0.z & 0xff; nsrc_c = s0.z >> 8;
  }
  decode(lin, row0, col0);
  auto ra_c = mk_a(row0), ra_n = ra_c;
  auto rb_c = mk_b(col0), rb_n = rb_c;
  int a_off_c[NA], a_off_n[NA];
  a_offsets(row0, a_off_c);
#pragma unroll
  for (int i = 0; i < NA; ++i) a_off_n[i] = a_off_c[i];
  G4_DMA_A(ra_c, a_off_c, kb_c, 0);
  G4_DMA_W(rb_c, kb_c, 0);
  G4_DMA_A(ra_c, a_off_c, kb_c + 1, 1);
  if (grp == 1) {
    G4_DMA_W(rb_c, kb_c + 1, 1);
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
    if ((tt_) < nt) G4_DMA_A(ra_c, a_off_c, kb_c + (tt_), (slot_));         \
    else G4_DMA_A(ra_n, a_off_n, kb_n + (tt_) - nt, (slot_));               \
  } while (0)
#define G4_ISSUE_W(tt_, slot_)                                              \
  do {                                                                      \
    if ((tt_) < nt) G4_DMA_W(rb_c, kb_c + (tt_), (slot_));                  \
    else G4_DMA_W(rb_n, kb_n + (tt_) - nt, (slot_));                        \
  } while (0)

  int sa = 0, sw = 0;   // ring slots of the current K-tile (carried across tiles)
  uint4 ob[12];         // finished pieces of the previous tile, stored during this tile's K loop
  bf16_t* orow[2] = {nullptr, nullptr};
  bool pend = false;
  for (;;) {
    int nlin = lin + nblk, nt_n = nt, kind_n = SK_FULL, nsrc_n = 0;
    bool has_next = nlin < (sk.tile_limit > 0 ? sk.tile_limit : ntiles);
    if constexpr (SK) {
      const int4 sn = sk_seg(sk, bid, seg_i + 1);   // every list ends with a lin = -1 row
      nlin = sn.x; has_next = nlin >= 0;
      kb_n = sn.y & 0xffff; nt_n = (sn.y >> 16) - kb_n; kind_n = sn.z & 0xff; nsrc_n = sn.z >> 8;
    }
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
      if (SK && kind_c == SK_DUMP) {
        if (SK_LAB_ON(1)) g4_sk_dump(sk, acc, bid, wave_u, lane);
        else if (lane == 0) __hip_atomic_store(sk.flags + bid * 8 + wave_u, sk.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // lab: flag only
      } else {
        if (SK && kind_c == SK_FINAL && SK_LAB_ON(2)) g4_sk_gather(sk, acc, bid, wave_u, lane, nsrc_c);
        if constexpr (SK) {   // no deferred stores in the SK body (registers): two halves, stored in place
          uint4 ob[12];   // (shadows the deferred-store registers of the persistent body: nothing of this tile outlives the block)
          bf16_t* orow[2] = {nullptr, nullptr};
          g4_epilogue<EPI, ABL, 0, 1>(p, acc, row0, col0, wm, wn, l31, hi, ob, orow);
          G4_STORE(0); G4_STORE(1); G4_STORE(2); G4_STORE(3); G4_STORE(4); G4_STORE(5);
          G4_SB();
          g4_epilogue<EPI, ABL, 1, 2>(p, acc, row0, col0, wm, wn, l31, hi, ob, orow);
          G4_STORE(6); G4_STORE(7); G4_STORE(8); G4_STORE(9); G4_STORE(10); G4_STORE(11);
        } else {
          g4_epilogue<EPI, ABL>(p, acc, row0, col0, wm, wn, l31, hi, ob, orow);   // same interval as LOAD(0) of the next tile
          pend = true;
        }
      }
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
      if (SK && kind_c == SK_DUMP) {
        if (SK_LAB_ON(1)) g4_sk_dump(sk, acc, bid, wave_u, lane);
        else if (lane == 0) __hip_atomic_store(sk.flags + bid * 8 + wave_u, sk.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // lab: flag only
      } else {
        if (SK && kind_c == SK_FINAL && SK_LAB_ON(2)) g4_sk_gather(sk, acc, bid, wave_u, lane, nsrc_c);
        if constexpr (SK) {
          uint4 ob[12];   // (shadows the deferred-store registers of the persistent body: nothing of this tile outlives the block)
          bf16_t* orow[2] = {nullptr, nullptr};
          g4_epilogue<EPI, ABL, 0, 1>(p, acc, row0, col0, wm, wn, l31, hi, ob, orow);
          G4_STORE(0); G4_STORE(1); G4_STORE(2); G4_STORE(3); G4_STORE(4); G4_STORE(5);
          G4_SB();
          g4_epilogue<EPI, ABL, 1, 2>(p, acc, row0, col0, wm, wn, l31, hi, ob, orow);
          G4_STORE(6); G4_STORE(7); G4_STORE(8); G4_STORE(9); G4_STORE(10); G4_STORE(11);
        } else {
          g4_epilogue<EPI, ABL>(p, acc, row0, col0, wm, wn, l31, hi, ob, orow);   // right behind the last MFMA segment, in front of its barrier
          pend = true;
        }
      }
      G4_STAMP(st_e);
      if (has_next) {
        G4_BAR();
        G4_STAMP(st_b2);
      }
    }
    if (!has_next) break;   // (the pieces of the last tile are stored below)
    lin = nlin;
    if constexpr (SK) { kb_c = kb_n; nt = nt_n; kind_c = kind_n; nsrc_c = nsrc_n; ++seg_i; }
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
      long long* d = stamps + ((long long)bid * 8 + wave_u) * 8;
      d[0] = (long long)st_l; d[1] = (long long)st_b1; d[2] = (long long)st_m; d[3] = (long long)st_b2;
      d[4] = (long long)(end - st_begin); d[5] = (long long)st_e; d[6] = (long long)st_begin; d[7] = nt;
    }
  }
#endif
}

template <int EPI, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm4_kernel(GemmParams p, long long* stamps, SkArgs sk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm4_body<EPI, ABL, false>(p, stamps, sk, smem, (int)blockIdx.x, (int)gridDim.x);
}

// Two roles in ONE grid of 2 G workgroups (no kernel boundary, no second launch): workgroups [0, G) walk the tiles of the full
// rounds (tile_limit), workgroups [G, 2 G) are dispatched as the first ones retire and run the segment lists of the partial round.
// Workgroup G + b lands on XCD b % 8 like workgroup b (G is a multiple of 8).
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm4_sk_kernel(GemmParams p, SkArgs sk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int G = (int)gridDim.x >> 1;
  if ((int)blockIdx.x < G) gemm4_body<EPI, 0, false>(p, nullptr, sk, smem, (int)blockIdx.x, G);
  else gemm4_body<EPI, 0, true>(p, nullptr, sk, smem, (int)blockIdx.x - G, G);
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

// ---- stream-K plan (host).  Segment lists per workgroup for the tiles [ntiles - ntiles % grid, ntiles) of (ntiles, nt, grid);
// see the kernel comment.  Returns false when the shape gains nothing (no partial round) or cannot be cut into pieces of >= 2
// K-tiles with at most one DUMP per workgroup.
bool sk_plan(int ntiles, int nt, int grid, std::vector<int4>& segs, int& nseg_max) {
  const int R = ntiles / grid, rem = ntiles % grid;
  if (R < 1 || rem == 0 || nt < 4 || grid % 8 != 0 || nt >= 0x7fff) return false;
  const int G = grid / 8;
  std::vector<std::vector<int4>> dumps(grid), fulls(grid), finals(grid);   // the tiles of the partial round only
  for (int x = 0; x < 8; ++x) {
    int cx = 0;
    while (R * grid + x + 8 * cx < ntiles) ++cx;   // tiles of the partial round whose linear id is on XCD x
    if (cx == 0) continue;
    const long long units = (long long)cx * nt;
    std::vector<int> earlier(cx, 0);
    auto cut = [&](int j) {
      long long c = (long long)j * units / G;
      const int m = (int)(c % nt);
      if (m == 1) c -= 1; else if (m == nt - 1) c += 1;   // no 1-K-tile pieces at a tile border
      return c;
    };
    for (int j = 0; j < G; ++j) {
      const int b = x + 8 * j;
      long long u = cut(j);
      const long long u1 = cut(j + 1);
      while (u < u1) {
        const int i = (int)(u / nt), kb = (int)(u % nt);
        const int ke = (int)std::min<long long>(nt, kb + (u1 - u));
        if (ke - kb < 2) return false;
        const int lin = R * grid + x + 8 * i;
        if (kb == 0 && ke == nt) fulls[b].push_back(make_int4(lin, nt << 16, SK_FULL, 0));
        else if (ke < nt) {
          dumps[b].push_back(make_int4(lin, kb | (ke << 16), SK_DUMP, 0));
          ++earlier[i];
        } else {
          if (earlier[i] < 1 || 8 * earlier[i] > b) return false;
          finals[b].push_back(make_int4(lin, kb | (ke << 16), SK_FINAL | (earlier[i] << 8), 0));
        }
        u += ke - kb;
      }
      if (dumps[b].size() > 1 || finals[b].size() > 1) return false;   // one workspace slot per workgroup
    }
  }
  // a FINAL of workgroup b reads the slots of b - 8, ..., b - 8 nsrc: they must be exactly the DUMPs of the same tile
  for (int b = 0; b < grid; ++b)
    for (const int4& f : finals[b])
      for (int k = 1; k <= (f.z >> 8); ++k) {
        const int src = b - 8 * k;
        if (src < 0 || dumps[src].size() != 1 || dumps[src][0].x != f.x) return false;
      }
  size_t mx = 0;
  for (int b = 0; b < grid; ++b) mx = std::max(mx, dumps[b].size() + fulls[b].size() + finals[b].size());
  nseg_max = (int)mx + 1;
  segs.assign((size_t)grid * nseg_max, make_int4(-1, 0, 0, 0));
  for (int b = 0; b < grid; ++b) {
    size_t o = (size_t)b * nseg_max;
    for (const int4& v : dumps[b]) segs[o++] = v;
    for (const int4& v : fulls[b]) segs[o++] = v;
    for (const int4& v : finals[b]) segs[o++] = v;
  }
  return true;
}

namespace {
struct SkPlanDev { int4* d_segs = nullptr; int nseg_max = 0; bool ok = false; };
struct SkWorkspace { float* ws = nullptr; int* flags = nullptr; int epoch = 0; int grid = 0; };
std::mutex g_sk_mutex;
std::map<std::tuple<int, int, int>, SkPlanDev> g_sk_plans;      // (ntiles, nt, grid)
std::map<hipStream_t, SkWorkspace> g_sk_ws;                      // launches on one stream are serialised: one workspace per stream
}  // namespace

// fills sk for a launch on `stream`; false: run the plain persistent kernel.  Plans and workspaces live on the device that is current
// at first use: this library's process model is one process per GPU (engine.py / torchrun), as is the reference's.
static bool sk_prepare(const GemmParams& p, int grid, hipStream_t stream, SkArgs& sk) {
  const int ntiles = ((p.M + BM - 1) / BM) * (p.N / BN), nt = p.K / BK;
  std::lock_guard<std::mutex> lock(g_sk_mutex);
  SkPlanDev& pl = g_sk_plans[std::make_tuple(ntiles, nt, grid)];
  if (pl.d_segs == nullptr && !pl.ok && pl.nseg_max == 0) {
    pl.nseg_max = -1;   // built (possibly without success)
    std::vector<int4> segs;
    int nseg_max = 0;
    if (sk_plan(ntiles, nt, grid, segs, nseg_max)) {
      if (hipMalloc(&pl.d_segs, segs.size() * sizeof(int4)) == hipSuccess &&
          hipMemcpy(pl.d_segs, segs.data(), segs.size() * sizeof(int4), hipMemcpyHostToDevice) == hipSuccess) {
        pl.nseg_max = nseg_max;
        pl.ok = true;
      }
    }
  }
  if (!pl.ok) return false;
  SkWorkspace& w = g_sk_ws[stream];
  if (w.ws == nullptr || w.grid < grid) {
    if (w.ws != nullptr) { (void)hipFree(w.ws); (void)hipFree(w.flags); w.ws = nullptr; }
    const size_t wb = (size_t)grid * 8 * 24 * 64 * sizeof(float4), fb = (size_t)grid * 8 * sizeof(int);
    if (hipMalloc(&w.ws, wb) != hipSuccess) { w.ws = nullptr; return false; }
    if (hipMalloc(&w.flags, fb) != hipSuccess || hipMemset(w.flags, 0, fb) != hipSuccess) { (void)hipFree(w.ws); w.ws = nullptr; return false; }
    w.grid = grid;
    w.epoch = 0;
  }
  sk.segs = pl.d_segs;
  sk.nseg_max = pl.nseg_max;
  sk.ws = w.ws;
  sk.flags = w.flags;
  sk.epoch = ++w.epoch;
  return true;
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
    hipLaunchKernelGGL((gemm4_kernel<EPI_BIAS, A_>), dim3(grid), dim3(512), LDS_BYTES, stream, p, st, SkArgs{});         \
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

// persistent: 0 = one tile per workgroup, 1 = one workgroup per CU walks the tiles, 2 = 1 + stream-K split of the partial last round
int launch_gemm4(const GemmParams& p, int epi, int persistent, hipStream_t stream) {
  if (!gemm4_supports(p, epi)) return VSYS_ERR_SHAPE;
  int sk_abl = 0;
  if (persistent > 2) { sk_abl = persistent - 2; persistent = 2; }   // lab: 3 .. 9 = stream-K with ablation bits persistent - 2
  const int grid = g4_grid(p, persistent);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm4_kernel<EPI_BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm4_kernel<EPI_BIAS_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm4_kernel<EPI_GATE_RES>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm4_sk_kernel<EPI_BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm4_sk_kernel<EPI_BIAS_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm4_sk_kernel<EPI_GATE_RES>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  SkArgs sk{};
  if (persistent == 2 && sk_prepare(p, grid, stream, sk)) {
    // one grid, two roles: workgroups [0, grid) = full rounds on tiles [0, R grid), workgroups [grid, 2 grid) = the segment lists
    const int ntiles = ((p.M + BM - 1) / BM) * (p.N / BN);
    sk.tile_limit = ntiles - ntiles % grid;
    sk.abl = sk_abl;
    switch (epi) {
      case EPI_BIAS: hipLaunchKernelGGL((gemm4_sk_kernel<EPI_BIAS>), dim3(2 * grid), dim3(512), LDS_BYTES, stream, p, sk); break;
      case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm4_sk_kernel<EPI_BIAS_GELU>), dim3(2 * grid), dim3(512), LDS_BYTES, stream, p, sk); break;
      case EPI_GATE_RES: hipLaunchKernelGGL((gemm4_sk_kernel<EPI_GATE_RES>), dim3(2 * grid), dim3(512), LDS_BYTES, stream, p, sk); break;
      default: return VSYS_ERR_ARG;
    }
    return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
  }
  switch (epi) {
    case EPI_BIAS: hipLaunchKernelGGL((gemm4_kernel<EPI_BIAS>), dim3(grid), dim3(512), LDS_BYTES, stream, p, (long long*)nullptr, sk); break;
    case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm4_kernel<EPI_BIAS_GELU>), dim3(grid), dim3(512), LDS_BYTES, stream, p, (long long*)nullptr, sk); break;
    case EPI_GATE_RES: hipLaunchKernelGGL((gemm4_kernel<EPI_GATE_RES>), dim3(grid), dim3(512), LDS_BYTES, stream, p, (long long*)nullptr, sk); break;
    default: return VSYS_ERR_ARG;
  }
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

}  // namespace vsys
