// bf16 MFMA GEMM, "two workgroups per CU" geometry (variant 20 of vsys_tune_gemm_variant).
//
// Same contract, tile (256 x 192) and epilogues as gemm_bf16.hip; what changes is who shares a SIMD.  In the 8-wave kernel
// the two waves of a SIMD belong to ONE workgroup: they reach the tile barrier, the prologue and the epilogue together, and
// the matrix pipe idles through all three (ablations at config 2: the epilogue + relaunch + first-tile latency are 16-30 %
// of the K = 1152 GEMMs).  Here a workgroup is 4 waves (2 x 2), every wave owns a 128 x 96 output tile = 4 x 3
// v_mfma_f32_32x32x16_bf16 accumulators (192 VGPRs), the K step is 32 so the staging footprint is 72 KiB, and TWO
// independent workgroups are resident per CU (144 KiB of LDS, 2 waves per SIMD, one from each).  While one workgroup
// converts, stores, relaunches or waits for its first operand tile, the other one has the whole matrix pipe.
//
// LDS per workgroup: three A slots (256 rows x 64 B) + two W slots (192 rows x 64 B).  64-byte rows: four rows per
// 256-byte bank row; 16-byte chunk c of row r is stored at chunk c ^ ((r>>2)&3), so the 16 rows a ds_read_b128 lane group
// touches hit 16 distinct slots.  The swizzle sits on the source side of the LDS-DMA (buffer_load_dwordx4 ... lds).
// Pipeline: A(t+2) and W(t+1) are issued during stage t, W first; the wait in front of the stage barrier is vmcnt(4).
#include "common.h"
#include "vsys_internal.h"

namespace vsys {
namespace {

constexpr int BM = 256, BK = 32;
constexpr int A_SLOT = BM * BK * 2;   // 16384
constexpr int W_BASE = 3 * A_SLOT;    // 49152
constexpr int OUT_ROW_BYTES = 96 * 2 + 16;
constexpr int OUT_WAVE_BYTES = 64 * OUT_ROW_BYTES;  // 13312 per wave and pass

// NWN = waves along N.  2: 4 waves, tile 256 x 192, 72 KiB, TWO workgroups per CU (variant 20).  4: 8 waves, tile 256 x 384,
// one workgroup per CU (variant 30): the loop needs (256 + 384) / (256 * 384) = 0.0065 operand bytes per flop instead of 0.0091,
// and 384 still divides every N of the denoise path (1152 = 3, 3456 = 9, 4608 = 12 tiles).
template <int NWN>
struct G2 {
  static constexpr int NW = 2 * NWN;
  static constexpr int NT = 64 * NW;
  static constexpr int BN = 96 * NWN;
  static constexpr int NA = 16 / NW;             // A pieces (16 rows x 64 B) per wave and stage: 4 or 2
  static constexpr int W_SLOT = BN * BK * 2;     // 12288 or 24576
  // W prefetch distance in stages.  An LDS-DMA piece lands ~1.1 us after issue (MI355X_MICROARCH.md, ldsdma-fill) but a stage
  // is only ~0.65 us of MFMAs, so a one-stage lead leaves every stage barrier waiting for W; the wide tile has the LDS for a
  // third W slot (3 x 16 + 3 x 24 = 120 KiB) and prefetches BOTH operands two stages ahead.
  static constexpr int WLEAD = NWN == 4 ? 2 : 1;
  static constexpr int NWSLOT = WLEAD + 1;
  static constexpr int STAGING = W_BASE + NWSLOT * W_SLOT;
  static constexpr int LDS_BYTES = STAGING > NW * OUT_WAVE_BYTES ? STAGING : NW * OUT_WAVE_BYTES;  // 73728 or 106496
  // EPI_LN_*: (mu, rstd) of the tile's 256 rows + (cs, cv) of its columns behind the staging area (common.h ln_stage_tile)
  static constexpr int LN_BYTES = (BM + BN) * 8;
};

// STAMP = 1 (lab, variant 31): s_memtime stamps around the four phases of every stage (fragment reads landed / MFMA block issued /
// counted waits / barrier), summed per wave into p.aux viewed as int64[blocks][waves][8] — the cycle accounting of one stage.
// MF = 1 (round 6): the same tile walk on v_mfma_f32_16x16x32_bf16.  Under the 1400 W cap the matrix pipe itself is ~12 % cheaper per
// flop in that shape (bare loop, random operands: 2.05 vs 1.83 PFLOP/s; with the fragment reads of this wave tile 1.90 vs 1.67, with
// the LDS-DMA stream on top 1.26 vs 1.18: tools/micro/mfma_peak.hip, energy_probe.hip modes 10-14) — half the accumulator bytes per
// flop.  One stage (BK = 32) is ONE k-step: 8 token blocks x 6 column blocks of 16 x 16, taken in two halves of four token blocks so
// that the fragment registers stay at ten (the cadence of the 32x32x16 loop: six MFMAs = 96 pipe cycles, then a reload + DMA pieces).
// A lane owns token (16 i + lane % 16) and the four columns 16 j + 4 (lane / 16) ..: the epilogues below index by that.
// LDS image: chunk c of row r at c ^ F((r >> 2) & 3) with F = (0, 2, 3, 1) — the 16-lane groups of ds_read_b128 ({0-3, 12-15, 20-27}, ...)
// then meet 16 distinct slots with rows l % 16 and chunk l / 16 (the (r >> 2) & 3 swizzle of the 32-row fragments would 2-way conflict).
template <int EPI, int NWN, int STAMP = 0, int MF = 0>
__global__ __launch_bounds__(G2<NWN>::NT, 2) void gemm2_kernel(GemmParams p) {
  using G = G2<NWN>;
  constexpr int BN = G::BN, NA = G::NA, W_SLOT = G::W_SLOT, WLEAD = G::WLEAD;
#if __HIP_DEVICE_COMPILE__  // buffer-resource types exist in the device pass only
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
  const int l31 = lane & 31, hi = lane >> 5;
  const int l15 = lane & 15, lq = lane >> 4;   // MF = 1: fragment row / 16-byte k-chunk, accumulator token / column quad
  static_assert(!MF || (EPI != EPI_F32_SLICES && !STAMP), "the 16x16x32 form covers the token-row epilogues only");
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // tile order: same W-resident raster as gemm_bf16.hip (column groups of 6 inside 8 row-panel groups)
  const int nbn = p.N / BN;
  int tile = xcd_remap(blockIdx.x, gridDim.x);
  // EPI_F32_SLICES: the 1-D grid is (K slice, tile) in slice-major order, so the contiguous run of it that xcd_remap gives an
  // XCD holds one or two K slices: the activation columns of a slice (384 x ks bf16, < 1 MB) stay in that XCD's L2 while its
  // weight panels stream through — every workgroup re-reads them, and with the slices spread over all XCDs each L2 saw ALL of
  // the activations (3-8 MB against 4 MB of L2) next to the weight stream
  int kslice = 0;
  if constexpr (EPI == EPI_F32_SLICES) {
    const int ntile = ((p.M + BM - 1) / BM) * nbn;
    kslice = tile / ntile;
    tile -= kslice * ntile;
  }
  int bm, bn;
  gemm_raster(tile, (p.M + BM - 1) / BM, nbn, p.raster_gw, p.raster_ph, bm, bn);
  const int row0 = bm * BM, col0 = bn * BN;

  // ---- LDS-DMA assignment: a piece = 1 KiB = 16 rows x 64 B; lane l -> row (l>>2), physical chunk l&3, which holds
  // logical chunk (l&3) ^ ((row>>2)&3) = (l&3) ^ ((l>>4)&3) (piece bases are multiples of 16 rows).
  // wave w stages A rows [64w, 64w+64) (4 pieces) and W rows [48w, 48w+48) (3 pieces) of every stage.
  const int dchunk = ((lane & 3) ^ (MF ? ((0x78 >> (2 * ((lane >> 4) & 3))) & 3) : ((lane >> 4) & 3))) * 16;
  int a_off[NA], b_off[3];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int r = wave_u * (16 * NA) + i * 16 + (lane >> 2);
    const int rl = row0 + r < p.M ? r : p.M - 1 - row0;  // rows past M re-read the last row (never stored)
    a_off[i] = rl * (int)p.lda * 2 + dchunk;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) b_off[i] = (wave_u * 48 + i * 16 + (lane >> 2)) * (int)p.ldw * 2 + dchunk;
  const int64_t a_bytes = ((int64_t)(p.M - 1 - row0) * p.lda + p.K) * 2, b_bytes = ((int64_t)(p.N - 1 - col0) * p.ldw + p.K) * 2;
  const auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)row0 * p.lda), 0,
                                                        (int)(a_bytes < 0x7fffffff ? a_bytes : 0x7fffffff), 0x00020000);
  const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)col0 * p.ldw), 0,
                                                        (int)(b_bytes < 0x7fffffff ? b_bytes : 0x7fffffff), 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  // EPI_F32_SLICES: blockIdx.y picks a K slice of p.ks columns; every other epilogue walks the whole K
  const int kbase2 = EPI == EPI_F32_SLICES ? kslice * p.ks * 2 : 0;
  auto dma_a = [&](int i, int t, int slot) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(smem + slot * A_SLOT + (wave_u * (16 * NA) + i * 16) * 64), 16, a_off[i],
                                             kbase2 + t * (BK * 2), 0, 0);
  };
  auto dma_w = [&](int i, int t, int slot) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(smem + W_BASE + slot * W_SLOT + (wave_u * 48 + i * 16) * 64), 16,
                                             b_off[i], kbase2 + t * (BK * 2), 0, 0);
  };

  // ---- fragment read offsets: row r, k-step ks: logical chunk 2ks + hi at physical chunk ^ ((r>>2)&3); all fragment rows of
  // a lane are l31 + a multiple of 32, so (r>>2)&3 = (l31>>2)&3 and the k-step is an XOR of bit 5
  const int fsw = MF ? ((lq ^ ((0x78 >> (2 * ((l15 >> 2) & 3))) & 3)) << 4) : ((hi ^ ((l31 >> 2) & 3)) << 4);
  const int frow = MF ? l15 : l31;
  const int xo = (wm * 128 + frow) * 64 + fsw;           // + i*2048 for m-block i (MF: + i*1024 for token block i)
  const int wo = W_BASE + (wn * 96 + frow) * 64 + fsw;   // + j*2048 for n-block j (MF: + j*1024 for column block j)

  f32x16 acc[MF ? 1 : 4][MF ? 1 : 3];
  f32x4 acc16[MF ? 8 : 1][MF ? 6 : 1];   // MF: token block i (16 tokens), column block j (16 columns)
#pragma unroll
  for (int i = 0; i < (MF ? 1 : 4); ++i)
#pragma unroll
    for (int j = 0; j < (MF ? 1 : 3); ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
  for (int i = 0; i < (MF ? 8 : 1); ++i)
#pragma unroll
    for (int j = 0; j < (MF ? 6 : 1); ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.f;

  constexpr bool LN = EPI == EPI_LN_BIAS || EPI == EPI_LN_GELU;
  float2* ln_lds = reinterpret_cast<float2*>(smem + G::LDS_BYTES);
  // (slices: the last one may be shorter — K need not divide evenly)
  const int nt = (EPI == EPI_F32_SLICES ? (p.ks < p.K - kslice * p.ks ? p.ks : p.K - kslice * p.ks) : p.K) / BK;
#pragma unroll
  for (int i = 0; i < NA; ++i) dma_a(i, 0, 0);
#pragma unroll
  for (int i = 0; i < 3; ++i) dma_w(i, 0, 0);
  if (nt > 1) {
#pragma unroll
    for (int i = 0; i < NA; ++i) dma_a(i, 1, 1);
    if constexpr (WLEAD == 2) {
#pragma unroll
      for (int i = 0; i < 3; ++i) dma_w(i, 1, 1);
    }
  }
  // behind the first LDS-DMA pieces: the partials' round trip runs beside the operand fetch the loop start waits for anyway (hipcc
  // waits vmcnt(0) for these ordinary loads, i.e. for the pieces issued so far as well: stage 1 lands ~0.2 us after stage 0)
  if constexpr (LN) {
    __builtin_amdgcn_sched_barrier(0);
    ln_stage_tile(ln_lds, p.ln_stats, p.ln_ld, p.ln_nb, p.ln_eps, p.M, row0, BM, p.cs, p.cv, col0, BN, tid, G::NT);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (nt > 1) {
    if constexpr (WLEAD == 2) {
      asm volatile("s_waitcnt vmcnt(5)" ::: "memory");   // stage 1 (2 A + 3 W pieces) may still be in flight
    } else if constexpr (NA == 4) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if constexpr (LN) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's part of the (mu, rstd) / (cs, cv) image
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  int sa = 0, sw = 0;
  uint64_t st_read = 0, st_mfma = 0, st_wait = 0, st_bar = 0, st_t0 = 0, st_begin = 0;
  if constexpr (STAMP) { st_begin = __builtin_amdgcn_s_memtime(); st_t0 = st_begin; }
  for (int t = 0; t < nt; ++t) {
    const int sa1 = sa == 2 ? 0 : sa + 1, sa2 = sa1 == 2 ? 0 : sa1 + 1;
    const int sw1 = WLEAD == 2 ? sa1 : (sw ^ 1);          // with the two-stage lead W uses the same 3-slot ring as A
    const int swn = WLEAD == 2 ? sa2 : sw1;               // slot the W pieces issued in this stage go to
    const int tw = t + WLEAD;                             // ... and their stage
    const bool n2 = t + 2 < nt;
    const bool n1 = WLEAD == 2 ? n2 : (t + 1 < nt);       // "W pieces are issued in this stage"
    const char* ab = smem + sa * A_SLOT;
    const char* wb = smem + sw * W_SLOT;
    bf16x8 x0, x1, x2, x3, w0, w1, w2, v0, v1, v2;  // x: A fragments of the current k-step; w / v: W fragments of k-step 0 / 1
    if constexpr (MF) {
      // (w0..w2, v0..v2) = the six column blocks, x0..x3 = token blocks 0..3, reloaded with blocks 4..7 behind their MFMAs
      x0 = *reinterpret_cast<const bf16x8*>(ab + xo);
      w0 = *reinterpret_cast<const bf16x8*>(wb + wo);
      w1 = *reinterpret_cast<const bf16x8*>(wb + wo + 1024);
      w2 = *reinterpret_cast<const bf16x8*>(wb + wo + 2048);
      v0 = *reinterpret_cast<const bf16x8*>(wb + wo + 3072);
      v1 = *reinterpret_cast<const bf16x8*>(wb + wo + 4096);
      v2 = *reinterpret_cast<const bf16x8*>(wb + wo + 5120);
      x1 = *reinterpret_cast<const bf16x8*>(ab + xo + 1024);
      x2 = *reinterpret_cast<const bf16x8*>(ab + xo + 2048);
      x3 = *reinterpret_cast<const bf16x8*>(ab + xo + 3072);
      __builtin_amdgcn_sched_barrier(0);
#define G16_ROW(i_, X_)                                                                                \
  acc16[i_][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, X_, acc16[i_][0], 0, 0, 0);               \
  acc16[i_][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, X_, acc16[i_][1], 0, 0, 0);               \
  acc16[i_][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, X_, acc16[i_][2], 0, 0, 0);               \
  acc16[i_][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v0, X_, acc16[i_][3], 0, 0, 0);               \
  acc16[i_][4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, X_, acc16[i_][4], 0, 0, 0);               \
  acc16[i_][5] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v2, X_, acc16[i_][5], 0, 0, 0)
      G16_ROW(0, x0);
      __builtin_amdgcn_sched_barrier(0);
      x0 = *reinterpret_cast<const bf16x8*>(ab + xo + 4096);
      if (n1) { dma_w(0, tw, swn); dma_w(1, tw, swn); }
      __builtin_amdgcn_sched_barrier(0);
      G16_ROW(1, x1);
      __builtin_amdgcn_sched_barrier(0);
      x1 = *reinterpret_cast<const bf16x8*>(ab + xo + 5120);
      if (n1) dma_w(2, tw, swn);
      if (n2) dma_a(0, t + 2, sa2);
      __builtin_amdgcn_sched_barrier(0);
      G16_ROW(2, x2);
      __builtin_amdgcn_sched_barrier(0);
      x2 = *reinterpret_cast<const bf16x8*>(ab + xo + 6144);
      if (n2) {
        dma_a(1, t + 2, sa2);
        if constexpr (NA > 2) dma_a(2 < NA ? 2 : 0, t + 2, sa2);
      }
      __builtin_amdgcn_sched_barrier(0);
      G16_ROW(3, x3);
      __builtin_amdgcn_sched_barrier(0);
      x3 = *reinterpret_cast<const bf16x8*>(ab + xo + 7168);
      if constexpr (NA > 3) {
        if (n2) dma_a(3 < NA ? 3 : 0, t + 2, sa2);
      }
      __builtin_amdgcn_sched_barrier(0);
      G16_ROW(4, x0);
      G16_ROW(5, x1);
      G16_ROW(6, x2);
      G16_ROW(7, x3);
#undef G16_ROW
    } else {
    x0 = *reinterpret_cast<const bf16x8*>(ab + xo);
    w0 = *reinterpret_cast<const bf16x8*>(wb + wo);
    w1 = *reinterpret_cast<const bf16x8*>(wb + wo + 2048);
    w2 = *reinterpret_cast<const bf16x8*>(wb + wo + 4096);
    x1 = *reinterpret_cast<const bf16x8*>(ab + xo + 2048);
    x2 = *reinterpret_cast<const bf16x8*>(ab + xo + 4096);
    x3 = *reinterpret_cast<const bf16x8*>(ab + xo + 6144);
    v0 = *reinterpret_cast<const bf16x8*>(wb + (wo ^ 32));
    v1 = *reinterpret_cast<const bf16x8*>(wb + (wo ^ 32) + 2048);
    v2 = *reinterpret_cast<const bf16x8*>(wb + (wo ^ 32) + 4096);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (STAMP) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const uint64_t now = __builtin_amdgcn_s_memtime();
      st_read += now - st_t0; st_t0 = now;
      __builtin_amdgcn_sched_barrier(0);
    }
#define G2_ROW(i_, X_, W0_, W1_, W2_)                                                                  \
  acc[i_][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W0_, X_, acc[i_][0], 0, 0, 0);                  \
  acc[i_][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W1_, X_, acc[i_][1], 0, 0, 0);                  \
  acc[i_][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2_, X_, acc[i_][2], 0, 0, 0)
    // k-step 0: after each m-block's three MFMAs its A fragment register is reloaded with the k-step-1 fragment; the
    // DMA pieces of the next stages are slotted in between the m-blocks (behind MFMAs of this wave)
    G2_ROW(0, x0, w0, w1, w2);
    __builtin_amdgcn_sched_barrier(0);
    x0 = *reinterpret_cast<const bf16x8*>(ab + (xo ^ 32));
    if (n1) { dma_w(0, tw, swn); dma_w(1, tw, swn); }
    __builtin_amdgcn_sched_barrier(0);
    G2_ROW(1, x1, w0, w1, w2);
    __builtin_amdgcn_sched_barrier(0);
    x1 = *reinterpret_cast<const bf16x8*>(ab + (xo ^ 32) + 2048);
    if (n1) dma_w(2, tw, swn);
    if (n2) dma_a(0, t + 2, sa2);
    __builtin_amdgcn_sched_barrier(0);
    G2_ROW(2, x2, w0, w1, w2);
    __builtin_amdgcn_sched_barrier(0);
    x2 = *reinterpret_cast<const bf16x8*>(ab + (xo ^ 32) + 4096);
    if (n2) {
      dma_a(1, t + 2, sa2);
      if constexpr (NA > 2) dma_a(2 < NA ? 2 : 0, t + 2, sa2);
    }
    __builtin_amdgcn_sched_barrier(0);
    G2_ROW(3, x3, w0, w1, w2);
    __builtin_amdgcn_sched_barrier(0);
    x3 = *reinterpret_cast<const bf16x8*>(ab + (xo ^ 32) + 6144);
    if constexpr (NA > 3) {
      if (n2) dma_a(3 < NA ? 3 : 0, t + 2, sa2);
    }
    __builtin_amdgcn_sched_barrier(0);
    // k-step 1
    G2_ROW(0, x0, v0, v1, v2);
    G2_ROW(1, x1, v0, v1, v2);
    G2_ROW(2, x2, v0, v1, v2);
    G2_ROW(3, x3, v0, v1, v2);
#undef G2_ROW
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (STAMP) {
      const uint64_t now = __builtin_amdgcn_s_memtime();
      st_mfma += now - st_t0; st_t0 = now;
      __builtin_amdgcn_sched_barrier(0);
    }
    if (n2) {
      if constexpr (WLEAD == 2) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
      else if constexpr (NA == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (STAMP) {
      const uint64_t now = __builtin_amdgcn_s_memtime();
      st_wait += now - st_t0; st_t0 = now;
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_barrier();  // stage t+1 has landed everywhere; nobody reads the slots of stage t any more
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (STAMP) {
      const uint64_t now = __builtin_amdgcn_s_memtime();
      st_bar += now - st_t0; st_t0 = now;
      __builtin_amdgcn_sched_barrier(0);
    }
    sa = sa1;
    sw = sw1;
  }

  if constexpr (EPI == EPI_F32_SLICES) {
    // fp32 K-slice partials straight from the accumulators, stored TRANSPOSED: the row operand is the weight (row = output
    // feature n), the column operand the few activation rows (column = m), and the consumer wants [m][n].  For a fixed
    // accumulator register the 32 lanes of a half wave hold 32 consecutive n of one m: a 128-byte segment per store.
    float* o32 = p.out32 + (int64_t)kslice * p.slab;
    const int ncol0 = col0 + wn * 96;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = row0 + wm * 128 + i * 32 + l31;
      if (n < p.M) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int m = ncol0 + j * 32 + 8 * g + 4 * hi + r;
              if (m < p.rows_per_sample) o32[(int64_t)m * p.ldo32 + n] = acc[i][j][4 * g + r];   // (rows_per_sample = real activation rows)
            }
      }
    }
    return;
  }
  // ---- epilogue in two passes of 64 rows per wave (m-blocks {0,1}, then {2,3}); the per-wave LDS image is private to the
  // wave, so only wave-local ordering is needed between its writes and reads (the barrier above covers the staging reads)
  char* st = smem + wave * OUT_WAVE_BYTES;
  const int ncol0 = col0 + wn * 96;
  uint2 bb[3][4];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) bb[j][g] = make_uint2(0, 0);
  if (p.bias != nullptr && EPI != EPI_LN_BIAS && EPI != EPI_LN_GELU && !MF) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) bb[j][g] = *reinterpret_cast<const uint2*>(p.bias + ncol0 + j * 32 + 8 * g + 4 * hi);
  }
  uint2 bb16[6];   // MF: bias of this lane's column quads 16 j + 4 lq ..
#pragma unroll
  for (int j = 0; j < 6; ++j) bb16[j] = make_uint2(0, 0);
  if (MF && p.bias != nullptr && EPI != EPI_LN_BIAS && EPI != EPI_LN_GELU) {
#pragma unroll
    for (int j = 0; j < 6; ++j) bb16[j] = *reinterpret_cast<const uint2*>(p.bias + ncol0 + j * 16 + 4 * lq);
  }
  const bool full = row0 + BM <= p.M;
#pragma unroll
  for (int ih = 0; ih < 2; ++ih) {
    const int wrow0 = row0 + wm * 128 + ih * 64;
    if constexpr (MF) {
      // ---- 16x16x32 accumulators: lane = token 16 i2 + l15 of this 64-row pass, columns 16 j + 4 lq .. + 3
      if constexpr (LN) {
        float mu[4], rs[4];
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
          const float2 ms = ln_lds[wm * 128 + ih * 64 + i2 * 16 + l15];   // staged before the K loop
          mu[i2] = ms.x;
          rs[i2] = ms.y;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const float4* cp = reinterpret_cast<const float4*>(ln_lds + BM + wn * 96 + j * 16 + 4 * lq);   // (cs, cv) of four columns
          const float4 a = cp[0], b = cp[1];
          const float c_s[4] = {a.x, a.z, b.x, b.z}, c_v[4] = {a.y, a.w, b.y, b.w};
#pragma unroll
          for (int i2 = 0; i2 < 4; ++i2) {
            const int i = ih * 4 + i2;
            const float nmu = -mu[i2], r_ = rs[i2];
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaf(r_, fmaf(nmu, c_s[r], acc16[i][j][r]), c_v[r]);
            if (EPI == EPI_LN_GELU) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = gelu_tanh(v[r]);
            }
            uint2 o;
            o.x = pack2bf(v[0], v[1]);
            o.y = pack2bf(v[2], v[3]);
            *reinterpret_cast<uint2*>(st + (i2 * 16 + l15) * OUT_ROW_BYTES + (j * 16 + 4 * lq) * 2) = o;
          }
        }
      } else {
        constexpr bool GATED16 = EPI == EPI_GATE_RES || EPI == EPI_GATE_RES_STATS;
        if (GATED16 && p.res != nullptr) {
          // The residual rows of this pass go INTO the wave's image first (two bursts of six 16-byte units per lane: the accumulators of
          // both passes are live), so that the accumulator pass sees x_new = bf16(bf16(u) + res) where a lane owns a token — the two
          // roundings of the 32x32x16 epilogues — and the store phase is a copy.
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            uint4 r6[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
              const int q = lane + 64 * (6 * half + k);
              const int m_local = q / 12, c = q - m_local * 12;
              int grow = wrow0 + m_local;
              grow = grow < p.M ? grow : p.M - 1;
              r6[k] = *reinterpret_cast<const uint4*>(p.res + (int64_t)grow * p.ldr + ncol0 + c * 8);
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) {
              const int q = lane + 64 * (6 * half + k);
              const int m_local = q / 12, c = q - m_local * 12;
              *reinterpret_cast<uint4*>(st + m_local * OUT_ROW_BYTES + c * 16) = r6[k];
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
          const int i = ih * 4 + i2;
          const int m_local = i2 * 16 + l15;
          const bf16_t* gate_row = nullptr;
          if (GATED16 && p.gate != nullptr) {
            int grow = wrow0 + m_local;
            grow = grow < p.M ? grow : p.M - 1;
            const int sample = grow / p.rows_per_sample;
            gate_row = p.gate + (int64_t)sample * p.gate_stride + ncol0 + 4 * lq;
            if (p.seg_split > 0 && grow - sample * p.rows_per_sample < p.seg_split) gate_row += p.gate_alt;
          }
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            uint2 gg = make_uint2(0x3f803f80u, 0x3f803f80u);  // bf16 1.0 pairs
            if (GATED16 && gate_row != nullptr) gg = *reinterpret_cast<const uint2*>(gate_row + j * 16);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc16[i][j][r];
            v[0] += bflo(bb16[j].x); v[1] += bfhi(bb16[j].x); v[2] += bflo(bb16[j].y); v[3] += bfhi(bb16[j].y);
            if (EPI == EPI_BIAS_GELU) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = gelu_tanh(v[r]);
            }
            if (GATED16) {
              v[0] *= bflo(gg.x); v[1] *= bfhi(gg.x); v[2] *= bflo(gg.y); v[3] *= bfhi(gg.y);
            }
            uint2 o;
            o.x = pack2bf(v[0], v[1]);
            o.y = pack2bf(v[2], v[3]);
            if (GATED16 && p.res != nullptr) {
              const uint2 rr = *reinterpret_cast<const uint2*>(st + m_local * OUT_ROW_BYTES + (j * 16 + 4 * lq) * 2);
              o.x = pack2bf(bflo(o.x) + bflo(rr.x), bfhi(o.x) + bfhi(rr.x));
              o.y = pack2bf(bflo(o.y) + bflo(rr.y), bfhi(o.y) + bfhi(rr.y));
            }
            *reinterpret_cast<uint2*>(st + m_local * OUT_ROW_BYTES + (j * 16 + 4 * lq) * 2) = o;
          }
        }
        if constexpr (EPI == EPI_GATE_RES_STATS) {
          // LayerNorm partials of the 64 rows of this pass from the image: a lane pair per row, the 48 columns of a lane in the order
          // of the 32x32x16 accumulator pass of gemm_kernel (same bits as ln_row_stats gives for these rows)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int i2 = 0; i2 < 2; ++i2) {
            const int m_local = i2 * 32 + l31;
            LnAcc lacc;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const uint2 o = *reinterpret_cast<const uint2*>(st + m_local * OUT_ROW_BYTES + (j * 32 + 8 * g + 4 * hi) * 2);
                if (j == 0 && g == 0) lacc.init(bflo(o.x));
                lacc.add(bflo(o.x)); lacc.add(bfhi(o.x)); lacc.add(bflo(o.y)); lacc.add(bfhi(o.y));
              }
            const float2 mine = lacc.finish(48.f);
            const float2 other = make_float2(__shfl_xor(mine.x, 32, 64), __shfl_xor(mine.y, 32, 64));
            const float2 blk = ln_merge_equal(mine, other, 48.f);
            const int grow = wrow0 + m_local;
            if (hi == 0 && grow < p.M) p.stats_out[(int64_t)(ncol0 / LN_BLOCK) * p.stats_ld + grow] = blk;
          }
        }
      }
    } else {
    if constexpr (LN) {
      // AdaLN folded into the GEMM: the operand rows were the RAW residual stream and W = bf16(W0 (1 + scale)); the LayerNorm of
      // row m and the shift enter here:  out = rstd_m (acc - mu_m cs[n]) + cv[n]  (vsys_internal.h GemmParams).
      float mu[2], rs[2];
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) {
        const float2 ms = ln_lds[wm * 128 + ih * 64 + i2 * 32 + l31];   // staged before the K loop
        mu[i2] = ms.x;
        rs[i2] = ms.y;
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float4 c_s[4], c_v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {   // (cs, cv) pairs of four consecutive columns
          const float4* cp = reinterpret_cast<const float4*>(ln_lds + BM + wn * 96 + j * 32 + 8 * g + 4 * hi);
          const float4 a = cp[0], b = cp[1];
          c_s[g] = make_float4(a.x, a.z, b.x, b.z);
          c_v[g] = make_float4(a.y, a.w, b.y, b.w);
        }
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
          const int i = ih * 2 + i2;
          const int m_local = i2 * 32 + l31;
          const float nmu = -mu[i2], r_ = rs[i2];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n_local = j * 32 + 8 * g + 4 * hi;
            float v[4];
            v[0] = fmaf(r_, fmaf(nmu, c_s[g].x, acc[i][j][4 * g + 0]), c_v[g].x);
            v[1] = fmaf(r_, fmaf(nmu, c_s[g].y, acc[i][j][4 * g + 1]), c_v[g].y);
            v[2] = fmaf(r_, fmaf(nmu, c_s[g].z, acc[i][j][4 * g + 2]), c_v[g].z);
            v[3] = fmaf(r_, fmaf(nmu, c_s[g].w, acc[i][j][4 * g + 3]), c_v[g].w);
            if (EPI == EPI_LN_GELU) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = gelu_tanh(v[r]);
            }
            uint2 o;
            o.x = pack2bf(v[0], v[1]);
            o.y = pack2bf(v[2], v[3]);
            *reinterpret_cast<uint2*>(st + m_local * OUT_ROW_BYTES + n_local * 2) = o;
          }
        }
      }
    }
#pragma unroll
    for (int i2 = 0; i2 < (LN ? 0 : 2); ++i2) {
      const int i = ih * 2 + i2;
      const int m_local = i2 * 32 + l31;
      const bf16_t* gate_row = nullptr;
      if (EPI == EPI_GATE_RES && p.gate != nullptr) {
        int grow = wrow0 + m_local;
        grow = grow < p.M ? grow : p.M - 1;
        const int sample = grow / p.rows_per_sample;
        gate_row = p.gate + (int64_t)sample * p.gate_stride + ncol0 + 4 * hi;
        if (p.seg_split > 0 && grow - sample * p.rows_per_sample < p.seg_split) gate_row += p.gate_alt;
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        uint2 gg[4];  // gate of this 32-column block only (register budget: 192 accumulators are live)
        if (EPI == EPI_GATE_RES) {
#pragma unroll
          for (int g = 0; g < 4; ++g) gg[g] = make_uint2(0x3f803f80u, 0x3f803f80u);  // bf16 1.0 pairs
          if (gate_row != nullptr) {
#pragma unroll
            for (int g = 0; g < 4; ++g) gg[g] = *reinterpret_cast<const uint2*>(gate_row + j * 32 + 8 * g);
          }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n_local = j * 32 + 8 * g + 4 * hi;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[i][j][4 * g + r];
          v[0] += bflo(bb[j][g].x); v[1] += bfhi(bb[j][g].x); v[2] += bflo(bb[j][g].y); v[3] += bfhi(bb[j][g].y);
          if (EPI == EPI_BIAS_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_tanh(v[r]);
          }
          if (EPI == EPI_GATE_RES) {
            v[0] *= bflo(gg[g].x); v[1] *= bfhi(gg[g].x); v[2] *= bflo(gg[g].y); v[3] *= bfhi(gg[g].y);
          }
          uint2 o;
          o.x = pack2bf(v[0], v[1]);
          o.y = pack2bf(v[2], v[3]);
          *reinterpret_cast<uint2*>(st + m_local * OUT_ROW_BYTES + n_local * 2) = o;
        }
      }
    }
    }
    // residual rows are fetched only now: the accumulators of this pass are dead, so the 48 registers are free (the
    // other workgroup of the CU covers the latency)
    uint4 rres[12];
    if (EPI == EPI_GATE_RES && !MF) {
#pragma unroll
      for (int it = 0; it < 12; ++it) rres[it] = make_uint4(0, 0, 0, 0);
      if (p.res != nullptr) {
#pragma unroll
        for (int it = 0; it < 12; ++it) {
          const int q = lane + 64 * it;
          const int m_local = q / 12, c = q - m_local * 12;
          int grow = wrow0 + m_local;
          grow = grow < p.M ? grow : p.M - 1;
          rres[it] = *reinterpret_cast<const uint4*>(p.res + (int64_t)grow * p.ldr + ncol0 + c * 8);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    uint4 val[12];
#pragma unroll
    for (int it = 0; it < 12; ++it) {
      const int q = lane + 64 * it;
      const int m_local = q / 12, c = q - m_local * 12;
      val[it] = *reinterpret_cast<const uint4*>(st + m_local * OUT_ROW_BYTES + c * 16);
    }
#pragma unroll
    for (int it = 0; it < 12; ++it) {
      const int q = lane + 64 * it;
      const int m_local = q / 12, c = q - m_local * 12;
      const int64_t grow = wrow0 + m_local;
      const int gcol = ncol0 + c * 8;
      const bool ok = full || grow < p.M;
      uint4 v = val[it];
      if (EPI == EPI_GATE_RES && !MF) {   // (MF: the image already holds x_new; launch_gemm2 takes no aux there)
        if (p.aux != nullptr && ok) *reinterpret_cast<uint4*>(p.aux + grow * p.ldaux + gcol) = v;
        if (p.res != nullptr) {
          float a[8], b[8];
          unpack8(v, a);
          unpack8(rres[it], b);
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] += b[e];
          v = pack8(a);
        }
      }
      if (ok) *reinterpret_cast<uint4*>(p.out + grow * p.ldo + gcol) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // image reads done before the next pass overwrites it
    __builtin_amdgcn_wave_barrier();
  }
  if constexpr (STAMP) {
    if (lane == 0 && p.aux != nullptr) {
      int64_t* d = reinterpret_cast<int64_t*>(p.aux) + ((int64_t)blockIdx.x * G::NW + wave) * 8;
      const uint64_t end = __builtin_amdgcn_s_memtime();
      d[0] = (int64_t)st_read; d[1] = (int64_t)st_mfma; d[2] = (int64_t)st_wait; d[3] = (int64_t)st_bar;
      d[4] = (int64_t)(st_t0 - st_begin); d[5] = (int64_t)(end - st_t0); d[6] = (int64_t)st_begin; d[7] = nt;
    }
  }
#endif
}

}  // namespace

template <int NWN, int MF = 0>
static int launch_gemm2_t(const GemmParams& p, int epi, hipStream_t stream) {
  using G = G2<NWN>;
  if (p.N % G::BN != 0) return VSYS_ERR_SHAPE;
  const int nbm = (p.M + BM - 1) / BM, nbn = p.N / G::BN;
  const int grid = nbm * nbn;
  static std::atomic<unsigned long long> attr_seen{0};   // per device (and per template instance)
  for (DeviceOnce once(attr_seen); once.todo(); once.done()) {
    (void)hipFuncSetAttribute((const void*)gemm2_kernel<EPI_BIAS, NWN, 0, MF>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm2_kernel<EPI_BIAS_GELU, NWN, 0, MF>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm2_kernel<EPI_GATE_RES, NWN, 0, MF>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm2_kernel<EPI_LN_BIAS, NWN, 0, MF>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES + G::LN_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm2_kernel<EPI_LN_GELU, NWN, 0, MF>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES + G::LN_BYTES);
    if constexpr (MF)
      (void)hipFuncSetAttribute((const void*)gemm2_kernel<EPI_GATE_RES_STATS, NWN, 0, MF>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
  }
  switch (epi) {
    case EPI_LN_BIAS: hipLaunchKernelGGL((gemm2_kernel<EPI_LN_BIAS, NWN, 0, MF>), dim3(grid), dim3(G::NT), G::LDS_BYTES + G::LN_BYTES, stream, p); break;
    case EPI_LN_GELU: hipLaunchKernelGGL((gemm2_kernel<EPI_LN_GELU, NWN, 0, MF>), dim3(grid), dim3(G::NT), G::LDS_BYTES + G::LN_BYTES, stream, p); break;
    case EPI_BIAS: hipLaunchKernelGGL((gemm2_kernel<EPI_BIAS, NWN, 0, MF>), dim3(grid), dim3(G::NT), G::LDS_BYTES, stream, p); break;
    case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm2_kernel<EPI_BIAS_GELU, NWN, 0, MF>), dim3(grid), dim3(G::NT), G::LDS_BYTES, stream, p); break;
    case EPI_GATE_RES:
      if (MF && (p.aux || p.add1 || p.add2 || p.stats_out)) return VSYS_ERR_ARG;   // (the 16x16x32 gated epilogue stores x_new only)
      hipLaunchKernelGGL((gemm2_kernel<EPI_GATE_RES, NWN, 0, MF>), dim3(grid), dim3(G::NT), G::LDS_BYTES, stream, p); break;
    case EPI_GATE_RES_STATS:
      if constexpr (MF) {
        hipLaunchKernelGGL((gemm2_kernel<EPI_GATE_RES_STATS, NWN, 0, MF>), dim3(grid), dim3(G::NT), G::LDS_BYTES, stream, p);
        break;
      }
      return VSYS_ERR_ARG;
    default: return VSYS_ERR_ARG;
  }
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

#ifdef VSYS_LAB
int launch_gemm2_stamp(const GemmParams& p_, hipStream_t stream) {
  using G = G2<4>;
  GemmParams p = p_;
  p.aux = reinterpret_cast<bf16_t*>(get_lab_debug_buffer());  // set with vsys_lab_flash_debug_buffer: int64[blocks * 8 * 8]
  if (p.N % G::BN != 0 || p.aux == nullptr) return VSYS_ERR_SHAPE;
  const int grid = ((p.M + BM - 1) / BM) * (p.N / G::BN);
  (void)hipFuncSetAttribute((const void*)gemm2_kernel<EPI_BIAS, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
  hipLaunchKernelGGL((gemm2_kernel<EPI_BIAS, 4, 1>), dim3(grid), dim3(G::NT), G::LDS_BYTES, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

#endif  // VSYS_LAB

// Weight-streaming linears (few activation rows against a large weight: ops.linear_skinny, T5 at 300 tokens).  The WEIGHT is the
// row operand "A" [M = features, K], the activations the column operand "W" [N = rows padded to 384, K]: on the 256 x 384 tile a
// weight panel is read by exactly one workgroup and ALL activation rows ride along (2.5 x the weight bytes through LDS instead
// of 4.2 x on the 128-column kernel), K is cut in ``slices`` so that ~one workgroup per CU pulls on HBM, and every
// slice leaves fp32 partials out32[s][n_row = activation row][feature] for vsys_splitk_reduce.
int launch_gemm2_slices(const GemmParams& p, int slices, hipStream_t stream) {
  using G = G2<4>;
  if (p.M <= 0) return 0;
  if (p.N % G::BN != 0 || slices < 1 || slices > 65535 || p.ks <= 0 || p.ks % BK != 0 || p.K % BK != 0 || (int64_t)p.ks * slices < p.K ||
      (int64_t)p.ks * (slices - 1) >= p.K || p.out32 == nullptr || (p.lda % 8) || (p.ldw % 8) || p.rows_per_sample <= 0)
    return VSYS_ERR_SHAPE;
  if (p.lda * 512 + (int64_t)p.K * 2 >= 0x7fffffff || p.ldw * 768 + (int64_t)p.K * 2 >= 0x7fffffff) return VSYS_ERR_SHAPE;
  const int nbm = (p.M + BM - 1) / BM, nbn = p.N / G::BN;
  static std::atomic<unsigned long long> attr_seen{0};
  for (DeviceOnce once(attr_seen); once.todo(); once.done())
    (void)hipFuncSetAttribute((const void*)gemm2_kernel<EPI_F32_SLICES, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
  hipLaunchKernelGGL((gemm2_kernel<EPI_F32_SLICES, 4>), dim3(nbm * nbn * slices), dim3(G::NT), G::LDS_BYTES, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

// wide = 0: 256 x 192 tile, two workgroups per CU (variant 20); wide = 1: 256 x 384 tile, one 8-wave workgroup per CU (variant 30)
// wide = 2: the 256 x 192 tile on v_mfma_f32_16x16x32_bf16 (gemm2_kernel MF = 1)
int launch_gemm2(const GemmParams& p, int epi, int wide, hipStream_t stream) {
  if (wide == 2) return launch_gemm2_t<2, 1>(p, epi, stream);
  if (wide == 3) return launch_gemm2_t<4, 1>(p, epi, stream);   // the 256 x 384 tile on 16x16x32 (A/B id 34)
  return wide ? launch_gemm2_t<4>(p, epi, stream) : launch_gemm2_t<2>(p, epi, stream);
}

}  // namespace vsys
