// bf16 MFMA GEMM for the STDiT3 linears (nn.Linear semantics: out = x @ W^T + b) with fused epilogues.
//
// Replaces the torch/cuBLAS GEMM + separate elementwise kernels at these reference call sites
// (/root/reference/videosys):
//   models/modules/attentions.py:59      qkv            (EPI_BIAS)
//   models/modules/attentions.py:107 +   proj, then  x + gate_msa * proj(...)   open_sora_transformer_3d.py:219,228 (EPI_GATE_RES)
//   models/modules/attentions.py:156,183 cross q_linear (EPI_BIAS) / proj + residual (EPI_GATE_RES, gate = null)
//   timm Mlp fc1 + GELU(tanh)            open_sora_transformer_3d.py:130-132,267 (EPI_BIAS_GELU)
//   timm Mlp fc2, x + gate_mlp * (...)   open_sora_transformer_3d.py:270,284     (EPI_GATE_RES)
//
// Design (gfx950): 256x192x64 block tile, 512 threads = 8 waves as 4(M) x 2(N), each wave a 64x96 output tile of
// 2x3 v_mfma_f32_32x32x16_bf16 accumulators (96 acc VGPRs).  The MFMA is issued "swapped" (A operand = W fragment,
// B operand = X fragment) so each lane ends up with ONE token row and runs of 4 consecutive output columns: bias/gate
// broadcast along columns and the bf16 pack is a plain 8-byte LDS write.  Operand tiles are staged HBM/L2 -> VGPR ->
// LDS (double-buffered, loads for tile t+1 issued before the MFMAs of tile t, one barrier per K-tile) in a 128-byte-row
// image whose 16-byte slots are XOR-swizzled by (row>>1)&7 so every ds_read_b128 lane group hits 16 distinct slots.
// The output tile goes back through LDS so HBM sees whole 192-byte row segments (16-byte stores), where the residual
// is added and the optional PAB cache copy is written.  1-D grid with an XCD-aware bijective remap: consecutive
// logical tiles (same token panel, different column tile) share an XCD's L2.
#include <map>
#include <mutex>

#include "common.h"
#include "vsys_internal.h"

namespace vsys {

namespace {

constexpr int BN = 192, BK = 64;
constexpr int B_BYTES = BN * BK * 2;            // 24576
constexpr int OUT_ROW_BYTES = 96 * 2 + 16;      // per-wave epilogue image row (padded)
constexpr int OUT_WAVE_BYTES = 64 * OUT_ROW_BYTES;  // 13312

// Geometry for a BM_ x 192 x 64 block tile: BM_/64 (M) x 2 (N) waves, every wave a 64 x 96 output tile.
//   BM_ = 256: 8 waves, 112 KiB of staging, one workgroup per CU (two waves per SIMD from ONE workgroup);
//   BM_ = 128: 4 waves,  80 KiB of staging, two workgroups per CU (the two waves of a SIMD belong to DIFFERENT
//              workgroups, so one workgroup's barrier / epilogue overlaps the other's MFMAs).
template <int BM_>
struct Geo {
  static constexpr int NW = BM_ / 32;
  static constexpr int NT = NW * 64;
  static constexpr int A_BYTES = BM_ * BK * 2;
  static constexpr int STAGE = A_BYTES + B_BYTES;
  static constexpr int NA = 4;             // A LDS-DMA pieces (8 rows x 128 B) per wave per K-tile: BM_/NW/8
  static constexpr int NB = BN / NW / 8;   // B pieces per wave per K-tile: 3 or 6
  static_assert(NW * OUT_WAVE_BYTES <= 2 * STAGE, "epilogue image must fit in the staging buffers");
};

// Variant id = BASE + 10 * ABL.  BASE selects the main-loop schedule:
//   3: LDS-DMA double buffer, the pieces of tile t+1 issued back-to-back at the top of tile t, one barrier per tile;
//   6: fragment registers double-buffered by hand (k-step s+1 is read while k-step s multiplies), the tile barrier
//      moved in front of the LAST k-step so the first fragments of tile t+1 are read under the MFMAs of tile t, and
//      the DMA pieces of tile t+2 interleaved one by one with the MFMAs (a piece holds its wave ~60-180 cycles at
//      issue; behind an MFMA of the same wave that time is covered by the matrix pipe);
//   8: schedule 6 on THREE A slots + TWO W slots with counted vmcnt waits (A(t+2) and W(t+1) in flight during tile t)
//      — the shipped default (measured at config 2: qkv 0.33 -> 0.31 ms, fc2 0.44 -> 0.39 ms over schedule 3).
// ABL (lab builds only): 1 = no epilogue (accumulators kept alive), 2 = no LDS-DMA inside the loop, 3 = every tile reads
// the A rows of tile 0 (A becomes L2-resident: separates HBM-miss latency from DMA issue / LDS-write cost), 4 = epilogue
// arithmetic without the HBM stores, 5 = streaming (nt) stores.
// PROD = 1 (schedule 8 only, variant 28): four extra PRODUCER waves (one per SIMD) issue every LDS-DMA piece; the eight consumer
// waves only read fragments and issue MFMAs.  A piece costs its issuing wave 60-185 cycles (MI355X_MICROARCH.md), 7 pieces per
// wave and tile against 24 MFMAs = 768 cycles: with the DMA on a third wave of the SIMD that time no longer comes out of an
// MFMA-issuing wave.  Needs <= 168 VGPRs (three waves per SIMD), which the store-only epilogues meet (165).
// MF = 1 (round 6; schedule 8, 256-row geometry): the same tile walk on v_mfma_f32_16x16x32_bf16 — the matrix pipe is ~12 % cheaper per
// flop in that shape under the 1400 W cap (gemm2_bf16.hip MF has the measurements), and the results are the SAME BITS.  A K-tile of 64 is
// two k-steps of 32; a k-step is 4 token blocks x 6 column blocks of 16 x 16 and runs as two PHASES of two token blocks (12 MFMAs = 192
// pipe cycles = one k-step of the 32x32x16 loop, so the LDS-DMA pieces keep their places).  Fragment registers: W sets A / B (six
// column blocks each, alternating per k-step) and X pairs A / B (two token blocks, alternating per phase), every set read one phase
// ahead of its use.  A lane owns token 16 i + lane % 16 and the columns 16 j + 4 (lane / 16) .. + 3: the epilogues index by that;
// the statistics of EPI_GATE_RES_STATS are taken from the wave's LDS image in the canonical 48-column order (same bits).
// KS = 1 (round 6; 128-row geometry on 16x16x32 only): TWO workgroups per tile, each walks half of the K-tiles.  For problems with fewer
// tiles than CUs (one rank of an 8-way DSP group: 4864 rows, 228 tiles) a CU then holds two workgroups whose LDS-DMA round trips,
// barriers and epilogues overlap, and every K loop is half as long.  Workgroup 2 i (the first K half, dispatched first, same XCD as its
// partner) dumps its fp32 accumulators with write-through stores, drains them and raises one flag per wave; workgroup 2 i + 1 adds
// them to its own (second-half) sums — wave w, lane l reads exactly what wave w, lane l wrote — lowers the flags again (the next
// launch, also a replayed one, finds them down) and runs the epilogue.  fp32 summation order differs from the unsplit kernels
// (a + b instead of one running sum): deterministic, not bit-identical; dispatched only for tile counts <= the CU count.
template <int EPI, int PIPE, int BM_, int RASTER = 1, int PROD = 0, int MF = 0, int KS = 0>
__global__ __launch_bounds__(Geo<BM_>::NT + PROD * 256, PROD ? 3 : 2) void gemm_kernel(GemmParams p) {
#if __HIP_DEVICE_COMPILE__  // the buffer-resource type below exists in the device pass only; the host pass needs just the stub
  using G = Geo<BM_>;
  constexpr int BASE = PIPE % 10, ABL = PIPE / 10;
  constexpr int A_BYTES = G::A_BYTES, STAGE_BYTES = G::STAGE, NA = G::NA, NB = G::NB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  const int l15 = lane & 15, lq = lane >> 4;   // MF: fragment row / 16-byte k-chunk; accumulator token / column quad
  static_assert(!MF || ((PIPE == 8 && !PROD) || ((PIPE == 3 || PIPE == 9) && BM_ == 128)), "the 16x16x32 form exists for schedule 8 and for schedules 3 / 9 on the 128-row geometry");

  static_assert(!KS || (MF && PIPE == 3 && BM_ == 128), "split K exists for the 128-row geometry on 16x16x32");
  const int nbn = p.N / BN;
  // KS: blockIdx = 16 g + 8 half + x  <->  pair 8 g + x, K half: both halves of a pair on XCD x, the dumping half dispatched first
  const int ks_half = KS ? (int)((blockIdx.x >> 3) & 1) : 0;
  const int ks_pair = KS ? (int)((blockIdx.x >> 4) * 8 + (blockIdx.x & 7)) : 0;
  const int ks_ntiles = ((p.M + BM_ - 1) / BM_) * nbn;
  if (KS && ks_pair >= ks_ntiles) return;
  const int tile = KS ? xcd_remap(ks_pair, ks_ntiles) : xcd_remap(blockIdx.x, gridDim.x);
  // Raster (RASTER = 1): the row panels are split into 8 contiguous groups, one per XCD chunk of the remap; inside a group
  // the order is column-GROUP major (GW = 6 column tiles = 2.65 MB of W at K = 1152): for g: for panel: for j.  The 32
  // tiles an XCD runs at once are then ~5 row panels x 6 column tiles, so the W slab of the group stays in the XCD's
  // 4 MiB L2 for the whole sweep over the group's panels (W is the operand with the ONE-tile DMA lead) while A, the
  // operand with the two-tile lead, is re-read once per column group from L2 / MALL.  With plain row-major order every
  // wave of tiles re-reads all of W (8-10.6 MB > L2) from the fabric: 586 MB fetched for 98 MB of operands at qkv shape.
  int bm, bn;
  gemm_raster(tile, (p.M + BM_ - 1) / BM_, nbn, RASTER ? p.raster_gw : 0, p.raster_ph, bm, bn);
  const int row0 = bm * BM_, col0 = bn * BN;

  // ---- LDS-DMA staging assignment.  global_load_lds writes lane l of a wave to  M0_base + 16*l  (lane-linear), so each
  // wave instruction fills 8 whole 128-byte rows; the bank swizzle therefore sits on the SOURCE side: the lane that
  // lands in physical 16-byte slot s of row r fetches logical chunk s ^ ((r>>1)&7) (the involution the reads apply).
  // The buffer form (buffer_load_dwordx4 ... offen lds) is used on purpose: the global form is FLAT-encoded and touches two
  // address spaces, which makes hipcc's wait-count pass treat every later wait as "flat pending" and emit lgkmcnt(0)
  // instead of counted waits, draining the fragment-read pipeline at every k-step.  It also keeps the K advance in the
  // scalar offset, so a piece costs no address VALU.
  int a_off[NA], b_off[NB];  // per-lane byte offsets from the tile's first A row / W row
  int a_lds[NA], b_lds[NB];
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int r = wave_u * (NA * 8) + i * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    a_lds[i] = (wave_u * (NA * 8) + i * 8) * 128;  // wave-uniform row-block base
    const int rl = row0 + r < p.M ? r : p.M - 1 - row0;  // rows past M re-read the last row (never stored)
    a_off[i] = (rl * (int)p.lda + c * 8) * 2;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int r = wave_u * (NB * 8) + i * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    b_lds[i] = A_BYTES + (wave_u * (NB * 8) + i * 8) * 128;
    b_off[i] = (r * (int)p.ldw + c * 8) * 2;
  }
  // descriptors start at the tile's first row and end with the matrix: reads past the end return 0 instead of faulting
  const int64_t a_row0 = (ABL == 3) ? 0 : row0;  // lab: every tile streams the SAME (L2-resident) A rows
  const int64_t a_bytes = ((int64_t)(p.M - 1 - row0) * p.lda + p.K) * 2, b_bytes = ((int64_t)(p.N - 1 - col0) * p.ldw + p.K) * 2;
  const auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + a_row0 * p.lda), 0,
                                                        (int)(a_bytes < 0x7fffffff ? a_bytes : 0x7fffffff), 0x00020000);
  const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)col0 * p.ldw), 0,
                                                        (int)(b_bytes < 0x7fffffff ? b_bytes : 0x7fffffff), 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int ks_k0 = KS ? ks_half * ((p.K / BK) / 2) : 0;   // the K-tile this workgroup's range starts at
  // piece i of tile kt_ into buffer buf_ (i < NA: A rows, else W rows); i must be a constant after unrolling
  auto dma_piece = [&](int i, int kt_, int buf_) {
    char* base_ = smem + buf_ * STAGE_BYTES;
    if (i < NA)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(base_ + a_lds[i < NA ? i : 0]), 16, a_off[i < NA ? i : 0], (kt_ + ks_k0) * (BK * 2), 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(base_ + b_lds[i >= NA ? i - NA : 0]), 16, b_off[i >= NA ? i - NA : 0], (kt_ + ks_k0) * (BK * 2), 0, 0);
  };
#define GEMM_DMA_RANGE(lo_, hi_, kt_, buf_)                          \
  do {                                                               \
    _Pragma("unroll") for (int i_ = (lo_); i_ < (hi_); ++i_) dma_piece(i_, (kt_), (buf_)); \
  } while (0)
  constexpr int NP = NA + NB;  // 7 or 10 pieces per wave per K-tile

  // ---- fragment read offsets (per lane).  swz(row, chunk) = row*128 + ((chunk ^ ((row>>1)&7)) << 4) with
  // chunk = 2*ks + hi  ==  (row*128 + ((hi ^ ((row>>1)&7)) << 4)) ^ (ks << 5): one base per fragment row, the k-step is an XOR.
  int xo[2], wo[3];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = wm * 64 + i * 32 + l31;
    xo[i] = (r << 7) + ((hi ^ ((r >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int r = wn * 96 + j * 32 + l31;
    wo[j] = A_BYTES + (r << 7) + ((hi ^ ((r >> 1) & 7)) << 4);
  }

  f32x16 acc[MF ? 1 : 2][MF ? 1 : 3];
  f32x4 acc16[MF ? 4 : 1][MF ? 6 : 1];   // MF: token block i, column block j
#pragma unroll
  for (int i = 0; i < (MF ? 1 : 2); ++i)
#pragma unroll
    for (int j = 0; j < (MF ? 1 : 3); ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
  for (int i = 0; i < (MF ? 4 : 1); ++i)
#pragma unroll
    for (int j = 0; j < (MF ? 6 : 1); ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.f;
  // MF fragment offsets: row r, k-step ks (32 wide), chunk 4 ks + lq at slot (4 ks + lq) ^ ((r >> 1) & 7): one base per fragment row, the
  // k-step is an XOR of bit 6.  (The 16-lane groups of ds_read_b128 — {0-3, 12-15, 20-27}, ... — then meet rows 0-3, 12-15 with chunk c
  // and rows 4-11 with chunk c ^ 1: eight distinct slots x two row parities.)
  int xo16[4], wo16[6];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wm * 64 + i * 16 + l15;
    xo16[i] = (r << 7) + ((lq ^ ((r >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int r = wn * 96 + j * 16 + l15;
    wo16[j] = A_BYTES + (r << 7) + ((lq ^ ((r >> 1) & 7)) << 4);
  }

  auto compute = [&](int cur) {
    const char* sb_ = smem + cur * STAGE_BYTES;
    if constexpr (MF) {   // (schedule 3 on the 128-row geometry: two k-steps of 32, 4 x 6 blocks of 16 x 16 each)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 xf[4], wf[6];
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(sb_ + (xo16[i] ^ (ks << 6)));
#pragma unroll
        for (int j = 0; j < 6; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(sb_ + (wo16[j] ^ (ks << 6)));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 6; ++j)
            acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc16[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 xf[2], wf[3];
#pragma unroll
      for (int i = 0; i < 2; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(sb_ + (xo[i] ^ (ks << 5)));
#pragma unroll
      for (int j = 0; j < 3; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(sb_ + (wo[j] ^ (ks << 5)));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
    }
    }
  };

  // EPI_LN_*: (mu, rstd) of the tile's rows + (cs, cv) of its columns in LDS (common.h ln_stage_tile).  256-row tiles have the
  // room behind the staging slots and fill it here, in front of the LDS-DMA stream; 128-row tiles (two workgroups per CU, no
  // room) fill the part of the staging area the epilogue images leave free, after the K loop.
  constexpr bool LNE = EPI == EPI_LN_BIAS || EPI == EPI_LN_GELU;
  constexpr int LN_OFF = BM_ == 256 ? (BASE == 8 ? 3 * A_BYTES + 2 * B_BYTES : 2 * STAGE_BYTES) : G::NW * OUT_WAVE_BYTES;
  static_assert(BM_ == 256 || LN_OFF + (BM_ + BN) * 8 <= 2 * STAGE_BYTES, "no room for the LayerNorm image");
  float2* ln_lds = reinterpret_cast<float2*>(smem + LN_OFF);
  auto ln_prologue = [&]() {   // called right behind the first tile's LDS-DMA pieces: its global round trip runs beside them
    if constexpr (LNE && BM_ == 256) {
      ln_stage_tile(ln_lds, p.ln_stats, p.ln_ld, p.ln_nb, p.ln_eps, p.M, row0, BM_, p.cs, p.cv, col0, BN, tid, G::NT);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (ordered for the other waves by the first tile barrier below)
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // Residual rows of the gated epilogues (12 x 16 B per lane, rows clamped instead of branched): requested after the K loop, their HBM
  // latency hides under the accumulator -> LDS transposition.  Requesting them in FRONT of the loop was measured twice and bought nothing:
  // schedule 8 at 38 912 rows 0.132 vs 0.126-0.129 ms for proj; schedule 9 (one wave per SIMD, where a round trip is fully exposed)
  // 22.8 vs 22.8 us per launch in the rank-of-eight trace.  What the gated epilogue did lose there were FOUR dependent round trips for
  // the gate rows (one per token block; now one burst, below).
  constexpr bool GATED_ = EPI == EPI_GATE_RES || EPI == EPI_GATE_RES_STATS;
  constexpr bool RES_EARLY = false;
  uint4 rres[12];
  auto fetch_res = [&]() {
    if (GATED_) {
#pragma unroll
      for (int it = 0; it < 12; ++it) rres[it] = make_uint4(0, 0, 0, 0);
      if (p.res != nullptr) {
#pragma unroll
        for (int it = 0; it < 12; ++it) {
          const int q = lane + 64 * it;
          const int m_local = q / 12, c = q - m_local * 12;
          int grow = row0 + wm * 64 + m_local;
          grow = grow < p.M ? grow : p.M - 1;
          rres[it] = *reinterpret_cast<const uint4*>(p.res + (int64_t)grow * p.ldr + col0 + wn * 96 + c * 8);
        }
      }
    }
  };
  if constexpr (RES_EARLY) fetch_res();
  const int nt_all = p.K / BK;
  const int nt = KS ? (ks_half ? nt_all - nt_all / 2 : nt_all / 2) : nt_all;
  const int last = nt - 1;
  // schedule 9: ring of NS9 stages (four = the whole 160 KiB of a CU was measured too: fc2 0.0532 vs 0.0530 ms, proj 0.0214 vs 0.0204 — no gain
  // over three: with two tiles in flight a CU fills its LDS at ~70 GB/s, 0.57 us per 40-KiB K-tile, and more depth does not raise that)
  constexpr int NS9 = 3;
  auto wait9 = [&](int tiles_in_flight) {   // counted wait: everything but this wave's newest ``tiles_in_flight`` tiles has landed
    if (tiles_in_flight >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
    else if (tiles_in_flight == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  if constexpr (BASE == 9) {   // tiles 0 .. NS9 - 2 are requested, tile 0 is waited for (counted: the others stay in flight)
    static_assert(NS9 <= 4, "wait9 counts up to two tiles in flight behind the awaited one");
    const int npro = nt < NS9 - 1 ? nt : NS9 - 1;
    for (int j = 0; j < npro; ++j) GEMM_DMA_RANGE(0, NP, j, j);
    wait9(npro - 1);
    __builtin_amdgcn_s_barrier();   // raw: __syncthreads() would drain the DMA queue
  } else if constexpr (BASE != 8) {
    GEMM_DMA_RANGE(0, NP, 0, 0);
    ln_prologue();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  if constexpr (BASE == 8) {
    // ---- three A slots + two W slots (144 KiB).  A is streamed from HBM exactly once while W stays L2/MALL resident,
    // and one tile of lead (~1.5-2k cycles) does not cover an HBM miss under load: A(t+2) and W(t+1) are issued during
    // tile t, W first, so the wait in front of tile t's barrier is the COUNTED vmcnt(4) = "everything but the four newest
    // pieces (= A(t+2)) has landed".  Raw s_barrier + inline waits: a __syncthreads() would drain the DMA queue.
    // Fragment double-buffering and the barrier in front of the last k-step are as in schedule 6.
    static_assert(BM_ == 256 || BASE != 8 || MF, "schedule 8 on the 4-wave geometry exists in the 16x16x32 form only");
    constexpr int W_BASE = 3 * A_BYTES;
    auto dma_a = [&](int i, int kt_, int slot) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(smem + slot * A_BYTES + a_lds[i]), 16, a_off[i], kt_ * (BK * 2), 0, 0);
    };
    auto dma_w = [&](int i, int kt_, int slot) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(smem + W_BASE + slot * B_BYTES + (b_lds[i] - A_BYTES)), 16, b_off[i], kt_ * (BK * 2), 0, 0);
    };
    bf16x8 f0x0, f0x1, f0w0, f0w1, f0w2, f1x0, f1x1, f1w0, f1w1, f1w2;
#define V8_READ(S, sa_, sw_, ks_)                                                               \
  do {                                                                                          \
    const char* ab_ = smem + (sa_) * A_BYTES;                                                   \
    const char* wb_ = smem + W_BASE - A_BYTES + (sw_) * B_BYTES;                                \
    S##x0 = *reinterpret_cast<const bf16x8*>(ab_ + (xo[0] ^ ((ks_) << 5)));                     \
    S##x1 = *reinterpret_cast<const bf16x8*>(ab_ + (xo[1] ^ ((ks_) << 5)));                     \
    S##w0 = *reinterpret_cast<const bf16x8*>(wb_ + (wo[0] ^ ((ks_) << 5)));                     \
    S##w1 = *reinterpret_cast<const bf16x8*>(wb_ + (wo[1] ^ ((ks_) << 5)));                     \
    S##w2 = *reinterpret_cast<const bf16x8*>(wb_ + (wo[2] ^ ((ks_) << 5)));                     \
  } while (0)
#define V8_SB() __builtin_amdgcn_sched_barrier(0)
    // one k-step of set S: MFMA pair, DMA d0_, MFMA pair, DMA d1_, MFMA pair (a DMA statement may be empty)
#define V8_STEP(S, d0_, d1_)                                                                    \
  do {                                                                                          \
    V8_SB();                                                                                    \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S##w0, S##x0, acc[0][0], 0, 0, 0);      \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S##w1, S##x0, acc[0][1], 0, 0, 0);      \
    V8_SB();                                                                                    \
    d0_;                                                                                        \
    V8_SB();                                                                                    \
    acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S##w2, S##x0, acc[0][2], 0, 0, 0);      \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S##w0, S##x1, acc[1][0], 0, 0, 0);      \
    V8_SB();                                                                                    \
    d1_;                                                                                        \
    V8_SB();                                                                                    \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S##w1, S##x1, acc[1][1], 0, 0, 0);      \
    acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S##w2, S##x1, acc[1][2], 0, 0, 0);      \
    V8_SB();                                                                                    \
  } while (0)
    const bool dma_on = (ABL != 2) && !PROD;
    if constexpr (PROD) {
      if (wave >= G::NW) {
        // ---- producer wave pw stages the rows the consumer waves 2 pw and 2 pw + 1 would have staged: 8 A + 6 W pieces / tile
        const int pw = wave_u - G::NW;
        int pa_off[2][NA], pb_off[2][NB], pa_lds[2][NA], pb_lds[2][NB];
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
          const int cw = 2 * pw + s_;
#pragma unroll
          for (int i = 0; i < NA; ++i) {
            const int r = cw * (NA * 8) + i * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            pa_lds[s_][i] = (cw * (NA * 8) + i * 8) * 128;
            const int rl = row0 + r < p.M ? r : p.M - 1 - row0;
            pa_off[s_][i] = (rl * (int)p.lda + c * 8) * 2;
          }
#pragma unroll
          for (int i = 0; i < NB; ++i) {
            const int r = cw * (NB * 8) + i * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            pb_lds[s_][i] = (cw * (NB * 8) + i * 8) * 128;
            pb_off[s_][i] = (r * (int)p.ldw + c * 8) * 2;
          }
        }
        auto pdma_a = [&](int kt_, int slot) {
#pragma unroll
          for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
            for (int i = 0; i < NA; ++i)
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(smem + slot * A_BYTES + pa_lds[s_][i]), 16, pa_off[s_][i],
                                                       kt_ * (BK * 2), 0, 0);
        };
        auto pdma_w = [&](int kt_, int slot) {
#pragma unroll
          for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
            for (int i = 0; i < NB; ++i)
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(smem + W_BASE + slot * B_BYTES + pb_lds[s_][i]), 16,
                                                       pb_off[s_][i], kt_ * (BK * 2), 0, 0);
        };
        pdma_a(0, 0);
        pdma_w(0, 0);
        if (nt > 1) {
          pdma_a(1, 1);
          asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        V8_SB();
        __builtin_amdgcn_s_barrier();
        V8_SB();
        int psa = 0, psw = 0;
        for (int t = 0; t < nt; ++t) {
          const int sa1 = psa == 2 ? 0 : psa + 1, sa2 = sa1 == 2 ? 0 : sa1 + 1, sw1 = psw ^ 1;
          if (t + 1 < nt) pdma_w(t + 1, sw1);   // slot of tile t-1: free since barrier t-1
          if (t + 2 < nt) {
            pdma_a(t + 2, sa2);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // all but A(t+2): A(t+1) and W(t+1) have landed
          } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          V8_SB();
          __builtin_amdgcn_s_barrier();
          V8_SB();
          psa = sa1;
          psw = sw1;
        }
        __syncthreads();
        return;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NA; ++i) dma_a(i, 0, 0);
#pragma unroll
      for (int i = 0; i < NB; ++i) dma_w(i, 0, 0);
      if (nt > 1) {
#pragma unroll
        for (int i = 0; i < NA; ++i) dma_a(i, 1, 1);
      }
      ln_prologue();
      if (nt > 1) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    V8_SB();
    __builtin_amdgcn_s_barrier();
    V8_SB();
    int sa = 0, sw = 0;  // LDS slots of the current tile
    if constexpr (MF) {
      bf16x8 wa0, wa1, wa2, wa3, wa4, wa5, wb0, wb1, wb2, wb3, wb4, wb5, xa0, xa1, xb0, xb1;
#define M16_X(D_, sa_, i_, ks_) D_ = *reinterpret_cast<const bf16x8*>(smem + (sa_) * A_BYTES + (xo16[i_] ^ ((ks_) << 6)))
#define M16_W(D_, sw_, j_, ks_) D_ = *reinterpret_cast<const bf16x8*>(smem + W_BASE - A_BYTES + (sw_) * B_BYTES + (wo16[j_] ^ ((ks_) << 6)))
#define M16_MMA(W_, X_, i_, j_) acc16[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W_, X_, acc16[i_][j_], 0, 0, 0)
      // one phase: token blocks I0, I0 + 1 (fragments X0, X1) against the six column blocks of set W; DMA statements behind MFMAs 4 and 8
#define M16_PHASE(W, X0, X1, I0, d0_, d1_)                                                                  \
  do {                                                                                                      \
    V8_SB();                                                                                                \
    M16_MMA(W##0, X0, I0, 0); M16_MMA(W##1, X0, I0, 1); M16_MMA(W##2, X0, I0, 2); M16_MMA(W##3, X0, I0, 3);  \
    V8_SB();                                                                                                \
    d0_;                                                                                                    \
    V8_SB();                                                                                                \
    M16_MMA(W##4, X0, I0, 4); M16_MMA(W##5, X0, I0, 5); M16_MMA(W##0, X1, I0 + 1, 0); M16_MMA(W##1, X1, I0 + 1, 1); \
    V8_SB();                                                                                                \
    d1_;                                                                                                    \
    V8_SB();                                                                                                \
    M16_MMA(W##2, X1, I0 + 1, 2); M16_MMA(W##3, X1, I0 + 1, 3); M16_MMA(W##4, X1, I0 + 1, 4); M16_MMA(W##5, X1, I0 + 1, 5); \
    V8_SB();                                                                                                \
  } while (0)
      // the 4-wave (128-row) geometry stages SIX W pieces per wave and K-tile: three DMA statements per phase, behind MFMAs 3, 6 and 9 —
      // W(t+1) pieces 2..5 first, A(t+2) last, so the counted wait in front of the barrier is still "everything but the four newest"
#define M16_PHASE3(W, X0, X1, I0, d0_, d1_, d2_)                                                            \
  do {                                                                                                      \
    V8_SB();                                                                                                \
    M16_MMA(W##0, X0, I0, 0); M16_MMA(W##1, X0, I0, 1); M16_MMA(W##2, X0, I0, 2);                            \
    V8_SB();                                                                                                \
    d0_;                                                                                                    \
    V8_SB();                                                                                                \
    M16_MMA(W##3, X0, I0, 3); M16_MMA(W##4, X0, I0, 4); M16_MMA(W##5, X0, I0, 5);                            \
    V8_SB();                                                                                                \
    d1_;                                                                                                    \
    V8_SB();                                                                                                \
    M16_MMA(W##0, X1, I0 + 1, 0); M16_MMA(W##1, X1, I0 + 1, 1); M16_MMA(W##2, X1, I0 + 1, 2);                \
    V8_SB();                                                                                                \
    d2_;                                                                                                    \
    V8_SB();                                                                                                \
    M16_MMA(W##3, X1, I0 + 1, 3); M16_MMA(W##4, X1, I0 + 1, 4); M16_MMA(W##5, X1, I0 + 1, 5);                \
    V8_SB();                                                                                                \
  } while (0)
      M16_W(wa0, 0, 0, 0); M16_W(wa1, 0, 1, 0); M16_W(wa2, 0, 2, 0); M16_W(wa3, 0, 3, 0); M16_W(wa4, 0, 4, 0); M16_W(wa5, 0, 5, 0);
      M16_X(xa0, 0, 0, 0); M16_X(xa1, 0, 1, 0);
      if (dma_on && nt > 1) { dma_w(0, 1, 1); dma_w(1, 1, 1); }
      for (int t = 0; t < nt; ++t) {
        const int sa1 = sa == 2 ? 0 : sa + 1, sa2 = sa1 == 2 ? 0 : sa1 + 1, sw1 = sw ^ 1;
        const bool n1 = dma_on && (t + 1 < nt), n2 = dma_on && (t + 2 < nt);
        // phase 0: k-step 0, token blocks 0 1 | ahead: X pair B of k-step 0, first half of W set B (k-step 1)
        M16_X(xb0, sa, 2, 0); M16_X(xb1, sa, 3, 0); M16_W(wb0, sw, 0, 1); M16_W(wb1, sw, 1, 1); M16_W(wb2, sw, 2, 1);
        if constexpr (NB == 6) M16_PHASE3(wa, xa0, xa1, 0, if (n1) dma_w(2, t + 1, sw1), if (n1) dma_w(3, t + 1, sw1), if (n1) dma_w(4, t + 1, sw1));
        else M16_PHASE(wa, xa0, xa1, 0, if (n1) dma_w(2, t + 1, sw1), if (n2) dma_a(0, t + 2, sa2));
        // phase 1: k-step 0, token blocks 2 3 | ahead: X pair A of k-step 1, second half of W set B
        M16_X(xa0, sa, 0, 1); M16_X(xa1, sa, 1, 1); M16_W(wb3, sw, 3, 1); M16_W(wb4, sw, 4, 1); M16_W(wb5, sw, 5, 1);
        if constexpr (NB == 6) M16_PHASE3(wa, xb0, xb1, 2, if (n1) dma_w(5 < NB ? 5 : 0, t + 1, sw1), if (n2) dma_a(0, t + 2, sa2), if (n2) dma_a(1, t + 2, sa2));
        else M16_PHASE(wa, xb0, xb1, 2, if (n2) dma_a(1, t + 2, sa2), if (n2) dma_a(2, t + 2, sa2));
        // phase 2: k-step 1, token blocks 0 1 | ahead: X pair B of k-step 1
        M16_X(xb0, sa, 2, 1); M16_X(xb1, sa, 3, 1);
        if constexpr (NB == 6) M16_PHASE3(wb, xa0, xa1, 0, if (n2) dma_a(2, t + 2, sa2), if (n2) dma_a(3, t + 2, sa2), (void)0);
        else M16_PHASE(wb, xa0, xa1, 0, if (n2) dma_a(3, t + 2, sa2), (void)0);
        if (n2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        V8_SB();
        __builtin_amdgcn_s_barrier();  // tile t+1 has landed everywhere; nobody reads the slots of tile t any more
        V8_SB();
        // phase 3: k-step 1, token blocks 2 3 | ahead: W set A and X pair A of the NEXT tile's k-step 0
        M16_W(wa0, sw1, 0, 0); M16_W(wa1, sw1, 1, 0); M16_W(wa2, sw1, 2, 0); M16_W(wa3, sw1, 3, 0); M16_W(wa4, sw1, 4, 0); M16_W(wa5, sw1, 5, 0);
        M16_X(xa0, sa1, 0, 0); M16_X(xa1, sa1, 1, 0);
        M16_PHASE(wb, xb0, xb1, 2, if (n2) dma_w(0, t + 2, sw), if (n2) dma_w(1, t + 2, sw));
        sa = sa1;
        sw = sw1;
      }
#undef M16_X
#undef M16_W
#undef M16_MMA
#undef M16_PHASE
#undef M16_PHASE3
    } else {
    V8_READ(f0, 0, 0, 0);
    if (dma_on && nt > 1) { dma_w(0, 1, 1); dma_w(1, 1, 1); }
    for (int t = 0; t < nt; ++t) {
      const int sa1 = sa == 2 ? 0 : sa + 1, sa2 = sa1 == 2 ? 0 : sa1 + 1, sw1 = sw ^ 1;
      const bool n1 = dma_on && (t + 1 < nt), n2 = dma_on && (t + 2 < nt);
      V8_READ(f1, sa, sw, 1);
      V8_STEP(f0, if (n1) dma_w(2, t + 1, sw1), if (n2) dma_a(0, t + 2, sa2));
      V8_READ(f0, sa, sw, 2);
      V8_STEP(f1, if (n2) dma_a(1, t + 2, sa2), if (n2) dma_a(2, t + 2, sa2));
      V8_READ(f1, sa, sw, 3);
      V8_STEP(f0, if (n2) dma_a(3, t + 2, sa2), (void)0);
      if (n2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      V8_SB();
      __builtin_amdgcn_s_barrier();  // tile t+1 has landed everywhere; nobody reads the slots of tile t any more
      V8_SB();
      V8_READ(f0, sa1, sw1, 0);
      V8_STEP(f1, if (n2) dma_w(0, t + 2, sw), if (n2) dma_w(1, t + 2, sw));
      sa = sa1;
      sw = sw1;
    }
    }
    __syncthreads();
#undef V8_READ
#undef V8_SB
#undef V8_STEP
  } else if constexpr (BASE == 6) {
    // fragment sets F0 / F1 as named registers (arrays selected by a runtime index would go to scratch)
    bf16x8 f0x0, f0x1, f0w0, f0w1, f0w2, f1x0, f1x1, f1w0, f1w1, f1w2;
#define V6_READ(S, buf_, ks_)                                                                   \
  do {                                                                                          \
    const char* b_ = smem + (buf_) * STAGE_BYTES;                                               \
    S##x0 = *reinterpret_cast<const bf16x8*>(b_ + (xo[0] ^ ((ks_) << 5)));                      \
    S##x1 = *reinterpret_cast<const bf16x8*>(b_ + (xo[1] ^ ((ks_) << 5)));                      \
    S##w0 = *reinterpret_cast<const bf16x8*>(b_ + (wo[0] ^ ((ks_) << 5)));                      \
    S##w1 = *reinterpret_cast<const bf16x8*>(b_ + (wo[1] ^ ((ks_) << 5)));                      \
    S##w2 = *reinterpret_cast<const bf16x8*>(b_ + (wo[2] ^ ((ks_) << 5)));                      \
  } while (0)
#define V6_SB() __builtin_amdgcn_sched_barrier(0)
#define V6_MMA2(S, i_, j0_, j1_)                                                                           \
  do {                                                                                                     \
    acc[i_][j0_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S##w##j0_, S##x##i_, acc[i_][j0_], 0, 0, 0);    \
    acc[i_][j1_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S##w##j1_, S##x##i_, acc[i_][j1_], 0, 0, 0);    \
  } while (0)
    // one k-step: 6 MFMAs of set S in three pairs, the DMA pieces [lo_, hi_) of tile dkt_ slotted in after the pairs
#define V6_STEP(S, do_dma_, lo_, hi_, dkt_, dbuf_)                                              \
  do {                                                                                          \
    V6_SB();                                                                                    \
    V6_MMA2(S, 0, 0, 1);                                                                        \
    V6_SB();                                                                                    \
    if (do_dma_) GEMM_DMA_RANGE((lo_), ((lo_) + (hi_) + 1) / 2, (dkt_), (dbuf_));               \
    V6_SB();                                                                                    \
    acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S##w2, S##x0, acc[0][2], 0, 0, 0);      \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S##w0, S##x1, acc[1][0], 0, 0, 0);      \
    V6_SB();                                                                                    \
    if (do_dma_) GEMM_DMA_RANGE(((lo_) + (hi_) + 1) / 2, (hi_), (dkt_), (dbuf_));               \
    V6_SB();                                                                                    \
    V6_MMA2(S, 1, 1, 2);                                                                        \
    V6_SB();                                                                                    \
  } while (0)
    // piece split over the four steps that follow a tile barrier: D(ks=3 of tile t), A, B, C(ks=0..2 of tile t+1)
    constexpr int P1 = (NP * 3 + 9) / 10, P2 = (NP * 6 + 9) / 10, P3 = (NP * 9 + 9) / 10;  // 7: 3,5,7,7   10: 3,6,9,10
    // tile kt in buffer cur_: F0 holds (or is receiving) its ks=0 fragments; pieces [0,P1) of tile kt+1 are in flight
#define V6_TILE(kt_, cur_)                                                                      \
  do {                                                                                          \
    const bool n1_ = (ABL != 2) && ((kt_) + 1 < nt), n2_ = (ABL != 2) && ((kt_) + 2 < nt);     \
    V6_READ(f1, cur_, 1);                                                                       \
    V6_STEP(f0, n1_, P1, P2, (kt_) + 1, (cur_) ^ 1);                                            \
    V6_READ(f0, cur_, 2);                                                                       \
    V6_STEP(f1, n1_, P2, P3, (kt_) + 1, (cur_) ^ 1);                                            \
    V6_READ(f1, cur_, 3);                                                                       \
    V6_STEP(f0, n1_, P3, NP, (kt_) + 1, (cur_) ^ 1);                                            \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                 \
    V6_SB();                                                                                    \
    __builtin_amdgcn_s_barrier(); /* tile kt+1 has landed everywhere; nobody reads buffer cur_ any more */ \
    V6_SB();                                                                                    \
    V6_READ(f0, (cur_) ^ 1, 0);                                                                 \
    V6_STEP(f1, n2_, 0, P1, (kt_) + 2, (cur_));                                                 \
  } while (0)
    if (ABL != 2 && nt > 1) GEMM_DMA_RANGE(0, P1, 1, 1);
    V6_READ(f0, 0, 0);
    int kt = 0;
    for (; kt + 1 < nt; kt += 2) {
      V6_TILE(kt, 0);
      V6_TILE(kt + 1, 1);
    }
    if (kt < nt) V6_TILE(kt, 0);
    __syncthreads();
#undef V6_READ
#undef V6_SB
#undef V6_MMA2
#undef V6_STEP
#undef V6_TILE
  } else if constexpr (BASE == 9) {
    // ---- the few-tile schedule (128-row geometry, 16x16x32; no more tiles than CUs, so a CU holds ONE workgroup = one wave per SIMD and
    // nothing else covers a DMA round trip): ring of NS9 = THREE 40-KiB stages, tile t + 2 requested during tile t — both operands two tiles
    // ahead — with its ten pieces per wave placed one behind every fourth MFMA (a piece holds its wave ~60-180 cycles at issue; there
    // that time runs beside the matrix pipe), counted wait ("all but my newest ten pieces" = tile t + 1 has landed), raw s_barrier.
    // Schedule 3 here has one tile in flight and schedule 8 one W tile: measured 0.72 us per K-tile for 0.37 us of MFMA whatever the
    // tile count (60 / 114 / 228 tiles: fc2 58.7 / 60.3 / 62.4 us) — one L2 round trip per K-tile.
    static_assert(BASE != 9 || (MF && BM_ == 128 && !KS), "schedule 9 exists for the 128-row geometry on 16x16x32");
    auto tile9 = [&](auto dma_tag, int cur, int kt2, int buf2) {
      constexpr bool dma = decltype(dma_tag)::value;
      const char* sb_ = smem + cur * STAGE_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 xf[4], wf[6];
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(sb_ + (xo16[i] ^ (ks << 6)));
#pragma unroll
        for (int j = 0; j < 6; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(sb_ + (wo16[j] ^ (ks << 6)));
        if constexpr (dma) GEMM_DMA_RANGE(ks * (NP / 2), (ks + 1) * (NP / 2), kt2, buf2);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 6; ++j)
            acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc16[i][j], 0, 0, 0);
        if constexpr (dma) {   // (all ten pieces in the first k-step instead: fc2 0.86 vs 0.81 of schedule 3's time — the issue stalls then bunch up)
          __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
#pragma unroll
          for (int q = 0; q < NP / 2; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x010, 1, 0); }
          __builtin_amdgcn_sched_group_barrier(0x008, 24 - 4 * (NP / 2), 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    int cur = 0, kt = 0;
    for (; kt + NS9 - 1 < nt; ++kt) {   // steady state: tile kt + NS9 - 1 goes into the stage tile kt - 1 was read from
      tile9(std::true_type{}, cur, kt + NS9 - 1, cur == 0 ? NS9 - 1 : cur - 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS9 - 2) * NP) : "memory");   // everything but my newest NS9 - 2 tiles: tile kt + 1 has landed
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      cur = cur == NS9 - 1 ? 0 : cur + 1;
    }
    for (; kt + 1 < nt; ++kt) {   // the last NS9 - 1 tiles request nothing
      tile9(std::false_type{}, cur, 0, 0);
      wait9(nt - 2 - kt);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      cur = cur == NS9 - 1 ? 0 : cur + 1;
    }
    tile9(std::false_type{}, cur, 0, 0);
    __syncthreads();
  } else {
    for (int kt = 0; kt < last; ++kt) {
      const int cur = kt & 1;
      // safe: every wave finished reading buf[cur^1] before the barrier that ended step kt-1
      if (ABL != 2) GEMM_DMA_RANGE(0, NP, kt + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      compute(cur);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces have landed ...
      __syncthreads();                                   // ... and so have everybody else's
    }
    compute(last & 1);
    __syncthreads();
  }
#undef GEMM_DMA_RANGE

  if constexpr (KS) {
    // ---- hand-off of the first K half (cdna_hip_programming.md section 6 G16: write-through payload, drained, then the flag)
    typedef uint32_t ks_u32x4 __attribute__((ext_vector_type(4)));
    constexpr int KS_WAVE_BYTES = 24 * 64 * 16;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<char*>(p.sk_ws) + (int64_t)(tile * 4 + wv) * KS_WAVE_BYTES), 0,
                                                      KS_WAVE_BYTES, 0x00020000);
    int* flag = p.sk_flags + tile * 4 + wv;
    if (ks_half == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          ks_u32x4 v;
          v.x = __float_as_uint(acc16[i][j][0]); v.y = __float_as_uint(acc16[i][j][1]);
          v.z = __float_as_uint(acc16[i][j][2]); v.w = __float_as_uint(acc16[i][j][3]);
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16, (i * 6 + j) * 1024, 16 /* sc1 */);
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1) __builtin_amdgcn_s_sleep(4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ks_u32x4 v[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (i * 6 + j) * 1024, 16 /* sc1 */);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        acc16[i][j][0] += __uint_as_float(v[j].x); acc16[i][j][1] += __uint_as_float(v[j].y);
        acc16[i][j][2] += __uint_as_float(v[j].z); acc16[i][j][3] += __uint_as_float(v[j].w);
      }
    }
    if (lane == 0) __hip_atomic_store(flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the loads above have returned: their values were consumed)
  }

  if constexpr (ABL == 1 && !MF) {
    // lab: keep the accumulators alive without an epilogue; lane 0 of a never-true condition writes them
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }

  // ---- epilogue: acc (+bias, act, gate) -> bf16 -> per-wave LDS image [64 tokens][96 cols] -> 16-byte HBM stores
  // (Round 6: fetching them at the START of the K loop instead — 48 registers held across it, vmcnt(16) at the prologue — was built and
  //  measured: proj 0.132 vs 0.126-0.129 ms, 97.25 vs 96.9 ms per step.  As in round 2's two-K-tiles-early form, WHEN the rows are
  //  requested is not what the gate + residual epilogue costs.)
  // Residual rows for the store phase below are fetched NOW (12 x 16 B per lane, rows clamped instead of branched) so
  // their HBM latency hides under the accumulator -> LDS transposition; a load-wait-store chain per 16 bytes would
  // serialise 12 HBM round trips per wave (CDNA4 vmcnt also counts the stores).
  constexpr bool GATED = EPI == EPI_GATE_RES || EPI == EPI_GATE_RES_STATS;
  constexpr bool LN = EPI == EPI_LN_BIAS || EPI == EPI_LN_GELU;
  if constexpr (!RES_EARLY) fetch_res();
  char* st = smem + wave * OUT_WAVE_BYTES;
  const int ncol0 = col0 + wn * 96;
  if constexpr (EPI == EPI_GATE_RES_STATS) {
    // The residual rows go INTO the wave's image first (coalesced 16-byte units), so that the accumulator pass below sees
    // x_new = res + gate (acc + bias) in the layout where a lane owns one token row: it rounds x_new as the store phase of the
    // plain epilogue does, takes the LayerNorm partial of its 48 columns on the way, and the store phase becomes a copy.
#pragma unroll
    for (int it = 0; it < 12; ++it) {
      const int q = lane + 64 * it;
      const int m_local = q / 12, c = q - m_local * 12;
      *reinterpret_cast<uint4*>(st + m_local * OUT_ROW_BYTES + c * 16) = rres[it];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  // the LayerNorm partials of what the wave's image holds (x_new, a lane pair per token row, the 48 columns of a lane in the order of the
  // 32x32x16 accumulator pass): the form the folded store phase below uses, and what the 16x16x32 accumulator pass cannot do in registers
  auto stats_from_image = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m_local = i * 32 + l31;
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
      const int grow = row0 + wm * 64 + m_local;
      if (hi == 0 && grow < p.M) p.stats_out[(int64_t)(ncol0 / LN_BLOCK) * p.stats_ld + grow] = blk;
    }
  };
  if constexpr (MF) {
    // ---- 16x16x32 accumulators: token 16 i + l15 of the wave's 64 rows, columns 16 j + 4 lq .. + 3
    if constexpr (LN) {
      if constexpr (BM_ != 256) {   // (128-row tiles stage the LayerNorm image after the K loop: no room beside the staging slots)
        ln_stage_tile(ln_lds, p.ln_stats, p.ln_ld, p.ln_nb, p.ln_eps, p.M, row0, BM_, p.cs, p.cv, col0, BN, tid, G::NT);
        __syncthreads();
      }
      float mu[4], rs[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float2 ms = ln_lds[wm * 64 + i * 16 + l15];
        mu[i] = ms.x;
        rs[i] = ms.y;
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float4* cp = reinterpret_cast<const float4*>(ln_lds + BM_ + wn * 96 + j * 16 + 4 * lq);   // (cs, cv) of four columns
        const float4 a = cp[0], b = cp[1];
        const float c_s[4] = {a.x, a.z, b.x, b.z}, c_v[4] = {a.y, a.w, b.y, b.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float nmu = -mu[i], r_ = rs[i];
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
          *reinterpret_cast<uint2*>(st + (i * 16 + l15) * OUT_ROW_BYTES + (j * 16 + 4 * lq) * 2) = o;
        }
      }
    } else {
      uint2 bb16[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) bb16[j] = make_uint2(0, 0);
      if (p.bias != nullptr) {
#pragma unroll
        for (int j = 0; j < 6; ++j) bb16[j] = *reinterpret_cast<const uint2*>(p.bias + ncol0 + j * 16 + 4 * lq);
      }
      // the gate rows of all four token blocks in ONE burst (24 x 8 B per lane): loaded per token block inside the loop below they were
      // four dependent global round trips per wave (vmcnt(5) .. vmcnt(0) four times over), ~1 us each with one wave per SIMD
      uint2 gg[GATED ? 4 : 1][6];
      if (GATED) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 6; ++j) gg[i][j] = make_uint2(0x3f803f80u, 0x3f803f80u);  // bf16 1.0 pairs
        if (p.gate != nullptr) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            int grow = row0 + wm * 64 + i * 16 + l15;
            grow = grow < p.M ? grow : p.M - 1;
            const int sample = grow / p.rows_per_sample;
            const bf16_t* gate_row = p.gate + (int64_t)sample * p.gate_stride + ncol0 + 4 * lq;
            if (p.seg_split > 0 && grow - sample * p.rows_per_sample < p.seg_split) gate_row += p.gate_alt;
#pragma unroll
            for (int j = 0; j < 6; ++j) gg[i][j] = *reinterpret_cast<const uint2*>(gate_row + j * 16);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m_local = i * 16 + l15;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int n_local = j * 16 + 4 * lq;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc16[i][j][r];
          v[0] += bflo(bb16[j].x); v[1] += bfhi(bb16[j].x); v[2] += bflo(bb16[j].y); v[3] += bfhi(bb16[j].y);
          if (EPI == EPI_BIAS_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_tanh(v[r]);
          }
          if (GATED) {
            const uint2 g_ = gg[GATED ? i : 0][j];
            v[0] *= bflo(g_.x); v[1] *= bfhi(g_.x); v[2] *= bflo(g_.y); v[3] *= bfhi(g_.y);
          }
          uint2 o;
          o.x = pack2bf(v[0], v[1]);
          o.y = pack2bf(v[2], v[3]);
          if constexpr (EPI == EPI_GATE_RES_STATS) {
            // + residual (same two roundings as the plain epilogue: bf16(u), then bf16(u + res)); the statistics follow from the image
            const uint2 rr = *reinterpret_cast<const uint2*>(st + m_local * OUT_ROW_BYTES + n_local * 2);
            o.x = pack2bf(bflo(o.x) + bflo(rr.x), bfhi(o.x) + bfhi(rr.x));
            o.y = pack2bf(bflo(o.y) + bflo(rr.y), bfhi(o.y) + bfhi(rr.y));
          }
          *reinterpret_cast<uint2*>(st + m_local * OUT_ROW_BYTES + n_local * 2) = o;
        }
      }
      if constexpr (EPI == EPI_GATE_RES_STATS) stats_from_image();
    }
  }
  if constexpr (LN && !MF) {
    // AdaLN folded into the GEMM (vsys_internal.h GemmParams): out = rstd_m (acc - mu_m cs[n]) + cv[n]
    if constexpr (BM_ != 256) {
      ln_stage_tile(ln_lds, p.ln_stats, p.ln_ld, p.ln_nb, p.ln_eps, p.M, row0, BM_, p.cs, p.cv, col0, BN, tid, G::NT);
      __syncthreads();
    }
    float mu[2], rs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float2 ms = ln_lds[wm * 64 + i * 32 + l31];
      mu[i] = ms.x;
      rs[i] = ms.y;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float4 c_s[4], c_v[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {   // (cs, cv) pairs of four consecutive columns
        const float4* cp = reinterpret_cast<const float4*>(ln_lds + BM_ + wn * 96 + j * 32 + 8 * g + 4 * hi);
        const float4 a = cp[0], b = cp[1];
        c_s[g] = make_float4(a.x, a.z, b.x, b.z);
        c_v[g] = make_float4(a.y, a.w, b.y, b.w);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m_local = i * 32 + l31;
        const float nmu = -mu[i], r_ = rs[i];
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
  // bias for this lane's 12 column groups (4 consecutive columns each), loaded back-to-back under one uniform branch
  uint2 bb[3][4];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) bb[j][g] = make_uint2(0, 0);
  if (p.bias != nullptr && !LN && !MF) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) bb[j][g] = *reinterpret_cast<const uint2*>(p.bias + ncol0 + j * 32 + 8 * g + 4 * hi);
  }
#pragma unroll
  for (int i = 0; i < ((LN || MF) ? 0 : 2); ++i) {
    const int m_local = i * 32 + l31;
    uint2 gg[3][4];
    LnAcc lacc;
    if (GATED) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) gg[j][g] = make_uint2(0x3f803f80u, 0x3f803f80u);  // bf16 1.0 pairs
      if (p.gate != nullptr) {
        int grow = row0 + wm * 64 + m_local;
        grow = grow < p.M ? grow : p.M - 1;
        const int sample = grow / p.rows_per_sample;
        const bf16_t* gate_row = p.gate + (int64_t)sample * p.gate_stride + ncol0 + 4 * hi;
        if (p.seg_split > 0 && grow - sample * p.rows_per_sample < p.seg_split) gate_row += p.gate_alt;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) gg[j][g] = *reinterpret_cast<const uint2*>(gate_row + j * 32 + 8 * g);
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
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
        if (GATED) {
          v[0] *= bflo(gg[j][g].x); v[1] *= bfhi(gg[j][g].x); v[2] *= bflo(gg[j][g].y); v[3] *= bfhi(gg[j][g].y);
        }
        uint2 o;
        o.x = pack2bf(v[0], v[1]);
        o.y = pack2bf(v[2], v[3]);
        if constexpr (EPI == EPI_GATE_RES_STATS) {
          // + residual (same two roundings as the plain epilogue: bf16(u), then bf16(u + res)), statistics of what is stored
          const uint2 rr = *reinterpret_cast<const uint2*>(st + m_local * OUT_ROW_BYTES + n_local * 2);
          o.x = pack2bf(bflo(o.x) + bflo(rr.x), bfhi(o.x) + bfhi(rr.x));
          o.y = pack2bf(bflo(o.y) + bflo(rr.y), bfhi(o.y) + bfhi(rr.y));
          if (j == 0 && g == 0) lacc.init(bflo(o.x));
          lacc.add(bflo(o.x)); lacc.add(bfhi(o.x)); lacc.add(bflo(o.y)); lacc.add(bfhi(o.y));
        }
        *reinterpret_cast<uint2*>(st + m_local * OUT_ROW_BYTES + n_local * 2) = o;
      }
    }
    if constexpr (EPI == EPI_GATE_RES_STATS) {
      // the other 48 columns of this row's 96-column block live in lane l31 + 32
      const float2 mine = lacc.finish(48.f);
      const float2 other = make_float2(__shfl_xor(mine.x, 32, 64), __shfl_xor(mine.y, 32, 64));
      const float2 blk = ln_merge_equal(mine, other, 48.f);
      const int grow = row0 + wm * 64 + m_local;
      if (hi == 0 && grow < p.M) p.stats_out[(int64_t)(ncol0 / LN_BLOCK) * p.stats_ld + grow] = blk;
    }
  }
  __syncthreads();
  // Store phase.  Full tiles (every launch on the denoise path: M is a multiple of 256) take the branch-free form:
  // all 12 image reads are issued before the first store, so the stores do not each wait on their own LDS round trip.
  const bool full = row0 + BM_ <= p.M;
  if constexpr (EPI == EPI_GATE_RES) {
    if (p.add1 != nullptr || p.stats_out != nullptr) {
      // Folded PAB broadcasts (GemmParams::add1 / add2) and / or LayerNorm partials next to a PAB slab copy: the general form of the
      // store phase.  The extra operands are fetched in one burst while the image is read; the stored chunk goes back into the
      // image so that the statistics pass below finds x_new where a lane owns a token row (the layout of EPI_GATE_RES_STATS).
      uint4 val[12], r1[12], r2[12];
#pragma unroll
      for (int it = 0; it < 12; ++it) {
        const int q = lane + 64 * it;
        const int m_local = q / 12, c = q - m_local * 12;
        int grow = row0 + wm * 64 + m_local;
        grow = grow < p.M ? grow : p.M - 1;
        const int64_t off = (int64_t)grow * p.ldr + ncol0 + c * 8;
        r1[it] = p.add1 != nullptr ? *reinterpret_cast<const uint4*>(p.add1 + off) : make_uint4(0, 0, 0, 0);
        r2[it] = p.add2 != nullptr ? *reinterpret_cast<const uint4*>(p.add2 + off) : make_uint4(0, 0, 0, 0);
        val[it] = *reinterpret_cast<const uint4*>(st + m_local * OUT_ROW_BYTES + c * 16);
      }
#pragma unroll
      for (int it = 0; it < 12; ++it) {
        const int q = lane + 64 * it;
        const int m_local = q / 12, c = q - m_local * 12;
        const int grow = row0 + wm * 64 + m_local;
        const int gcol = ncol0 + c * 8;
        const bool ok = grow < p.M;
        uint4 v = val[it];
        if (p.aux != nullptr && ok) *reinterpret_cast<uint4*>(p.aux + (int64_t)grow * p.ldaux + gcol) = v;
        float a[8], b[8];
        if (p.res != nullptr) {
          unpack8(v, a);
          unpack8(rres[it], b);
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] += b[e];
          v = pack8(a);
        }
        if (p.add1 != nullptr) {
          unpack8(v, a);
          unpack8(r1[it], b);
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] += b[e];
          v = pack8(a);
        }
        if (p.add2 != nullptr) {
          unpack8(v, a);
          unpack8(r2[it], b);
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] += b[e];
          v = pack8(a);
        }
        if (ok) *reinterpret_cast<uint4*>(p.out + (int64_t)grow * p.ldo + gcol) = v;
        if (p.stats_out != nullptr) *reinterpret_cast<uint4*>(st + m_local * OUT_ROW_BYTES + c * 16) = v;
      }
      if (p.stats_out != nullptr) stats_from_image();
      return;
    }
  }
  if (full) {
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
      const int64_t grow = row0 + wm * 64 + m_local;
      const int gcol = ncol0 + c * 8;
      uint4 v = val[it];
      if (EPI == EPI_GATE_RES) {
        if (p.aux != nullptr) *reinterpret_cast<uint4*>(p.aux + grow * p.ldaux + gcol) = v;
        if (p.res != nullptr) {
          float a[8], b[8];
          unpack8(v, a);
          unpack8(rres[it], b);
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] += b[e];
          v = pack8(a);
        }
      }
      if constexpr (ABL == 4) {  // lab: epilogue arithmetic and LDS transposition, no HBM store
        asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
      } else if constexpr (ABL == 5) {  // lab: streaming (nt) stores
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
        u32x4 vv = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(vv, reinterpret_cast<u32x4*>(p.out + grow * p.ldo + gcol));
      } else {
        *reinterpret_cast<uint4*>(p.out + grow * p.ldo + gcol) = v;
      }
    }
    return;
  }
#pragma unroll
  for (int it = 0; it < 12; ++it) {
    const int q = lane + 64 * it;
    const int m_local = q / 12, c = q - m_local * 12;
    const int grow = row0 + wm * 64 + m_local;
    uint4 val = *reinterpret_cast<const uint4*>(st + m_local * OUT_ROW_BYTES + c * 16);
    const int gcol = ncol0 + c * 8;
    const bool ok = grow < p.M;
    if (EPI == EPI_GATE_RES) {
      if (p.aux != nullptr && ok) *reinterpret_cast<uint4*>(p.aux + (int64_t)grow * p.ldaux + gcol) = val;
      if (p.res != nullptr) {
        float a[8], b[8];
        unpack8(val, a);
        unpack8(rres[it], b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += b[e];
        val = pack8(a);
      }
    }
    if (ok) *reinterpret_cast<uint4*>(p.out + (int64_t)grow * p.ldo + gcol) = val;
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------
// Small/odd-shape linear: one wave per output element.  Used for the per-step vectors (t_embedder, fps_embedder,
// t_block: M = 2) and the once-per-video text projection (y_embedder: M = B*L), never for the token GEMMs.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_small_kernel(const bf16_t* __restrict__ x, int64_t ldx,
                                                           const bf16_t* __restrict__ w, int64_t ldw,
                                                           const bf16_t* __restrict__ bias, bf16_t* __restrict__ out,
                                                           int64_t ldo, int M, int N, int K, int act_in, int act_out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t idx = (int64_t)blockIdx.x * 4 + wave;
  if (idx >= (int64_t)M * N) return;
  const int m = (int)(idx / N), n = (int)(idx - (int64_t)m * N);
  const bf16_t* xr = x + (int64_t)m * ldx;
  const bf16_t* wr = w + (int64_t)n * ldw;
  float s = 0.f;
  for (int k = lane * 8; k < K; k += 64 * 8) {
    uint4 xv = *reinterpret_cast<const uint4*>(xr + k);
    uint4 wv = *reinterpret_cast<const uint4*>(wr + k);
    float a[8], b[8];
    unpack8(xv, a);
    unpack8(wv, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float xe = a[e];
      if (act_in == ACT_SILU) xe = bf2f(f2bf(silu(xe)));  // nn.SiLU on a bf16 tensor rounds before the Linear
      s += xe * b[e];
    }
  }
  s = wave_sum(s);
  if (lane == 0) {
    if (bias != nullptr) s += bf2f(bias[n]);
    if (act_out == ACT_SILU) s = silu(s);
    else if (act_out == ACT_GELU_TANH) s = gelu_tanh(s);
    out[(int64_t)m * ldo + n] = f2bf(s);
  }
}

}  // namespace

// Variant id understood by set_gemm_variant / vsys_tune_gemm_variant:  PIPE (+ 100 for the 128-row, two-workgroups-per-CU
// geometry).  0 = the shipped default.
// (process-wide A/B selector of the measurement tools and the schedule-equivalence tests; read once per launch)
static std::atomic<int> g_gemm_variant_a{0};
// tile raster of the measurement tools: ids 2GGPP of vsys_tune_gemm_variant set column-group width GG and panel-chunk height PP
// for every later launch (20600 = the default); any raster computes the same tiles, so the bits do not depend on it
static std::atomic<int> g_gemm_raster_a{600};
// 0 = shape dispatch (default).  The shipped library only accepts ids whose kernels produce VALID output (they differ in
// schedule / geometry only and are bit-identical); ablation and stamp variants exist in -DVSYS_LAB builds (VSYS_LAB=1 build()).
int set_gemm_variant(int v) {
  if (v >= 20000 && v < 30000) {
    g_gemm_raster_a.store(v - 20000, std::memory_order_relaxed);
    return 0;
  }
  switch (v) {
    case 0: case 3: case 6: case 8: case 9: case 16: case 20: case 24: case 34: case 28: case 30: case 50: case 103: case 106: case 113: case 118: case 119: case 123: break;
#ifdef VSYS_LAB
    case 60: case 70: case 80:   // ping-pong wave groups / persistent grid / stream-K tail (gemm4_bf16.hip): valid, measured, not shipped
    case 18: case 38: case 48: case 31: case 40: case 61: case 62: case 63: case 64: case 71: case 72: case 73: case 74: case 78: case 81: case 82: case 83: case 84: break;
#endif
    default: return VSYS_ERR_ARG;
  }
  g_gemm_variant_a.store(v, std::memory_order_relaxed);
  return 0;
}

template <int PIPE, int BM_, int RASTER = 1, int PROD = 0, int MF = 0, int KS = 0>
static int launch_gemm_t(const GemmParams& p, int epi, hipStream_t stream) {
  using G = Geo<BM_>;
  if (PROD && epi != EPI_BIAS && epi != EPI_BIAS_GELU) return launch_gemm_t<PIPE, BM_, RASTER, 0, MF, KS>(p, epi, stream);  // 208 VGPRs: no third wave
  const int nbm = (p.M + BM_ - 1) / BM_, nbn = p.N / BN;
  const int grid = KS ? ((nbm * nbn + 7) / 8) * 16 : nbm * nbn;   // KS: two workgroups per tile, pairs laid out XCD by XCD (gemm_kernel)
  const bool lnl = epi == EPI_LN_BIAS || epi == EPI_LN_GELU;
  const size_t lds = ((PIPE % 10 == 8) ? 3 * G::A_BYTES + 2 * B_BYTES : (PIPE % 10 == 9 ? 3 : 2) * G::STAGE) + (lnl && BM_ == 256 ? (BM_ + BN) * 8 : 0);
  static std::atomic<unsigned long long> attr_seen{0};   // per device (and per template instance)
  for (DeviceOnce once(attr_seen); once.todo(); once.done()) {
    const int lds = (int)(((PIPE % 10 == 8) ? 3 * G::A_BYTES + 2 * B_BYTES : (PIPE % 10 == 9 ? 3 : 2) * G::STAGE) + (BM_ == 256 ? (BM_ + BN) * 8 : 0));
    (void)hipFuncSetAttribute((const void*)gemm_kernel<EPI_BIAS, PIPE, BM_, RASTER, PROD, MF, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_kernel<EPI_BIAS_GELU, PIPE, BM_, RASTER, PROD, MF, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_kernel<EPI_GATE_RES, PIPE, BM_, RASTER, 0, MF, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if constexpr (!PROD) {
      (void)hipFuncSetAttribute((const void*)gemm_kernel<EPI_GATE_RES_STATS, PIPE, BM_, RASTER, 0, MF, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void*)gemm_kernel<EPI_LN_BIAS, PIPE, BM_, RASTER, 0, MF, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void*)gemm_kernel<EPI_LN_GELU, PIPE, BM_, RASTER, 0, MF, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
  }
  constexpr int NTH = G::NT + PROD * 256;
  switch (epi) {
    case EPI_BIAS: hipLaunchKernelGGL((gemm_kernel<EPI_BIAS, PIPE, BM_, RASTER, PROD, MF, KS>), dim3(grid), dim3(NTH), lds, stream, p); break;
    case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm_kernel<EPI_BIAS_GELU, PIPE, BM_, RASTER, PROD, MF, KS>), dim3(grid), dim3(NTH), lds, stream, p); break;
    case EPI_GATE_RES: hipLaunchKernelGGL((gemm_kernel<EPI_GATE_RES, PIPE, BM_, RASTER, 0, MF, KS>), dim3(grid), dim3(G::NT), lds, stream, p); break;
    case EPI_GATE_RES_STATS: hipLaunchKernelGGL((gemm_kernel<EPI_GATE_RES_STATS, PIPE, BM_, RASTER, 0, MF, KS>), dim3(grid), dim3(G::NT), lds, stream, p); break;
    case EPI_LN_BIAS: hipLaunchKernelGGL((gemm_kernel<EPI_LN_BIAS, PIPE, BM_, RASTER, 0, MF, KS>), dim3(grid), dim3(G::NT), lds, stream, p); break;
    case EPI_LN_GELU: hipLaunchKernelGGL((gemm_kernel<EPI_LN_GELU, PIPE, BM_, RASTER, 0, MF, KS>), dim3(grid), dim3(G::NT), lds, stream, p); break;
    default: return VSYS_ERR_ARG;
  }
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

static bool mf16_default() {
  static const bool on = [] { const char* e = getenv("VSYS_GEMM_MF16"); return !(e && e[0] == '0'); }();
  return on;
}

// split-K workspace of one stream (launches on a stream are serialised; two streams — the overlap mode of DSP — never share one):
// partial sums of up to ``tiles`` tiles x 4 waves x 24 x 64 lanes float4, and one flag per (tile, wave), zeroed once: the consuming
// workgroup lowers a flag after use, so a replayed launch program (same arguments every step) finds them down.  Allocated on first use,
// never resized (sized for one tile per CU), freed with the process.
namespace {
struct SplitKWs { float* ws = nullptr; int* flags = nullptr; int tiles = 0; };
std::mutex g_splitk_mutex;
std::map<hipStream_t, SplitKWs> g_splitk_ws;
}  // namespace
static bool splitk_workspace(hipStream_t stream, int tiles, GemmParams& p) {
  std::lock_guard<std::mutex> lock(g_splitk_mutex);
  SplitKWs& w = g_splitk_ws[stream];
  if (w.ws == nullptr) {
    const int cap = cu_count_this_device();
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return false;   // no allocation under capture
    if (hipMalloc(&w.ws, (size_t)cap * 4 * 24 * 64 * sizeof(float4)) != hipSuccess) { w.ws = nullptr; return false; }
    if (hipMalloc(&w.flags, (size_t)cap * 4 * sizeof(int)) != hipSuccess || hipMemset(w.flags, 0, (size_t)cap * 4 * sizeof(int)) != hipSuccess) {
      (void)hipFree(w.ws);
      w.ws = nullptr;
      return false;
    }
    (void)hipDeviceSynchronize();   // (the memset is ordered against the first launch on any stream)
    w.tiles = cap;
  }
  if (tiles > w.tiles) return false;
  p.sk_ws = w.ws;
  p.sk_flags = w.flags;
  return true;
}
// the 128-row geometry (few tiles): 16x16x32 by default as well.  Two workgroups per tile (KS) is OPT-IN (VSYS_GEMM_SPLITK=1 or id 123): at
// 4864 rows the hand-off (98 KB of fp32 sums per tile through L2 + the flags) costs more than half a K loop of 18 tiles saves — proj
// 31.1 vs 23.4 us, cross-q 27.6 vs 19.3, fc2 (K = 4608) 63.8 vs 62.9: profiles/r06_small_m_split_k.txt
static int launch_rows128(const GemmParams& p_, int epi, hipStream_t stream, bool force_split = false) {
  if (!mf16_default() && !force_split) return launch_gemm_t<3, 128>(p_, epi, stream);
  static const bool splitk_on = [] { const char* e = getenv("VSYS_GEMM_SPLITK"); return e && e[0] == '1'; }();
  const int tiles = ((p_.M + 127) / 128) * (p_.N / BN);
  if ((splitk_on || force_split) && tiles >= 64 && tiles <= cu_count_this_device() && p_.K / BK >= 8) {
    GemmParams p = p_;
    if (splitk_workspace(stream, tiles, p)) return launch_gemm_t<3, 128, 1, 0, 1, 1>(p, epi, stream);
  }
  // (No more tiles than CUs: a CU holds one workgroup whatever its LDS footprint, so the 128-row tile could run on schedule 8 — id 118,
  // VSYS_GEMM_ROWS128_S8=1.  Measured at 4864 rows: proj 22.6 vs 22.7 us, cross-q 19.0 vs 19.3, fc2 60.7 vs 62.8, one rank of eight
  // 19.6 vs 19.5 ms per step — the few-tile launches are not bound by the K loop's schedule; not dispatched.)
  static const bool s8_on = [] { const char* e = getenv("VSYS_GEMM_ROWS128_S8"); return e && e[0] == '1'; }();
  if (s8_on && tiles <= cu_count_this_device() && p_.K / BK >= 3) return launch_gemm_t<8, 128, 1, 0, 1>(p_, epi, stream);
  // What does bind them is the L2 -> LDS round trip: one CU = one workgroup = one wave per SIMD, and schedule 3 has ONE tile in flight
  // (schedule 8: one W tile): 0.72 us per K-tile for 0.37 us of MFMA whatever the tile count.  Schedule 9 (ring of three stages, both
  // operands two tiles ahead, id 119) at 4864 rows: fc2 62.3 -> 51.5 us (973-1035 TFLOP/s), proj 23.0 -> 20.0, cross-q 19.3 -> 17.0;
  // with MORE tiles than CUs two workgroups of schedule 3 share a CU and cover each other (qkv 43.4 vs 46.6 us): not dispatched there.
  static const bool ring_off = [] { const char* e = getenv("VSYS_GEMM_ROWS128_RING"); return e && e[0] == '0'; }();
  if (!ring_off && tiles <= cu_count_this_device() && p_.K / BK >= 3) return launch_gemm_t<9, 128, 1, 0, 1>(p_, epi, stream);
  return launch_gemm_t<3, 128, 1, 0, 1>(p_, epi, stream);
}
// schedule 8 on the 256-row geometry: the 16x16x32 form by default (same bits), VSYS_GEMM_MF16=0 / variant 8 the 32x32x16 form
static int launch_sched8(const GemmParams& p, int epi, hipStream_t stream) {
  return mf16_default() ? launch_gemm_t<8, 256, 1, 0, 1>(p, epi, stream) : launch_gemm_t<8, 256>(p, epi, stream);
}

int launch_gemm(const GemmParams& p_, int epi, hipStream_t stream) {
  if (p_.M <= 0) return 0;
  GemmParams p = p_;
  {
    const int ras = g_gemm_raster_a.load(std::memory_order_relaxed);
    p.raster_gw = ras / 100;
    p.raster_ph = ras % 100;
  }
  if (p.N % BN != 0 || p.K % BK != 0 || p.N <= 0 || p.K <= 0) return VSYS_ERR_SHAPE;
  if ((p.lda % 8) || (p.ldw % 8) || (p.ldo % 8) || (p.res && (p.ldr % 8)) || (p.aux && (p.ldaux % 8))) return VSYS_ERR_ALIGN;
  // tile-relative operand offsets are 32-bit (buffer addressing): checked for EVERY epilogue, in front of the per-epilogue dispatch
  if (p.lda * 512 + (int64_t)p.K * 2 >= 0x7fffffff || p.ldw * 384 + (int64_t)p.K * 2 >= 0x7fffffff) return VSYS_ERR_SHAPE;
  if ((epi == EPI_GATE_RES || epi == EPI_GATE_RES_STATS) && p.gate && p.rows_per_sample <= 0) return VSYS_ERR_SHAPE;
  const bool ln = epi == EPI_LN_BIAS || epi == EPI_LN_GELU;
  if (ln && (!p.cs || !p.cv || !p.ln_stats || p.ln_nb < 1 || p.ln_nb > 12 || p.ln_nb * LN_BLOCK != p.K || p.ln_ld < p.M)) return VSYS_ERR_SHAPE;
  if (epi == EPI_GATE_RES_STATS && (!p.stats_out || p.stats_ld < p.M || p.aux)) return VSYS_ERR_ARG;
  if (epi != EPI_GATE_RES && (p.add1 || p.add2)) return VSYS_ERR_ARG;
  if (epi == EPI_GATE_RES && (p.add1 || p.add2 || p.stats_out)) {   // folded broadcasts / statistics beside a slab copy: gemm_kernel only
    if ((p.add2 && !p.add1) || ((p.add1 || p.add2) && (!p.res || (p.ldr % 8))) || (p.stats_out && p.stats_ld < p.M)) return VSYS_ERR_ARG;
    if ((int64_t)((p.M + 255) / 256) * (p.N / BN) < 400) return launch_rows128(p, epi, stream, g_gemm_variant_a.load(std::memory_order_relaxed) == 123);
    return g_gemm_variant_a.load(std::memory_order_relaxed) == 8 ? launch_gemm_t<8, 256>(p, epi, stream) : launch_sched8(p, epi, stream);
  }
  // the statistics-emitting epilogue lives in gemm_kernel only (lab / forced variants of other kernel families fall back to it)
  if (epi == EPI_GATE_RES_STATS) {
    if ((int64_t)((p.M + 255) / 256) * (p.N / BN) < 400) return launch_rows128(p, epi, stream, g_gemm_variant_a.load(std::memory_order_relaxed) == 123);
    if (g_gemm_variant_a.load(std::memory_order_relaxed) == 24) return launch_gemm2(p, epi, 2, stream);   // (A/B id: the two-workgroup 16x16x32 kernel)
    return g_gemm_variant_a.load(std::memory_order_relaxed) == 8 ? launch_gemm_t<8, 256>(p, epi, stream) : launch_sched8(p, epi, stream);
  }
  if (ln) {   // same shape dispatch as the store-only epilogues below
    if ((int64_t)((p.M + 255) / 256) * (p.N / BN) < 400) return launch_rows128(p, epi, stream, g_gemm_variant_a.load(std::memory_order_relaxed) == 123);
    if (p.K <= 1536 && p.N >= 2304 && g_gemm_variant_a.load(std::memory_order_relaxed) != 8) return launch_gemm2(p, epi, mf16_default() ? 2 : 0, stream);
    return g_gemm_variant_a.load(std::memory_order_relaxed) == 8 ? launch_gemm_t<8, 256>(p, epi, stream) : launch_sched8(p, epi, stream);
  }
  const int g_gemm_variant = g_gemm_variant_a.load(std::memory_order_relaxed);
  switch (g_gemm_variant == 50 ? 0 : g_gemm_variant) {
    case 6: return launch_gemm_t<6, 256>(p, epi, stream);
#ifdef VSYS_LAB
    case 18: return launch_gemm_t<18, 256>(p, epi, stream);
    case 48: return launch_gemm_t<48, 256>(p, epi, stream);
    case 38: return launch_gemm_t<38, 256>(p, epi, stream);   // every tile streams the A rows of tile 0 (A L2-resident: fabric traffic = W only)
    case 40: return launch_gemm3(p, epi, stream);  // 5-slot ring, fragments always one k-step ahead (gemm3_bf16.hip)
    case 31: return (epi == EPI_BIAS && p.N % 384 == 0) ? launch_gemm2_stamp(p, stream) : VSYS_ERR_ARG;  // lab: cycle stamps
    case 61: case 62: case 63: case 64: return epi == EPI_BIAS ? launch_gemm4_lab(p, g_gemm_variant - 60, 0, stream) : VSYS_ERR_ARG;
    case 71: case 72: case 73: case 74: case 78: return epi == EPI_BIAS ? launch_gemm4_lab(p, g_gemm_variant - 70, 1, stream) : VSYS_ERR_ARG;
    case 60: case 70: case 80:  // ping-pong wave groups (gemm4_bf16.hip): 60 = one tile per workgroup, 70 = persistent (one
      // workgroup per CU), 80 = persistent + stream-K split of the partial last round (fp32 summation order differs: not bit-identical)
      if (!gemm4_supports(p, epi)) return launch_gemm_t<8, 256>(p, epi, stream);
      return launch_gemm4(p, epi, (g_gemm_variant - 60) / 10, stream);
    case 81: case 82: case 83: case 84:   // stream-K ablations (wrong results): 81 no DUMP, 82 no gather, 83 neither, 84 no SK role at all
      return gemm4_supports(p, epi) ? launch_gemm4(p, epi, 2 + g_gemm_variant - 80, stream) : VSYS_ERR_ARG;
#endif
    case 103: return launch_gemm_t<3, 128>(p, epi, stream);
    case 113: return launch_gemm_t<3, 128, 1, 0, 1>(p, epi, stream);   // ... on v_mfma_f32_16x16x32_bf16 (one workgroup per tile)
    case 118: return launch_gemm_t<8, 128, 1, 0, 1>(p, epi, stream);   // the 128-row geometry on schedule 8 (16x16x32)
    case 119: return launch_gemm_t<9, 128, 1, 0, 1>(p, epi, stream);   // the 128-row geometry on the three-stage ring (16x16x32)
    case 123: return launch_rows128(p, epi, stream, true);               // ... two workgroups per tile where the tile count allows (split K)
    case 106: return launch_gemm_t<6, 128>(p, epi, stream);   // the 128-row geometry on schedule 6 (DMA pieces interleaved with the MFMA pairs)
    case 20: return launch_gemm2(p, epi, 0, stream);  // 4-wave workgroups, two per CU (gemm2_bf16.hip)
    case 24:   // the same on v_mfma_f32_16x16x32_bf16 (same bits); a gated launch with a slab copy falls back to schedule 8
      if (epi == EPI_GATE_RES && p.aux) return launch_sched8(p, epi, stream);
      return launch_gemm2(p, epi, 2, stream);
    case 30: return p.N % 384 == 0 ? launch_gemm2(p, epi, 1, stream) : launch_gemm_t<8, 256>(p, epi, stream);  // 256 x 384 tile
    case 34: return (p.N % 384 == 0 && !(epi == EPI_GATE_RES && p.aux)) ? launch_gemm2(p, epi, 3, stream) : launch_sched8(p, epi, stream);  // ... on 16x16x32
    case 8: return launch_gemm_t<8, 256>(p, epi, stream);  // force schedule 8 (32x32x16) for every shape
    case 16: return launch_gemm_t<8, 256, 1, 0, 1>(p, epi, stream);  // schedule 8 on v_mfma_f32_16x16x32_bf16 for every shape
    case 28: return launch_gemm_t<8, 256, 1, 1>(p, epi, stream);  // schedule 8 + four producer waves (store-only epilogues)
    case 3: return launch_gemm_t<3, 256>(p, epi, stream);
    case 9: return launch_gemm_t<8, 256, 0>(p, epi, stream);  // schedule 8, plain row-major tile order
    default:
      // Shape dispatch (config-2 A/B, tools/kernel_bench.py --variants 8,20): with a short K loop and a store-only epilogue
      // the two-workgroups-per-CU geometry wins (qkv 1152->3456 -3 %, fc1 1152->4608 -7 %); with K = 4608 or the
      // gate+residual epilogue the 8-wave kernel is 10-18 % faster.  Both produce identical bits.
      // Few tiles (one rank of an 8-way DSP run has M = 4864: 114 tiles of 256 rows for the N = 1152 GEMMs on 256 CUs): the
      // 128-row geometry (two 4-wave workgroups per CU, 512 slots) fills the chip; measured at M = 4864 against schedule 8:
      // qkv -7 %, proj -15 %, fc2 -16 %, fc1 -2 % (tools/kernel_bench.py --rows 4864 --variants 8,20,103).
      if ((int64_t)((p.M + 255) / 256) * (p.N / BN) < 400) return launch_rows128(p, epi, stream, g_gemm_variant_a.load(std::memory_order_relaxed) == 123);
      // Round 6: the two-workgroup kernel on v_mfma_f32_16x16x32_bf16 (launch_gemm2 wide = 2; same bits, the matrix pipe ~12 % cheaper per
      // flop under the power cap): qkv 0.297 -> 0.276 ms, fc1 0.425 -> 0.404, and it now also beats schedule 8 on the N = K = 1152
      // store-only shape (cross-attention q: 0.111 -> 0.095 ms) — profiles/r06_kernel_bench_mf16.txt.  VSYS_GEMM_MF16=0: the 32x32x16 kernels.
      if (epi != EPI_GATE_RES && p.K <= 1536 && mf16_default()) return launch_gemm2(p, epi, 2, stream);
      if (epi != EPI_GATE_RES && p.K <= 1536 && p.N >= 2304) return launch_gemm2(p, epi, 0, stream);
      if (g_gemm_variant == 50 && p.K >= 2304 && p.N % 384 == 0) return launch_gemm2(p, epi, 1, stream);  // lab: long-K on the wide tile
      return launch_sched8(p, epi, stream);
  }
}

int launch_linear_small(const bf16_t* x, int64_t ldx, const bf16_t* w, int64_t ldw, const bf16_t* bias, bf16_t* out,
                        int64_t ldo, int M, int N, int K, int act_in, int act_out, hipStream_t stream) {
  if (M <= 0 || N <= 0) return 0;
  if (K % 8 != 0 || (ldx % 8) || (ldw % 8)) return VSYS_ERR_ALIGN;
  const int64_t total = (int64_t)M * N;
  const int64_t grid = (total + 3) / 4;
  if (grid > 0x7fffffff) return VSYS_ERR_SHAPE;
  hipLaunchKernelGGL(linear_small_kernel, dim3((unsigned)grid), dim3(256), 0, stream, x, ldx, w, ldw, bias, out, ldo, M, N,
                     K, act_in, act_out);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

}  // namespace vsys
