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
#include "common.h"
#include "vsys_internal.h"

namespace vsys {

namespace {

constexpr int BM = 256, BN = 192, BK = 64, NTHREADS = 512;
constexpr int A_BYTES = BM * BK * 2;  // 32768
constexpr int B_BYTES = BN * BK * 2;  // 24576
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;  // 57344
constexpr int OUT_ROW_BYTES = 96 * 2 + 16;      // per-wave epilogue image row (padded)
constexpr int OUT_WAVE_BYTES = 64 * OUT_ROW_BYTES;  // 13312
static_assert(8 * OUT_WAVE_BYTES <= 2 * STAGE_BYTES, "epilogue image must fit in the staging buffers");

__device__ __forceinline__ int swz(int row, int chunk) { return (row << 7) + ((chunk ^ ((row >> 1) & 7)) << 4); }

// PIPE = 1: one staging register set (loads of tile t+1 fly during the MFMAs of tile t);
// PIPE = 2: two sets (tile t+2 in flight while tile t+1 waits in registers) — the default.
template <int EPI, int PIPE>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_256x192_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;

  const int nbn = p.N / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int bm = tile / nbn, bn = tile - bm * nbn;
  const int row0 = bm * BM, col0 = bn * BN;

  // ---- staging assignment.
  // PIPE 1/2 (register staged): chunk q = tid + 512*i -> (row = q>>3, logical 16-byte chunk = q&7), written to the
  //   swizzled LDS slot by ds_write_b128.
  // PIPE 3 (LDS-DMA): global_load_lds writes lane l of a wave to  M0_base + 16*l  (lane-linear), so each wave
  //   instruction fills 8 whole 128-byte rows and the swizzle moves to the SOURCE: the lane that lands in physical
  //   slot s of row r fetches logical chunk s ^ ((r>>1)&7)  (same involution as the fragment reads).
  const bf16_t* aptr[4];
  const bf16_t* bptr[3];
  int a_lds[4], b_lds[3];
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r, c;
    if constexpr (PIPE >= 3) {
      r = wave_u * 32 + i * 8 + (lane >> 3);
      c = (lane & 7) ^ ((r >> 1) & 7);
      a_lds[i] = (wave_u * 32 + i * 8) * 128;  // wave-uniform row-block base
    } else {
      const int q = tid + NTHREADS * i;
      r = q >> 3;
      c = q & 7;
      a_lds[i] = swz(r, c);
    }
    int gr = row0 + r;
    gr = gr < p.M ? gr : p.M - 1;
    aptr[i] = p.A + (int64_t)gr * p.lda + c * 8;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int r, c;
    if constexpr (PIPE >= 3) {
      r = wave_u * 24 + i * 8 + (lane >> 3);
      c = (lane & 7) ^ ((r >> 1) & 7);
      b_lds[i] = A_BYTES + (wave_u * 24 + i * 8) * 128;
    } else {
      const int q = tid + NTHREADS * i;
      r = q >> 3;
      c = q & 7;
      b_lds[i] = A_BYTES + swz(r, c);
    }
    bptr[i] = p.W + (int64_t)(col0 + r) * p.ldw + c * 8;
  }

  // Two staging register sets (named scalars + macros on purpose: arrays captured by a lambda and written under a
  // condition get parked in scratch by hipcc).  Set X holds tile t+1 while the loads of tile t+2 fly into set Y, so a
  // load has TWO tile-times of MFMA work to land before its ds_write needs it (1 block/CU: nothing else hides it).
  uint4 ra0_0, ra1_0, ra2_0, ra3_0, rb0_0, rb1_0, rb2_0;
  uint4 ra0_1, ra1_1, ra2_1, ra3_1, rb0_1, rb1_1, rb2_1;
#define GEMM_GLOAD(S, kt_)                                                 \
  do {                                                                     \
    const int ko_ = (kt_) * BK;                                            \
    ra0_##S = *reinterpret_cast<const uint4*>(aptr[0] + ko_);              \
    ra1_##S = *reinterpret_cast<const uint4*>(aptr[1] + ko_);              \
    ra2_##S = *reinterpret_cast<const uint4*>(aptr[2] + ko_);              \
    ra3_##S = *reinterpret_cast<const uint4*>(aptr[3] + ko_);              \
    rb0_##S = *reinterpret_cast<const uint4*>(bptr[0] + ko_);              \
    rb1_##S = *reinterpret_cast<const uint4*>(bptr[1] + ko_);              \
    rb2_##S = *reinterpret_cast<const uint4*>(bptr[2] + ko_);              \
  } while (0)
#define GEMM_LSTORE(S, buf_)                                               \
  do {                                                                     \
    char* base_ = smem + (buf_) * STAGE_BYTES;                             \
    *reinterpret_cast<uint4*>(base_ + a_lds[0]) = ra0_##S;                 \
    *reinterpret_cast<uint4*>(base_ + a_lds[1]) = ra1_##S;                 \
    *reinterpret_cast<uint4*>(base_ + a_lds[2]) = ra2_##S;                 \
    *reinterpret_cast<uint4*>(base_ + a_lds[3]) = ra3_##S;                 \
    *reinterpret_cast<uint4*>(base_ + b_lds[0]) = rb0_##S;                 \
    *reinterpret_cast<uint4*>(base_ + b_lds[1]) = rb1_##S;                 \
    *reinterpret_cast<uint4*>(base_ + b_lds[2]) = rb2_##S;                 \
  } while (0)

  // ---- fragment read offsets (per lane): X rows (tokens) for i=0,1; W rows (out cols) for j=0..2
  int xrow[2], wrow[3];
#pragma unroll
  for (int i = 0; i < 2; ++i) xrow[i] = wm * 64 + i * 32 + l31;
#pragma unroll
  for (int j = 0; j < 3; ++j) wrow[j] = wn * 96 + j * 32 + l31;

  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int cur) {
    const char* sa = smem + cur * STAGE_BYTES;
    const char* sb = sa + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int chunk = ks * 2 + hi;
      bf16x8 xf[2], wf[3];
#pragma unroll
      for (int i = 0; i < 2; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(sa + swz(xrow[i], chunk));
#pragma unroll
      for (int j = 0; j < 3; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(sb + swz(wrow[j], chunk));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
    }
  };

  const int nt = p.K / BK;
  const int last = nt - 1;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define GEMM_DMA(kt_, buf_)                                                                                         \
  do {                                                                                                              \
    const int ko_ = (kt_) * BK;                                                                                     \
    char* base_ = smem + (buf_) * STAGE_BYTES;                                                                      \
    __builtin_amdgcn_global_load_lds((const void*)(aptr[0] + ko_), (lds_ptr_t)(base_ + a_lds[0]), 16, 0, 0);       \
    __builtin_amdgcn_global_load_lds((const void*)(aptr[1] + ko_), (lds_ptr_t)(base_ + a_lds[1]), 16, 0, 0);       \
    __builtin_amdgcn_global_load_lds((const void*)(aptr[2] + ko_), (lds_ptr_t)(base_ + a_lds[2]), 16, 0, 0);       \
    __builtin_amdgcn_global_load_lds((const void*)(aptr[3] + ko_), (lds_ptr_t)(base_ + a_lds[3]), 16, 0, 0);       \
    __builtin_amdgcn_global_load_lds((const void*)(bptr[0] + ko_), (lds_ptr_t)(base_ + b_lds[0]), 16, 0, 0);       \
    __builtin_amdgcn_global_load_lds((const void*)(bptr[1] + ko_), (lds_ptr_t)(base_ + b_lds[1]), 16, 0, 0);       \
    __builtin_amdgcn_global_load_lds((const void*)(bptr[2] + ko_), (lds_ptr_t)(base_ + b_lds[2]), 16, 0, 0);       \
  } while (0)
  if constexpr (PIPE == 4) {
    // ---- ping-pong: the two waves of a SIMD (wave w and w+4) run the same program one barrier interval apart, so
    // in every interval exactly one of them owns the matrix pipe (12 MFMAs = half a K-tile) while its partner issues
    // the ds_reads (and the LDS-DMA of the next tile) for its own next half-tile.  Raw s_barrier + counted waits only:
    // a __syncthreads() here would drain the in-flight DMA (vmcnt(0)) at every barrier.
    //   per wave, tile t:  R(2t): DMA(t+1), read half 0 | M(2t): 12 MFMA | R(2t+1): read half 1 | M(2t+1): 12 MFMA
    //   buffer of tile t+1 is free at R(2t): all reads of tile t-1 ended (lgkmcnt(0)) before the previous barrier;
    //   DMA(t+1) is waited for (vmcnt(0)) one interval before the first wave reads it: at the end of M(2t+1) by the
    //   leading group, at the end of R(2t+1) by the lagging group (it is the same barrier for both).
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
    bf16x8 xf00, xf01, xf10, xf11, wf00, wf01, wf02, wf10, wf11, wf12;
#define PP_BARRIER()                                   \
  do {                                                 \
    __builtin_amdgcn_sched_barrier(0);                 \
    asm volatile("" ::: "memory");                     \
    __builtin_amdgcn_s_barrier();                      \
    asm volatile("" ::: "memory");                     \
    __builtin_amdgcn_sched_barrier(0);                 \
  } while (0)
#define PP_READ(cur_, half_)                                                                   \
  do {                                                                                         \
    const char* sa_ = smem + (cur_) * STAGE_BYTES;                                             \
    const char* sb_ = sa_ + A_BYTES;                                                           \
    const int c0_ = (2 * (half_)) * 2 + hi, c1_ = (2 * (half_) + 1) * 2 + hi;                  \
    xf00 = *reinterpret_cast<const bf16x8*>(sa_ + swz(xrow[0], c0_));                          \
    xf01 = *reinterpret_cast<const bf16x8*>(sa_ + swz(xrow[1], c0_));                          \
    wf00 = *reinterpret_cast<const bf16x8*>(sb_ + swz(wrow[0], c0_));                          \
    wf01 = *reinterpret_cast<const bf16x8*>(sb_ + swz(wrow[1], c0_));                          \
    wf02 = *reinterpret_cast<const bf16x8*>(sb_ + swz(wrow[2], c0_));                          \
    xf10 = *reinterpret_cast<const bf16x8*>(sa_ + swz(xrow[0], c1_));                          \
    xf11 = *reinterpret_cast<const bf16x8*>(sa_ + swz(xrow[1], c1_));                          \
    wf10 = *reinterpret_cast<const bf16x8*>(sb_ + swz(wrow[0], c1_));                          \
    wf11 = *reinterpret_cast<const bf16x8*>(sb_ + swz(wrow[1], c1_));                          \
    wf12 = *reinterpret_cast<const bf16x8*>(sb_ + swz(wrow[2], c1_));                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
  } while (0)
#define PP_MMA()                                                                               \
  do {                                                                                         \
    __builtin_amdgcn_s_setprio(1);                                                             \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf00, xf00, acc[0][0], 0, 0, 0);       \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf01, xf00, acc[0][1], 0, 0, 0);       \
    acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf02, xf00, acc[0][2], 0, 0, 0);       \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf00, xf01, acc[1][0], 0, 0, 0);       \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf01, xf01, acc[1][1], 0, 0, 0);       \
    acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf02, xf01, acc[1][2], 0, 0, 0);       \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf10, xf10, acc[0][0], 0, 0, 0);       \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf11, xf10, acc[0][1], 0, 0, 0);       \
    acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf12, xf10, acc[0][2], 0, 0, 0);       \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf10, xf11, acc[1][0], 0, 0, 0);       \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf11, xf11, acc[1][1], 0, 0, 0);       \
    acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf12, xf11, acc[1][2], 0, 0, 0);       \
    __builtin_amdgcn_s_setprio(0);                                                             \
  } while (0)
    GEMM_DMA(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_BARRIER();
    if (grp) PP_BARRIER();  // the lagging group starts one interval late
    for (int kt = 0; kt < nt; ++kt) {
      const int cur = kt & 1;
      // R(2t)
      if (kt < last) GEMM_DMA(kt + 1, cur ^ 1);
      PP_READ(cur, 0);
      PP_BARRIER();
      // M(2t)
      PP_MMA();
      PP_BARRIER();
      // R(2t+1)
      PP_READ(cur, 1);
      if (grp) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PP_BARRIER();
      // M(2t+1)
      PP_MMA();
      if (!grp) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PP_BARRIER();
    }
    if (!grp) PP_BARRIER();  // balance the barrier count of the two groups
    __syncthreads();
#undef PP_BARRIER
#undef PP_READ
#undef PP_MMA
  } else if constexpr (PIPE == 5) {
    // ---- LDS-DMA double buffer with the 7 DMA pieces of tile t+1 spread between the MFMA groups of tile t (2,2,2,1
    // per k-step) instead of issued back-to-back at the top of the tile: a global_load_lds costs the issuing wave
    // ~60-180 cycles, and seven of them in a row right after the barrier leave the matrix pipe idle on every SIMD.
#define GEMM_DMA_A(i_, kt_, buf_)                                                                                  \
  __builtin_amdgcn_global_load_lds((const void*)(aptr[i_] + (kt_) * BK), (lds_ptr_t)(smem + (buf_) * STAGE_BYTES + a_lds[i_]), 16, 0, 0)
#define GEMM_DMA_B(i_, kt_, buf_)                                                                                  \
  __builtin_amdgcn_global_load_lds((const void*)(bptr[i_] + (kt_) * BK), (lds_ptr_t)(smem + (buf_) * STAGE_BYTES + b_lds[i_]), 16, 0, 0)
#define GEMM_KSTEP(ks_, sa_, sb_)                                                                                  \
  do {                                                                                                             \
    const int chunk_ = (ks_) * 2 + hi;                                                                             \
    bf16x8 x0_ = *reinterpret_cast<const bf16x8*>((sa_) + swz(xrow[0], chunk_));                                   \
    bf16x8 x1_ = *reinterpret_cast<const bf16x8*>((sa_) + swz(xrow[1], chunk_));                                   \
    bf16x8 w0_ = *reinterpret_cast<const bf16x8*>((sb_) + swz(wrow[0], chunk_));                                   \
    bf16x8 w1_ = *reinterpret_cast<const bf16x8*>((sb_) + swz(wrow[1], chunk_));                                   \
    bf16x8 w2_ = *reinterpret_cast<const bf16x8*>((sb_) + swz(wrow[2], chunk_));                                   \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0_, x0_, acc[0][0], 0, 0, 0);                             \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1_, x0_, acc[0][1], 0, 0, 0);                             \
    acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2_, x0_, acc[0][2], 0, 0, 0);                             \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0_, x1_, acc[1][0], 0, 0, 0);                             \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1_, x1_, acc[1][1], 0, 0, 0);                             \
    acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2_, x1_, acc[1][2], 0, 0, 0);                             \
  } while (0)
    GEMM_DMA(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < last; ++kt) {
      const int cur = kt & 1, nxt = cur ^ 1;
      const char* sa = smem + cur * STAGE_BYTES;
      const char* sb = sa + A_BYTES;
      GEMM_DMA_A(0, kt + 1, nxt);
      GEMM_DMA_A(1, kt + 1, nxt);
      __builtin_amdgcn_sched_barrier(0);
      GEMM_KSTEP(0, sa, sb);
      __builtin_amdgcn_sched_barrier(0);
      GEMM_DMA_A(2, kt + 1, nxt);
      GEMM_DMA_A(3, kt + 1, nxt);
      __builtin_amdgcn_sched_barrier(0);
      GEMM_KSTEP(1, sa, sb);
      __builtin_amdgcn_sched_barrier(0);
      GEMM_DMA_B(0, kt + 1, nxt);
      GEMM_DMA_B(1, kt + 1, nxt);
      __builtin_amdgcn_sched_barrier(0);
      GEMM_KSTEP(2, sa, sb);
      __builtin_amdgcn_sched_barrier(0);
      GEMM_DMA_B(2, kt + 1, nxt);
      __builtin_amdgcn_sched_barrier(0);
      GEMM_KSTEP(3, sa, sb);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    compute(last & 1);
    __syncthreads();
#undef GEMM_DMA_A
#undef GEMM_DMA_B
#undef GEMM_KSTEP
  } else if constexpr (PIPE == 3) {
    GEMM_DMA(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < last; ++kt) {
      const int cur = kt & 1;
      GEMM_DMA(kt + 1, cur ^ 1);  // safe: every wave finished reading buf[cur^1] before the barrier that ended step kt-1
      __builtin_amdgcn_sched_barrier(0);
      compute(cur);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces have landed ...
      __syncthreads();                                   // ... and so have everybody else's
    }
    compute(last & 1);
    __syncthreads();
#undef GEMM_DMA
  } else if constexpr (PIPE == 1) {
    GEMM_GLOAD(0, 0);
    GEMM_LSTORE(0, 0);
    __syncthreads();
    for (int kt = 0; kt < last; ++kt) {  // last tile peeled: no conditional staging inside the loop
      const int cur = kt & 1;
      GEMM_GLOAD(0, kt + 1);
      __builtin_amdgcn_sched_barrier(0);
      compute(cur);
      __builtin_amdgcn_sched_barrier(0);
      GEMM_LSTORE(0, cur ^ 1);
      __syncthreads();
    }
    compute(last & 1);
    __syncthreads();
  } else {
    GEMM_GLOAD(0, 0);
    GEMM_LSTORE(0, 0);
    GEMM_GLOAD(0, last < 1 ? last : 1);
    __syncthreads();
    // Branch-free steady state: tile indices are clamped to the last tile, and the (at most two) stores past the end
    // land in a buffer nobody reads again.
    for (int kt = 0; kt + 1 < nt; kt += 2) {
      // even step: buf0 = tile kt, set 0 = tile kt+1; fetch tile kt+2 into set 1
      GEMM_GLOAD(1, (kt + 2 < last ? kt + 2 : last));
      __builtin_amdgcn_sched_barrier(0);  // keep the global loads ABOVE the MFMAs (hipcc otherwise sinks them to the ds_writes)
      compute(0);
      __builtin_amdgcn_sched_barrier(0);
      GEMM_LSTORE(0, 1);
      __syncthreads();
      // odd step: buf1 = tile kt+1, set 1 = tile kt+2; fetch tile kt+3 into set 0
      GEMM_GLOAD(0, (kt + 3 < last ? kt + 3 : last));
      __builtin_amdgcn_sched_barrier(0);
      compute(1);
      __builtin_amdgcn_sched_barrier(0);
      GEMM_LSTORE(1, 0);
      __syncthreads();
    }
    if (nt & 1) {
      compute(0);
      __syncthreads();
    }
  }
#undef GEMM_GLOAD
#undef GEMM_LSTORE

  // ---- epilogue: acc (+bias, act, gate) -> bf16 -> per-wave LDS image [64 tokens][96 cols] -> 16-byte HBM stores
  // Residual rows for the store phase below are fetched NOW (12 x 16 B per lane, rows clamped instead of branched) so
  // their HBM latency hides under the accumulator -> LDS transposition; a load-wait-store chain per 16 bytes would
  // serialise 12 HBM round trips per wave (CDNA4 vmcnt also counts the stores).
  uint4 rres[12];
  if (EPI == EPI_GATE_RES) {
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
  char* st = smem + wave * OUT_WAVE_BYTES;
  const int ncol0 = col0 + wn * 96;
  // bias for this lane's 12 column groups (4 consecutive columns each), loaded back-to-back under one uniform branch
  uint2 bb[3][4];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) bb[j][g] = make_uint2(0, 0);
  if (p.bias != nullptr) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) bb[j][g] = *reinterpret_cast<const uint2*>(p.bias + ncol0 + j * 32 + 8 * g + 4 * hi);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m_local = i * 32 + l31;
    uint2 gg[3][4];
    if (EPI == EPI_GATE_RES) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) gg[j][g] = make_uint2(0x3f803f80u, 0x3f803f80u);  // bf16 1.0 pairs
      if (p.gate != nullptr) {
        int grow = row0 + wm * 64 + m_local;
        grow = grow < p.M ? grow : p.M - 1;
        const bf16_t* gate_row = p.gate + (int64_t)(grow / p.rows_per_sample) * p.gate_stride + ncol0 + 4 * hi;
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
        if (EPI == EPI_GATE_RES) {
          v[0] *= bflo(gg[j][g].x); v[1] *= bfhi(gg[j][g].x); v[2] *= bflo(gg[j][g].y); v[3] *= bfhi(gg[j][g].y);
        }
        uint2 o;
        o.x = pack2bf(v[0], v[1]);
        o.y = pack2bf(v[2], v[3]);
        *reinterpret_cast<uint2*>(st + m_local * OUT_ROW_BYTES + n_local * 2) = o;
      }
    }
  }
  __syncthreads();
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

static int g_gemm_pipe = 3;
void set_gemm_variant(int v) { g_gemm_pipe = (v >= 1 && v <= 5) ? v : 3; }

template <int PIPE>
static int launch_gemm_pipe(const GemmParams& p, int epi, int grid, size_t lds, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_256x192_kernel<EPI_BIAS, PIPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_256x192_kernel<EPI_BIAS_GELU, PIPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_256x192_kernel<EPI_GATE_RES, PIPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  switch (epi) {
    case EPI_BIAS: hipLaunchKernelGGL((gemm_256x192_kernel<EPI_BIAS, PIPE>), dim3(grid), dim3(NTHREADS), lds, stream, p); break;
    case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm_256x192_kernel<EPI_BIAS_GELU, PIPE>), dim3(grid), dim3(NTHREADS), lds, stream, p); break;
    case EPI_GATE_RES: hipLaunchKernelGGL((gemm_256x192_kernel<EPI_GATE_RES, PIPE>), dim3(grid), dim3(NTHREADS), lds, stream, p); break;
    default: return VSYS_ERR_ARG;
  }
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_gemm(const GemmParams& p, int epi, hipStream_t stream) {
  if (p.M <= 0) return 0;
  if (p.N % BN != 0 || p.K % BK != 0 || p.N <= 0 || p.K <= 0) return VSYS_ERR_SHAPE;
  if ((p.lda % 8) || (p.ldw % 8) || (p.ldo % 8) || (p.res && (p.ldr % 8)) || (p.aux && (p.ldaux % 8))) return VSYS_ERR_ALIGN;
  if (epi == EPI_GATE_RES && p.gate && p.rows_per_sample <= 0) return VSYS_ERR_SHAPE;
  const int nbm = (p.M + BM - 1) / BM, nbn = p.N / BN;
  const int grid = nbm * nbn;
  const size_t lds = 2 * STAGE_BYTES;
  if (g_gemm_pipe == 1) return launch_gemm_pipe<1>(p, epi, grid, lds, stream);
  if (g_gemm_pipe == 3) return launch_gemm_pipe<3>(p, epi, grid, lds, stream);
  if (g_gemm_pipe == 4) return launch_gemm_pipe<4>(p, epi, grid, lds, stream);
  if (g_gemm_pipe == 5) return launch_gemm_pipe<5>(p, epi, grid, lds, stream);
  return launch_gemm_pipe<2>(p, epi, grid, lds, stream);
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
