// bf16 MFMA GEMM, "continuous k-step pipeline" (variant 40 of vsys_tune_gemm_variant).
//
// Built from the cycle accounting of a stage (tools/gemm_stamps.py): in the kernels with one barrier per stage
// and fragment reads AFTER it, ≈25 % of a stage passes with both waves of a SIMD waiting (read burst of all waves right after the
// barrier, counted waits) and nothing on the matrix pipe.  Here the stage barrier guarantees operand data TWO stages ahead (ring
// of five 28 KiB slots, BK = 32, LDS-DMA issued three stages ahead), so a wave may read the fragments of k-step s+1 — also across
// a stage boundary — while the six MFMAs of k-step s run: the loop is a uniform software pipeline of k-steps, the barrier only
// aligns the waves every second k-step and never has a read burst behind it.  Geometry of schedule 8 (8 waves as 4 x 2, wave tile
// 64 x 96 = 2 x 3 accumulators, two fragment register sets) on the 64-byte-row LDS layout of gemm2_bf16.hip.
#include "common.h"
#include "vsys_internal.h"

namespace vsys {
namespace {

constexpr int BM = 256, BN = 192, BK = 32, NS = 5;
constexpr int A_SLOT = BM * BK * 2;          // 16384
constexpr int W_SLOT = BN * BK * 2;          // 12288
constexpr int STAGE = A_SLOT + W_SLOT;       // 28672
constexpr int OUT_ROW_BYTES = 96 * 2 + 16;
constexpr int OUT_WAVE_BYTES = 64 * OUT_ROW_BYTES;  // 13312
constexpr int LDS_BYTES = NS * STAGE;        // 143360
static_assert(8 * OUT_WAVE_BYTES <= LDS_BYTES, "epilogue image must fit in the staging ring");

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm3_kernel(GemmParams p) {
#if __HIP_DEVICE_COMPILE__  // buffer-resource types exist in the device pass only
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  // tile order: W-resident raster of gemm_bf16.hip (column groups of 6 inside 8 row-panel groups)
  const int nbn = p.N / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  int bm, bn;
  if (nbn > 6) {
    const int nbm = (p.M + BM - 1) / BM;
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
  const int row0 = bm * BM, col0 = bn * BN;

  // ---- LDS-DMA assignment (1 KiB pieces = 16 rows x 64 B; chunk swizzle on the source side as in gemm2_bf16.hip):
  // A: 16 pieces, wave w stages rows [32w, 32w+32); W: 12 pieces, waves 0-3 two (rows [32w, 32w+32)), waves 4-7 one
  // (rows [128 + 16(w-4), +16)) — so a wave has 4 or 3 pieces per stage in flight.
  const int dchunk = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;
  const bool two_w = wave_u < 4;
  int a_off[2], b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = wave_u * 32 + i * 16 + (lane >> 2);
    const int rl = row0 + r < p.M ? r : p.M - 1 - row0;  // rows past M re-read the last row (never stored)
    a_off[i] = rl * (int)p.lda * 2 + dchunk;
  }
  const int wrow_base = two_w ? wave_u * 32 : 128 + (wave_u - 4) * 16;
#pragma unroll
  for (int i = 0; i < 2; ++i) b_off[i] = (wrow_base + i * 16 + (lane >> 2)) * (int)p.ldw * 2 + dchunk;
  const int64_t a_bytes = ((int64_t)(p.M - 1 - row0) * p.lda + p.K) * 2, b_bytes = ((int64_t)(p.N - 1 - col0) * p.ldw + p.K) * 2;
  const auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)row0 * p.lda), 0,
                                                        (int)(a_bytes < 0x7fffffff ? a_bytes : 0x7fffffff), 0x00020000);
  const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)col0 * p.ldw), 0,
                                                        (int)(b_bytes < 0x7fffffff ? b_bytes : 0x7fffffff), 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  auto dma_a = [&](int i, int t, int slot) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(smem + slot * STAGE + (wave_u * 32 + i * 16) * 64), 16, a_off[i],
                                             t * (BK * 2), 0, 0);
  };
  auto dma_w = [&](int i, int t, int slot) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(smem + slot * STAGE + A_SLOT + (wrow_base + i * 16) * 64), 16, b_off[i],
                                             t * (BK * 2), 0, 0);
  };
  auto dma_stage = [&](int t, int slot) {  // all pieces of stage t of this wave: 2 A + (2 | 1) W
    dma_a(0, t, slot);
    dma_a(1, t, slot);
    dma_w(0, t, slot);
    if (two_w) dma_w(1, t, slot);
  };

  const int fsw = ((hi ^ ((l31 >> 2) & 3)) << 4);
  const int xo = (wm * 64 + l31) * 64 + fsw;            // + i*2048 for m-block i
  const int wo = A_SLOT + (wn * 96 + l31) * 64 + fsw;   // + j*2048 for n-block j

  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nt = p.K / BK;
  // ---- prologue: stages 0, 1, 2 in flight; stages 0 and 1 must have landed before the loop (invariant: before stage t the
  // data of stages t and t+1 is visible to every wave)
  dma_stage(0, 0);
  if (nt > 1) dma_stage(1, 1);
  if (nt > 2) {
    dma_stage(2, 2);
    if (two_w) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  bf16x8 f0x0, f0x1, f0w0, f0w1, f0w2, f1x0, f1x1, f1w0, f1w1, f1w2;
#define G3_READ(S, slot_, ks_)                                                          \
  do {                                                                                  \
    const char* sb_ = smem + (slot_) * STAGE;                                           \
    S##x0 = *reinterpret_cast<const bf16x8*>(sb_ + (xo ^ ((ks_) << 5)));                \
    S##x1 = *reinterpret_cast<const bf16x8*>(sb_ + (xo ^ ((ks_) << 5)) + 2048);         \
    S##w0 = *reinterpret_cast<const bf16x8*>(sb_ + (wo ^ ((ks_) << 5)));                \
    S##w1 = *reinterpret_cast<const bf16x8*>(sb_ + (wo ^ ((ks_) << 5)) + 2048);         \
    S##w2 = *reinterpret_cast<const bf16x8*>(sb_ + (wo ^ ((ks_) << 5)) + 4096);         \
  } while (0)
#define G3_SB() __builtin_amdgcn_sched_barrier(0)
  // six MFMAs of fragment set S in three pairs; d0_ / d1_ are statements slotted in behind the first / second pair
#define G3_STEP(S, d0_, d1_)                                                                    \
  do {                                                                                          \
    G3_SB();                                                                                    \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S##w0, S##x0, acc[0][0], 0, 0, 0);      \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S##w1, S##x0, acc[0][1], 0, 0, 0);      \
    G3_SB();                                                                                    \
    d0_;                                                                                        \
    G3_SB();                                                                                    \
    acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S##w2, S##x0, acc[0][2], 0, 0, 0);      \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S##w0, S##x1, acc[1][0], 0, 0, 0);      \
    G3_SB();                                                                                    \
    d1_;                                                                                        \
    G3_SB();                                                                                    \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S##w1, S##x1, acc[1][1], 0, 0, 0);      \
    acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S##w2, S##x1, acc[1][2], 0, 0, 0);      \
    G3_SB();                                                                                    \
  } while (0)

  G3_READ(f0, 0, 0);
  int slot = 0;
  for (int t = 0; t < nt; ++t) {
    const int slot1 = slot == NS - 1 ? 0 : slot + 1;
    int slot3 = slot + 3;
    slot3 = slot3 >= NS ? slot3 - NS : slot3;
    const bool n3 = t + 3 < nt;
    // k-step 0 of stage t (set f0) while the k-step-1 fragments are fetched; the LDS-DMA of stage t+3 rides behind the MFMA pairs
    G3_READ(f1, slot, 1);
    G3_STEP(f0, if (n3) { dma_a(0, t + 3, slot3); dma_a(1, t + 3, slot3); }, if (n3) { dma_w(0, t + 3, slot3); if (two_w) dma_w(1, t + 3, slot3); });
    // k-step 1 (set f1) while the first fragments of stage t+1 are fetched: legal before the barrier, stage t+1 has landed
    if (t + 1 < nt) G3_READ(f0, slot1, 0);
    G3_STEP(f1, (void)0, (void)0);
    // stage t+2 must be visible before stage t+1 starts: everything but this wave's stage-t+3 pieces has landed
    if (n3) {
      if (two_w) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    G3_SB();
    __builtin_amdgcn_s_barrier();
    G3_SB();
    slot = slot1;
  }
#undef G3_READ
#undef G3_STEP
#undef G3_SB
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();  // every wave is done with the staging ring: it becomes the epilogue image

  // ---- epilogue: one pass of 64 rows per wave through a wave-private LDS image (whole 192-byte row segments to HBM)
  char* st = smem + wave * OUT_WAVE_BYTES;
  const int ncol0 = col0 + wn * 96;
  const int wrow0 = row0 + wm * 64;
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
  const bool full = row0 + BM <= p.M;
  uint4 rres[12];
  if (EPI == EPI_GATE_RES) {  // residual rows requested before the conversion pass (96 accumulators leave room for them)
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
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m_local = i * 32 + l31;
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
      uint2 gg[4];
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
    if (EPI == EPI_GATE_RES) {
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
#endif
}

}  // namespace

int launch_gemm3(const GemmParams& p, int epi, hipStream_t stream) {
  const int nbm = (p.M + BM - 1) / BM, nbn = p.N / BN;
  const int grid = nbm * nbn;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm3_kernel<EPI_BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm3_kernel<EPI_BIAS_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm3_kernel<EPI_GATE_RES>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  switch (epi) {
    case EPI_BIAS: hipLaunchKernelGGL(gemm3_kernel<EPI_BIAS>, dim3(grid), dim3(512), LDS_BYTES, stream, p); break;
    case EPI_BIAS_GELU: hipLaunchKernelGGL(gemm3_kernel<EPI_BIAS_GELU>, dim3(grid), dim3(512), LDS_BYTES, stream, p); break;
    case EPI_GATE_RES: hipLaunchKernelGGL(gemm3_kernel<EPI_GATE_RES>, dim3(grid), dim3(512), LDS_BYTES, stream, p); break;
    default: return VSYS_ERR_ARG;
  }
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

}  // namespace vsys
