// Implicit-GEMM convolution on bf16 MFMA for the VAE decoders: channels-last activations, a tap is a
// ROW SHIFT of the A operand.
//
// Activations live in HBM as a flat row matrix [rows, Cin] over a spatially zero-padded grid (T, H+2, W+2), so the input of
// tap (kt, kh, kw) for output row r is simply row r + kt*plane + kh*(W+2) + kw (the caller pre-shifts the base pointer by the
// most negative tap).  A conv is then ONE GEMM with K = taps*Cin whose A address jumps by a row delta every Cin/32 k-tiles:
// no im2col buffer, every activation byte is fetched from HBM once per N-tile column and the 27 (or 9) re-reads of a row hit
// L2.  Outputs are produced for every row of the padded grid; border rows hold junk that consumers (GroupNorm, residual,
// upsample) never read.  With taps = 1 the kernel is a plain C = A W^T GEMM with a 128-column tile (the 512 / 256 / 128
// channel counts of the decoders are not multiples of the 192-column tile of gemm_bf16.hip); the single-head d = 512
// mid-block attention of the 2-D decoder runs on it as well (fp32 score output, batch = frame in gridDim.y).
//
// Geometry follows gemm2_bf16.hip: 4 waves (2 x 2), wave tile 128 x 64 = 4 x 2 v_mfma_f32_32x32x16_bf16 accumulators, BK = 32,
// three A slots + two W slots = 64 KiB of LDS, two workgroups per CU; LDS-DMA staging (buffer_load ... lds) with the 16-byte
// chunk swizzle on the source side.
#include "common.h"
#include "vsys_internal.h"

namespace vsys {
namespace {

constexpr int BM = 256, BN = 128, BK = 32;
constexpr int A_SLOT = BM * BK * 2;   // 16384
constexpr int W_SLOT = BN * BK * 2;   // 8192
constexpr int W_BASE = 3 * A_SLOT;    // 49152
constexpr int LDS_BYTES = W_BASE + 2 * W_SLOT;  // 65536
constexpr int OUT_ROW_BYTES = 64 * 2 + 16;
constexpr int OUT_WAVE_BYTES = 64 * OUT_ROW_BYTES;  // 9216 per wave and pass
static_assert(4 * OUT_WAVE_BYTES <= LDS_BYTES, "epilogue image must fit in the staging buffers");

// MF = 1 (round 6): the same walk on v_mfma_f32_16x16x32_bf16, as gemm2_bf16.hip MF (the matrix pipe is ~12 % cheaper per flop in that
// shape under the power cap; same bits): one stage = one k-step of 8 token blocks x 4 column blocks of 16 x 16 in two halves of four token
// blocks; a lane owns token 16 i + lane % 16 and columns 16 j + 4 (lane / 16) .. + 3; LDS chunk swizzle F = (0, 2, 3, 1).
template <int F32OUT, int MF = 0>
__global__ __launch_bounds__(256, 2) void conv_kernel(ConvParams p) {
#if __HIP_DEVICE_COMPILE__  // buffer-resource types exist in the device pass only
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  const int l15 = lane & 15, lq = lane >> 4;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  // N-tiles of one row panel are neighbours in the (XCD-chunked) tile order: the panel's activation rows are fetched once
  const int nbn = p.N / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int bm = tile / nbn, bn = tile - bm * nbn;
  const int row0 = bm * BM, col0 = bn * BN;
  const int64_t zb = blockIdx.y;
  const bf16_t* Ab = p.A + zb * p.batch_a;
  const bf16_t* Wb = p.W + zb * p.batch_w;

  const int dchunk = ((lane & 3) ^ (MF ? ((0x78 >> (2 * ((lane >> 4) & 3))) & 3) : ((lane >> 4) & 3))) * 16;
  int a_off[4], b_off[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave_u * 64 + i * 16 + (lane >> 2);
    const int rl = row0 + r < p.M ? r : p.M - 1 - row0;  // rows past M re-read the last row (never stored)
    a_off[i] = rl * (int)p.lda * 2 + dchunk;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) b_off[i] = (wave_u * 32 + i * 16 + (lane >> 2)) * (int)p.ldw * 2 + dchunk;
  const int64_t a_bytes = ((int64_t)(p.M - 1 - row0) * p.lda + p.max_tap_rows * p.lda + p.cin) * 2;
  const int64_t b_bytes = ((int64_t)(p.N - 1 - col0) * p.ldw + p.K) * 2;
  const auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(Ab + (int64_t)row0 * p.lda), 0,
                                                        (int)(a_bytes < 0x7fffffff ? a_bytes : 0x7fffffff), 0x00020000);
  const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)(Wb + (int64_t)col0 * p.ldw), 0,
                                                        (int)(b_bytes < 0x7fffffff ? b_bytes : 0x7fffffff), 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  // byte offset of k-tile t inside the A panel: tap = t >> cshift selects the row shift, the low bits the channel chunk
  const int cmask = (1 << p.cshift) - 1;
  const int lda2 = (int)p.lda * 2;
  auto a_soff = [&](int t) -> int {
    const int tap = t >> p.cshift, kc = t & cmask;
    const int kt = (tap >= p.taps_hw ? 1 : 0) + (tap >= 2 * p.taps_hw ? 1 : 0);
    const int r = tap - kt * p.taps_hw;
    const int kh = (r >= p.kw ? 1 : 0) + (r >= 2 * p.kw ? 1 : 0);
    const int kwi = r - kh * p.kw;
    return (kt * p.plane_pitch + kh * p.row_pitch + kwi) * lda2 + kc * (BK * 2);
  };
  auto dma_a = [&](int i, int soff, int slot) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(smem + slot * A_SLOT + (wave_u * 64 + i * 16) * 64), 16, a_off[i],
                                             soff, 0, 0);
  };
  auto dma_w = [&](int i, int t, int slot) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(smem + W_BASE + slot * W_SLOT + (wave_u * 32 + i * 16) * 64), 16,
                                             b_off[i], t * (BK * 2), 0, 0);
  };

  const int fsw = MF ? ((lq ^ ((0x78 >> (2 * ((l15 >> 2) & 3))) & 3)) << 4) : ((hi ^ ((l31 >> 2) & 3)) << 4);
  const int frow = MF ? l15 : l31;
  const int xo = (wm * 128 + frow) * 64 + fsw;           // + i*2048 for m-block i (MF: + i*1024 for token block i)
  const int wo = W_BASE + (wn * 64 + frow) * 64 + fsw;   // + j*2048 for n-block j (MF: + j*1024 for column block j)

  f32x16 acc[MF ? 1 : 4][MF ? 1 : 2];
  f32x4 acc16[MF ? 8 : 1][MF ? 4 : 1];
#pragma unroll
  for (int i = 0; i < (MF ? 1 : 4); ++i)
#pragma unroll
    for (int j = 0; j < (MF ? 1 : 2); ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
  for (int i = 0; i < (MF ? 8 : 1); ++i)
#pragma unroll
    for (int j = 0; j < (MF ? 4 : 1); ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.f;

  const int nt = p.K / BK;
  {
    const int s0 = a_soff(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_a(i, s0, 0);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) dma_w(i, 0, 0);
  if (nt > 1) {
    const int s1 = a_soff(1);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_a(i, s1, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  int sa = 0, sw = 0;
  for (int t = 0; t < nt; ++t) {
    const int sa1 = sa == 2 ? 0 : sa + 1, sa2 = sa1 == 2 ? 0 : sa1 + 1, sw1 = sw ^ 1;
    const bool n1 = t + 1 < nt, n2 = t + 2 < nt;
    const int s2 = a_soff(t + 2);
    const char* ab = smem + sa * A_SLOT;
    const char* wb = smem + sw * W_SLOT;
    bf16x8 x0, x1, x2, x3, w0, w1, v0, v1;  // x: A fragments of the current k-step; w / v: W fragments of k-step 0 / 1
    if constexpr (MF) {
      // (w0, w1, v0, v1) = the four column blocks, x0..x3 = token blocks 0..3, reloaded with blocks 4..7 behind their MFMAs
      x0 = *reinterpret_cast<const bf16x8*>(ab + xo);
      w0 = *reinterpret_cast<const bf16x8*>(wb + wo);
      w1 = *reinterpret_cast<const bf16x8*>(wb + wo + 1024);
      v0 = *reinterpret_cast<const bf16x8*>(wb + wo + 2048);
      v1 = *reinterpret_cast<const bf16x8*>(wb + wo + 3072);
      x1 = *reinterpret_cast<const bf16x8*>(ab + xo + 1024);
      x2 = *reinterpret_cast<const bf16x8*>(ab + xo + 2048);
      x3 = *reinterpret_cast<const bf16x8*>(ab + xo + 3072);
      __builtin_amdgcn_sched_barrier(0);
#define CV16_ROW(i_, X_)                                                                              \
  acc16[i_][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, X_, acc16[i_][0], 0, 0, 0);              \
  acc16[i_][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, X_, acc16[i_][1], 0, 0, 0);              \
  acc16[i_][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v0, X_, acc16[i_][2], 0, 0, 0);              \
  acc16[i_][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, X_, acc16[i_][3], 0, 0, 0)
      CV16_ROW(0, x0);
      __builtin_amdgcn_sched_barrier(0);
      x0 = *reinterpret_cast<const bf16x8*>(ab + xo + 4096);
      if (n1) { dma_w(0, t + 1, sw1); dma_w(1, t + 1, sw1); }
      __builtin_amdgcn_sched_barrier(0);
      CV16_ROW(1, x1);
      __builtin_amdgcn_sched_barrier(0);
      x1 = *reinterpret_cast<const bf16x8*>(ab + xo + 5120);
      if (n2) { dma_a(0, s2, sa2); dma_a(1, s2, sa2); }
      __builtin_amdgcn_sched_barrier(0);
      CV16_ROW(2, x2);
      __builtin_amdgcn_sched_barrier(0);
      x2 = *reinterpret_cast<const bf16x8*>(ab + xo + 6144);
      if (n2) dma_a(2, s2, sa2);
      __builtin_amdgcn_sched_barrier(0);
      CV16_ROW(3, x3);
      __builtin_amdgcn_sched_barrier(0);
      x3 = *reinterpret_cast<const bf16x8*>(ab + xo + 7168);
      if (n2) dma_a(3, s2, sa2);
      __builtin_amdgcn_sched_barrier(0);
      CV16_ROW(4, x0);
      CV16_ROW(5, x1);
      CV16_ROW(6, x2);
      CV16_ROW(7, x3);
#undef CV16_ROW
    } else {
    x0 = *reinterpret_cast<const bf16x8*>(ab + xo);
    w0 = *reinterpret_cast<const bf16x8*>(wb + wo);
    w1 = *reinterpret_cast<const bf16x8*>(wb + wo + 2048);
    x1 = *reinterpret_cast<const bf16x8*>(ab + xo + 2048);
    x2 = *reinterpret_cast<const bf16x8*>(ab + xo + 4096);
    x3 = *reinterpret_cast<const bf16x8*>(ab + xo + 6144);
    v0 = *reinterpret_cast<const bf16x8*>(wb + (wo ^ 32));
    v1 = *reinterpret_cast<const bf16x8*>(wb + (wo ^ 32) + 2048);
    __builtin_amdgcn_sched_barrier(0);
#define CV_ROW(i_, X_, W0_, W1_)                                                                     \
  acc[i_][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W0_, X_, acc[i_][0], 0, 0, 0);                \
  acc[i_][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W1_, X_, acc[i_][1], 0, 0, 0)
    // k-step 0: after each m-block's MFMAs its A fragment register is reloaded with the k-step-1 fragment; the DMA pieces
    // of the next stages are slotted in between the m-blocks (behind MFMAs of this wave)
    CV_ROW(0, x0, w0, w1);
    __builtin_amdgcn_sched_barrier(0);
    x0 = *reinterpret_cast<const bf16x8*>(ab + (xo ^ 32));
    if (n1) { dma_w(0, t + 1, sw1); dma_w(1, t + 1, sw1); }
    __builtin_amdgcn_sched_barrier(0);
    CV_ROW(1, x1, w0, w1);
    __builtin_amdgcn_sched_barrier(0);
    x1 = *reinterpret_cast<const bf16x8*>(ab + (xo ^ 32) + 2048);
    if (n2) { dma_a(0, s2, sa2); dma_a(1, s2, sa2); }
    __builtin_amdgcn_sched_barrier(0);
    CV_ROW(2, x2, w0, w1);
    __builtin_amdgcn_sched_barrier(0);
    x2 = *reinterpret_cast<const bf16x8*>(ab + (xo ^ 32) + 4096);
    if (n2) dma_a(2, s2, sa2);
    __builtin_amdgcn_sched_barrier(0);
    CV_ROW(3, x3, w0, w1);
    __builtin_amdgcn_sched_barrier(0);
    x3 = *reinterpret_cast<const bf16x8*>(ab + (xo ^ 32) + 6144);
    if (n2) dma_a(3, s2, sa2);
    __builtin_amdgcn_sched_barrier(0);
    // k-step 1
    CV_ROW(0, x0, v0, v1);
    CV_ROW(1, x1, v0, v1);
    CV_ROW(2, x2, v0, v1);
    CV_ROW(3, x3, v0, v1);
#undef CV_ROW
    }
    __builtin_amdgcn_sched_barrier(0);
    if (n2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();  // stage t+1 has landed everywhere; nobody reads the slots of stage t any more
    __builtin_amdgcn_sched_barrier(0);
    sa = sa1;
    sw = sw1;
  }

  const int ncol0 = col0 + wn * 64;
  if (F32OUT) {
    // fp32 result straight from the accumulators (attention scores: no bf16 rounding before the softmax); element
    // (row l31 + 32 i, col 32 j + 8 g + 4 hi + r) sits in acc[i][j][4 g + r]
    float* o32 = p.out32 + zb * p.batch_o;
    if constexpr (MF) {   // element (row 16 i + l15, col 16 j + 4 lq + r) sits in acc16[i][j][r]
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t grow = row0 + wm * 128 + i * 16 + l15;
        if (grow < p.M) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float4 v;
            v.x = acc16[i][j][0] * p.out_scale; v.y = acc16[i][j][1] * p.out_scale;
            v.z = acc16[i][j][2] * p.out_scale; v.w = acc16[i][j][3] * p.out_scale;
            *reinterpret_cast<float4*>(o32 + grow * p.ldo + ncol0 + j * 16 + 4 * lq) = v;
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < (MF ? 0 : 4); ++i) {
      const int64_t grow = row0 + wm * 128 + i * 32 + l31;
      if (grow < p.M) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float4 v;
            v.x = acc[i][j][4 * g] * p.out_scale; v.y = acc[i][j][4 * g + 1] * p.out_scale;
            v.z = acc[i][j][4 * g + 2] * p.out_scale; v.w = acc[i][j][4 * g + 3] * p.out_scale;
            *reinterpret_cast<float4*>(o32 + grow * p.ldo + ncol0 + j * 32 + 8 * g + 4 * hi) = v;
          }
      }
    }
  } else {
    // bf16 output through a per-wave LDS image (two passes of 64 rows), so the global stores are whole 128-byte row segments
    bf16_t* ob = p.out + zb * p.batch_o;
    const bf16_t* rb = p.res != nullptr ? p.res + zb * p.batch_o : nullptr;
    char* st = smem + wave * OUT_WAVE_BYTES;
    uint2 bb[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) bb[j][g] = make_uint2(0, 0);
    if (p.bias != nullptr && !MF) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) bb[j][g] = *reinterpret_cast<const uint2*>(p.bias + ncol0 + j * 32 + 8 * g + 4 * hi);
    }
#pragma unroll
    for (int ih = 0; ih < 2; ++ih) {
      const int wrow0 = row0 + wm * 128 + ih * 64;
      uint4 rres[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) rres[it] = make_uint4(0, 0, 0, 0);
      if (rb != nullptr) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int q = lane + 64 * it;
          const int m_local = q >> 3, c = q & 7;
          int grow = wrow0 + m_local;
          grow = grow < p.M ? grow : p.M - 1;
          rres[it] = *reinterpret_cast<const uint4*>(rb + (int64_t)grow * p.ldr + ncol0 + c * 8);
        }
      }
      if constexpr (MF) {
        uint2 bb16[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bb16[j] = p.bias != nullptr ? *reinterpret_cast<const uint2*>(p.bias + ncol0 + j * 16 + 4 * lq) : make_uint2(0, 0);
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
          const int i = ih * 4 + i2;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc16[i][j][r];
            v[0] += bflo(bb16[j].x); v[1] += bfhi(bb16[j].x); v[2] += bflo(bb16[j].y); v[3] += bfhi(bb16[j].y);
            uint2 o;
            o.x = pack2bf(v[0], v[1]);
            o.y = pack2bf(v[2], v[3]);
            *reinterpret_cast<uint2*>(st + (i2 * 16 + l15) * OUT_ROW_BYTES + (j * 16 + 4 * lq) * 2) = o;
          }
        }
      }
#pragma unroll
      for (int i2 = 0; i2 < (MF ? 0 : 2); ++i2) {
        const int i = ih * 2 + i2;
        const int m_local = i2 * 32 + l31;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n_local = j * 32 + 8 * g + 4 * hi;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][4 * g + r];
            v[0] += bflo(bb[j][g].x); v[1] += bfhi(bb[j][g].x); v[2] += bflo(bb[j][g].y); v[3] += bfhi(bb[j][g].y);
            uint2 o;
            o.x = pack2bf(v[0], v[1]);
            o.y = pack2bf(v[2], v[3]);
            *reinterpret_cast<uint2*>(st + m_local * OUT_ROW_BYTES + n_local * 2) = o;
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      uint4 val[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int q = lane + 64 * it;
        val[it] = *reinterpret_cast<const uint4*>(st + (q >> 3) * OUT_ROW_BYTES + (q & 7) * 16);
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int q = lane + 64 * it;
        const int64_t grow = wrow0 + (q >> 3);
        const int gcol = ncol0 + (q & 7) * 8;
        uint4 v = val[it];
        if (rb != nullptr) {  // the conv result is rounded to bf16 first, then added: x + conv(h) as the reference computes it
          float a[8], b[8];
          unpack8(v, a);
          unpack8(rres[it], b);
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] += b[e];
          v = pack8(a);
        }
        if (grow < p.M) *reinterpret_cast<uint4*>(ob + grow * p.ldo + gcol) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // image reads done before the next pass overwrites it
      __builtin_amdgcn_wave_barrier();
    }
  }
#endif
}

}  // namespace

int launch_conv(const ConvParams& p, hipStream_t stream) {
  if (p.M <= 0 || p.batch <= 0) return 0;
  if (p.N <= 0 || p.N % BN != 0 || p.K <= 0 || p.K % BK != 0) return VSYS_ERR_SHAPE;
  if (p.cin % BK != 0 || p.taps < 1 || p.taps > 27 || (int64_t)p.cin * p.taps != p.K) return VSYS_ERR_SHAPE;
  if (p.taps > 1 && (p.cin / BK) != (1 << p.cshift)) return VSYS_ERR_SHAPE;  // channel chunks per tap must be a power of two
  if ((p.lda % 8) || (p.ldw % 8) || (p.ldo % 4) || (p.res && (p.ldr % 8))) return VSYS_ERR_ALIGN;
  if (p.out32 == nullptr && (p.ldo % 8)) return VSYS_ERR_ALIGN;
  if (((int64_t)BM + p.max_tap_rows) * p.lda * 2 + (int64_t)p.cin * 2 >= 0x7fffffff || p.ldw * 256 + (int64_t)p.K * 2 >= 0x7fffffff)
    return VSYS_ERR_SHAPE;  // tile-relative operand offsets are 32-bit (buffer addressing)
  const int64_t nbm = ((int64_t)p.M + BM - 1) / BM, nbn = p.N / BN;
  if (nbm * nbn > 0x7fffffff || p.batch > 65535) return VSYS_ERR_SHAPE;
  static std::atomic<unsigned long long> attr_seen{0};   // per device (and per template instance)
  for (DeviceOnce once(attr_seen); once.todo(); once.done()) {
    (void)hipFuncSetAttribute((const void*)conv_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)conv_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)conv_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)conv_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  }
  const dim3 grid((unsigned)(nbm * nbn), (unsigned)p.batch);
  static const bool mf16 = [] { const char* e = getenv("VSYS_GEMM_MF16"); return !(e && e[0] == '0'); }();   // (0: the 32x32x16 form)
  if (mf16) {
    if (p.out32 != nullptr) hipLaunchKernelGGL((conv_kernel<1, 1>), grid, dim3(256), LDS_BYTES, stream, p);
    else hipLaunchKernelGGL((conv_kernel<0, 1>), grid, dim3(256), LDS_BYTES, stream, p);
  } else if (p.out32 != nullptr) hipLaunchKernelGGL((conv_kernel<1, 0>), grid, dim3(256), LDS_BYTES, stream, p);
  else hipLaunchKernelGGL((conv_kernel<0, 0>), grid, dim3(256), LDS_BYTES, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

}  // namespace vsys
