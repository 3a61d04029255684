// attn_temporal_d72 on the matrix pipe — LAB kernel (vsys_tune_flash_variant(7), T <= 32), NOT dispatched by default.
// Written at the end of round 1 after the GPU budget was spent: it compiles for gfx950 and follows the fragment / accumulator
// conventions the other kernels of this library are tested with, but it has NOT run on hardware yet.  The parity tests that will
// judge it are tests/test_gpu_parity.py::test_attn_temporal_golden / test_attn_temporal_config2_vs_torch with the variant set.
//
// Why: the shipped kernel (attention.hip, attn_temporal_d72_v2) spends its time in 228 broadcast ds_read_b128 per wave and in
// 3-lane score reductions (2.2 TB/s of a ~5 TB/s stream).  Here one wave owns one (batch, pixel, head):
//   * lane (l31, hi) loads 16-byte chunks of frame l31 at head dims 16 s + 8 hi (s = 0..4) of q, k, v — one round trip, and
//     exactly the A / B operand layout of v_mfma_f32_32x32x16_bf16, so q and k never touch LDS;
//   * S^T[key][query] = K . Q^T in 5 MFMAs (d padded 72 -> 80 with a zero chunk); MFMA row m carries key
//     sigma(m) = 16 (m>>4) + 8 ((m>>2)&1) + 4 ((m>>3)&1) + (m&3), so a lane's 16 accumulators are keys 16 s2 + 8 hi + e: the softmax
//     is lane-local (+ one exchange with lane^32) and P^T converts to the PV B-fragment with no cross-lane traffic;
//   * V goes through a wave-private 5 KiB LDS tile once (row-major write, 2-byte column reads = the transpose the PV A-operand
//     needs: 48 ds_read_u16 per lane instead of 228 ds_read_b128 per wave);
//   * O^T[d][query] = V^T . P^T in 6 MFMAs (d padded to 96, keys to 32), divided by the row sum and stored as 8-byte runs.
// RMS qk-norm (LlamaRMSNorm rounding points), RoPE (tables staged once per workgroup in LDS) and the 72^-1/2 scale are as in
// the shipped kernel.
#include "common.h"
#include "vsys_internal.h"

namespace vsys {
namespace {

constexpr int HD = 72;
constexpr int VROWB = 160;            // bytes per V row in the wave-private LDS tile (80 bf16: 16-byte aligned chunks)
constexpr int VTILE = 32 * VROWB;     // 5120
constexpr float NEG_BIG = -1.0e30f;

__global__ __launch_bounds__(256) void attn_temporal_d72_mfma_kernel(const bf16_t* __restrict__ qkv, int64_t row_stride, int C,
                                                                     const bf16_t* __restrict__ q_norm_w, const bf16_t* __restrict__ k_norm_w,
                                                                     const float* __restrict__ rope_cos, const float* __restrict__ rope_sin,
                                                                     bf16_t* __restrict__ out, int64_t out_stride, int B, int T, int S,
                                                                     int heads, float eps, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS: [4 waves] V tiles, then cos[T][72], sin[T][72] fp32 (when RoPE is on)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  char* vtile = smem + wave * VTILE;
  float* cs = reinterpret_cast<float*>(smem + 4 * VTILE);
  float* sn = cs + T * HD;
  const bool rope = rope_cos != nullptr;
  if (rope) {
    for (int i = tid; i < T * HD; i += 256) { cs[i] = rope_cos[i]; sn[i] = rope_sin[i]; }
  }
  __syncthreads();

  const int hgroups = (heads + 3) / 4;
  const int64_t nwork = (int64_t)B * S * hgroups;
  // frame of the K row this lane feeds to MFMA row l31 (see header), and of its Q / V row
  const int fk = 16 * (l31 >> 4) + 8 * ((l31 >> 2) & 1) + 4 * ((l31 >> 3) & 1) + (l31 & 3);
  const int fq = l31;

  for (int64_t wk = blockIdx.x; wk < nwork; wk += gridDim.x) {
    const int hg = (int)(wk % hgroups);
    const int64_t bs = wk / hgroups;
    const int s = (int)(bs % S), b = (int)(bs / S);
    const int h = hg * 4 + wave;
    if (h < heads) {  // wave-uniform; no block-level barrier inside the loop
      // ---- one round trip: 5 chunks each of q (frame fq), k (frame fk), v (frame fq); chunk 4 of the hi = 1 half is padding
      uint4 rq[5], rk[5], rv[5];
      const bool okq = fq < T, okk = fk < T;
      const bf16_t* rowq = qkv + (((int64_t)b * T + (okq ? fq : 0)) * S + s) * row_stride + h * HD + 8 * hi;
      const bf16_t* rowk = qkv + (((int64_t)b * T + (okk ? fk : 0)) * S + s) * row_stride + h * HD + 8 * hi;
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        const bool in_d = (16 * c + 8 * hi) < HD;
        rq[c] = rk[c] = rv[c] = make_uint4(0, 0, 0, 0);
        if (in_d && okq) {
          rq[c] = *reinterpret_cast<const uint4*>(rowq + 16 * c);
          rv[c] = *reinterpret_cast<const uint4*>(rowq + 2 * C + 16 * c);
        }
        if (in_d && okk) rk[c] = *reinterpret_cast<const uint4*>(rowk + C + 16 * c);
      }
      // ---- V -> wave-private LDS tile, row-major (rows >= T and dims >= 72 are zeros)
#pragma unroll
      for (int c = 0; c < 5; ++c) *reinterpret_cast<uint4*>(vtile + fq * VROWB + (16 * c + 8 * hi) * 2) = rv[c];

      // ---- q, k: LlamaRMSNorm (fp32 statistics over the 72 dims held by lanes (l31, 0) and (l31, 1)), RoPE, back to bf16
      auto norm_rope = [&](const uint4* raw, const bf16_t* w, int frame, bf16x8* frag) {
        float x[5][8];
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < 5; ++c) {
          unpack8(raw[c], x[c]);
#pragma unroll
          for (int e = 0; e < 8; ++e) ss += x[c][e] * x[c][e];
        }
        ss += __shfl_xor(ss, 32, 64);
        const float rstd = rsqrtf(ss / (float)HD + eps);
#pragma unroll
        for (int c = 0; c < 5; ++c) {
          const int d0 = 16 * c + 8 * hi;
          const bool in_d = d0 < HD;
          if (w != nullptr && in_d) {
            float wv[8];
            unpack8(*reinterpret_cast<const uint4*>(w + d0), wv);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[c][e] = bf2f(f2bf(bf2f(f2bf(x[c][e] * rstd)) * wv[e]));
          }
          if (rope && in_d) {
            const float* cr = cs + frame * HD + d0;
            const float* sr = sn + frame * HD + d0;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              const float a = x[c][e], bb = x[c][e + 1];
              x[c][e] = bf2f(f2bf(a * cr[e] - bb * sr[e]));
              x[c][e + 1] = bf2f(f2bf(bb * cr[e + 1] + a * sr[e + 1]));
            }
          }
          const uint4 pk = pack8(x[c]);
          frag[c] = __builtin_bit_cast(bf16x8, pk);
        }
      };
      bf16x8 qf[5], kf[5];
      norm_rope(rq, q_norm_w, okq ? fq : 0, qf);
      norm_rope(rk, k_norm_w, okk ? fk : 0, kf);

      // ---- S^T = K . Q^T: D[m][n] with n = query l31, m = 8 g + 4 hi + r  <->  key 16 (g>>1) + 8 hi + 4 (g&1) + r
      f32x16 sacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
      for (int c = 0; c < 5; ++c) sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[c], qf[c], sacc, 0, 0, 0);
      float m = NEG_BIG;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 16 * (g >> 1) + 8 * hi + 4 * (g & 1) + r;
          const float v = key < T ? sacc[4 * g + r] * scale : NEG_BIG;
          sacc[4 * g + r] = v;
          m = fmaxf(m, v);
        }
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      float l = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __expf(sacc[r] - m);   // masked keys: exp(-1e30 - m) = 0
        sacc[r] = p;
        l += p;
      }
      l += __shfl_xor(l, 32, 64);
      // P^T B-fragments: step s2 holds keys 16 s2 + 8 hi + e = accumulators 8 s2 + e
      bf16x8 pf[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float t8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t8[e] = sacc[8 * s2 + e];
        const uint4 pk = pack8(t8);
        pf[s2] = __builtin_bit_cast(bf16x8, pk);
      }

      // ---- O^T[d][query] = V^T . P^T: A-fragment rows d = 32 blk + l31, k = keys 16 s2 + 8 hi + e read column-wise from the tile
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      f32x16 oacc[3];
#pragma unroll
      for (int blk = 0; blk < 3; ++blk) {
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[blk][r] = 0.f;
        const int d = 32 * blk + l31;   // dims 72..79 of the tile are zero; d >= 80 is outside the tile
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          uint32_t wv[4] = {0u, 0u, 0u, 0u};
          if (d < 80) {
            const bf16_t* col = reinterpret_cast<const bf16_t*>(vtile + (16 * s2 + 8 * hi) * VROWB) + d;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              wv[e] = (uint32_t)col[(2 * e) * (VROWB / 2)] | ((uint32_t)col[(2 * e + 1) * (VROWB / 2)] << 16);
          }
          uint4 pk;
          pk.x = wv[0]; pk.y = wv[1]; pk.z = wv[2]; pk.w = wv[3];
          oacc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pk), pf[s2], oacc[blk], 0, 0, 0);
        }
      }
      // ---- out[(b, frame l31, s)][h*72 + d], d = 32 blk + 8 g + 4 hi + r: runs of 4 dims = 8-byte stores
      if (okq) {
        const float inv = 1.0f / l;
        bf16_t* orow = out + (((int64_t)b * T + fq) * S + s) * out_stride + h * HD;
#pragma unroll
        for (int blk = 0; blk < 3; ++blk)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int d0 = 32 * blk + 8 * g + 4 * hi;
            if (d0 < HD) {
              uint2 o;
              o.x = pack2bf(oacc[blk][4 * g] * inv, oacc[blk][4 * g + 1] * inv);
              o.y = pack2bf(oacc[blk][4 * g + 2] * inv, oacc[blk][4 * g + 3] * inv);
              *reinterpret_cast<uint2*>(orow + d0) = o;
            }
          }
      }
      __builtin_amdgcn_wave_barrier();  // the tile is rewritten by the next work item of this wave
    }
  }
}

}  // namespace

int launch_attn_temporal_d72_mfma(const bf16_t* qkv, int64_t row_stride, int C, const bf16_t* q_norm_w, const bf16_t* k_norm_w,
                                  const float* rope_cos, const float* rope_sin, bf16_t* out, int64_t out_stride, int B, int T, int S,
                                  int heads, float eps, float scale, hipStream_t stream) {
  if (T > 32) return VSYS_ERR_SHAPE;
  const size_t lds = 4 * VTILE + (size_t)2 * T * HD * sizeof(float);
  const int64_t nwork = (int64_t)B * S * ((heads + 3) / 4);
  const int64_t grid = nwork < 256 * 16 ? nwork : 256 * 16;
  hipLaunchKernelGGL(attn_temporal_d72_mfma_kernel, dim3((unsigned)grid), dim3(256), lds, stream, qkv, row_stride, C, q_norm_w, k_norm_w,
                     rope_cos, rope_sin, out, out_stride, B, T, S, heads, eps, scale);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

}  // namespace vsys
