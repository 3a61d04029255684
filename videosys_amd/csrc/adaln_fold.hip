// AdaLN folded into the qkv / fc1 GEMMs (gfx950): the pieces that are not GEMM epilogues.
//
// Reference call sites replaced (/root/reference/videosys/models/transformers/open_sora_transformer_3d.py):
//   :196-197  x_m = t2i_modulate(self.norm1(x), shift_msa, scale_msa)   followed by attentions.py:59 (qkv)
//   :260-261  x_m = t2i_modulate(self.norm2(x), shift_mlp, scale_mlp)   followed by timm Mlp fc1 (:130-132,267)
//
//   out[m][n] = sum_k ((x[m][k] - mu_m) rstd_m (1 + s_k) + h_k) W[n][k] + b[n]
//             = rstd_m (sum_k x[m][k] W'[n][k]  -  mu_m cs[n])  +  cv[n]
//   W'[n][k] = bf16(W[n][k] (1 + s_k)),  cs[n] = sum_k W'[n][k] (of the ROUNDED W', so the mean cancels exactly),
//   cv[n] = sum_k h_k W[n][k] + b[n].
// The modulated activations never exist: the GEMM reads the raw residual stream, its epilogue applies (mu, rstd) and cv
// (gemm_bf16.hip / gemm2_bf16.hip, EPI_LN_*); the row statistics come out of the epilogue that produced the residual stream
// (EPI_GATE_RES_STATS) as per-96-column partials.  What remains is below:
//   adaln_prescale_kernel  W', cs, cv of EVERY site of a step in one launch (s, h depend on the timestep only: the CFG halves
//                          share them), 2 x the weight bytes per step instead of 2 x the activation bytes per site;
//   ln_row_stats_kernel    the partials of a tensor no GEMM epilogue produced (patch embedding, PAB broadcast steps).
#include "common.h"
#include "vsys_internal.h"

namespace vsys {
namespace {

// one site = 10 int64: W, bias, Wp, cs, cv (device addresses), shift_off, scale_off (element offsets into the modulation
// table handed to the launch), N, K, blk0 (first block of the site in the launch's 1-D grid; 4 rows per block)
constexpr int SITE_WORDS = 10;

__global__ __launch_bounds__(256) void adaln_prescale_kernel(const int64_t* __restrict__ sites, int nsites,
                                                             const bf16_t* __restrict__ mod) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int lo = 0, hi = nsites - 1;
  while (lo < hi) {   // last site whose first block is <= this block (wave-uniform scalar loads)
    const int mid = (lo + hi + 1) >> 1;
    if (sites[(int64_t)mid * SITE_WORDS + 9] <= (int64_t)blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const int64_t* s = sites + (int64_t)lo * SITE_WORDS;
  const bf16_t* W = reinterpret_cast<const bf16_t*>(s[0]);
  const bf16_t* bias = reinterpret_cast<const bf16_t*>(s[1]);
  bf16_t* Wp = reinterpret_cast<bf16_t*>(s[2]);
  float* cs = reinterpret_cast<float*>(s[3]);
  float* cv = reinterpret_cast<float*>(s[4]);
  const bf16_t* shift = mod + s[5];
  const bf16_t* scale = mod + s[6];
  const int N = (int)s[7], K = (int)s[8];
  const int n = ((int)blockIdx.x - (int)s[9]) * 4 + wave;
  if (n >= N) return;
  const bf16_t* wr = W + (int64_t)n * K;
  bf16_t* wo = Wp + (int64_t)n * K;
  float a_cs = 0.f, a_cv = 0.f;
  for (int c = lane; c < (K >> 3); c += 64) {
    float w[8], sc[8], sh[8], o[8];
    unpack8(*reinterpret_cast<const uint4*>(wr + c * 8), w);
    unpack8(*reinterpret_cast<const uint4*>(scale + c * 8), sc);
    unpack8(*reinterpret_cast<const uint4*>(shift + c * 8), sh);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      o[e] = w[e] * (1.0f + sc[e]);
      a_cv = fmaf(sh[e], w[e], a_cv);
    }
    const uint4 pk = pack8(o);
    *reinterpret_cast<uint4*>(wo + c * 8) = pk;
    float r[8];
    unpack8(pk, r);
#pragma unroll
    for (int e = 0; e < 8; ++e) a_cs += r[e];
  }
  a_cs = wave_sum(a_cs);
  a_cv = wave_sum(a_cv);
  if (lane == 0) {
    cs[n] = a_cs;
    cv[n] = a_cv + (bias != nullptr ? bf2f(bias[n]) : 0.f);
  }
}

// one wave per row, C = 96 nb <= 1536: lane l holds elements 24 l .. 24 l + 23 (three 16-byte units), so the four lanes of a quad
// hold one 96-column block
__global__ __launch_bounds__(256) void ln_row_stats_kernel(const bf16_t* __restrict__ x, int64_t rows, int C, float2* __restrict__ stats,
                                                           int64_t ld) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int e0 = 24 * lane;
  const bool act = e0 < C;
  float v[24];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const uint4 u = act ? *reinterpret_cast<const uint4*>(x + row * C + e0 + 8 * i) : make_uint4(0, 0, 0, 0);
    unpack8(u, v + 8 * i);
  }
  LnAcc a;
  a.init(v[0]);
#pragma unroll
  for (int e = 0; e < 24; ++e) a.add(v[e]);
  float2 m = a.finish(24.f);
  float2 o = make_float2(__shfl_xor(m.x, 1, 64), __shfl_xor(m.y, 1, 64));
  m = (lane & 1) ? ln_merge_equal(o, m, 24.f) : ln_merge_equal(m, o, 24.f);   // both lanes of a pair compute the same bits
  o = make_float2(__shfl_xor(m.x, 2, 64), __shfl_xor(m.y, 2, 64));
  m = (lane & 2) ? ln_merge_equal(o, m, 48.f) : ln_merge_equal(m, o, 48.f);
  if (act && (lane & 3) == 0) stats[(int64_t)(lane >> 2) * ld + row] = m;
}

}  // namespace

int launch_adaln_prescale(const int64_t* sites, int nsites, int64_t nblocks, const bf16_t* mod, hipStream_t stream) {
  if (nsites <= 0 || nblocks <= 0) return 0;
  if (nblocks > 0x7fffffff) return VSYS_ERR_SHAPE;
  hipLaunchKernelGGL(adaln_prescale_kernel, dim3((unsigned)nblocks), dim3(256), 0, stream, sites, nsites, mod);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_ln_row_stats(const bf16_t* x, int64_t rows, int C, float2* stats, int64_t ld, hipStream_t stream) {
  if (rows <= 0) return 0;
  if (C % LN_BLOCK != 0 || C > 64 * 24 || ld < rows) return VSYS_ERR_SHAPE;
  hipLaunchKernelGGL(ln_row_stats_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, rows, C, stats, ld);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

}  // namespace vsys
