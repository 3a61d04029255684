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

// Two rows per wave; lane (b, hi) = (q >> 1, q & 1), q = lane & 31 < 2 nb, accumulates the SAME 48 columns of block b in the SAME
// order as a lane of the EPI_GATE_RES_STATS epilogue (columns 96 b + 32 j + 8 g + 4 hi + 0..3 for j = 0..2, g = 0..3, pivot = the
// first of them) and the pair is merged the same way: a tensor gets the same partial bits from either producer, so a step that
// takes its statistics from this pass (x last written by a plain epilogue or a PAB broadcast) equals the step that got them from
// the GEMM epilogue bit for bit.  C = 96 nb, nb <= 16.
__global__ __launch_bounds__(256) void ln_row_stats_kernel(const bf16_t* __restrict__ x, int64_t rows, int C, float2* __restrict__ stats,
                                                           int64_t ld) {
  const int lane = threadIdx.x & 63;
  const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);
  const int q = lane & 31, b = q >> 1, hi = q & 1;
  const bool act = row < rows && b * LN_BLOCK < C;
  const bf16_t* xr = x + (act ? row : 0) * C + (act ? b : 0) * LN_BLOCK + 4 * hi;
  uint2 u[12];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) u[4 * j + g] = *reinterpret_cast<const uint2*>(xr + 32 * j + 8 * g);
  LnAcc a;
  a.init(bflo(u[0].x));
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    a.add(bflo(u[k].x)); a.add(bfhi(u[k].x)); a.add(bflo(u[k].y)); a.add(bfhi(u[k].y));
  }
  const float2 mine = a.finish(48.f);
  const float2 other = make_float2(__shfl_xor(mine.x, 1, 64), __shfl_xor(mine.y, 1, 64));
  const float2 blk = ln_merge_equal(mine, other, 48.f);   // (as the epilogue: evaluated by the hi = 0 lane, mine = its own half)
  if (act && hi == 0) stats[(int64_t)b * ld + row] = blk;
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
  if (C % LN_BLOCK != 0 || C > 16 * LN_BLOCK || ld < rows) return VSYS_ERR_SHAPE;
  hipLaunchKernelGGL(ln_row_stats_kernel, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, stream, x, rows, C, stats, ld);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

}  // namespace vsys
