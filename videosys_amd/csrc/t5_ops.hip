// T5 (v1.1, gated-GELU) encoder pieces that are not GEMMs — the text encoder every pipeline of the reference calls once per
// prompt (`T5EncoderModel`, transformers, third-party; call sites pipeline_open_sora.py:269-287, pipeline_latte.py,
// pipeline_cogvideox.py:211-247).  The linears run on conv_bf16.hip's 128-column GEMM; here: embedding gather, T5LayerNorm
// (RMS, no mean, no bias), the gated-GELU product and the self-attention with its additive relative-position bias.
#include "common.h"
#include "vsys_internal.h"

namespace vsys {
namespace {

// out[i, :] = table[ids[i], :]
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ table, const int64_t* __restrict__ ids,
                                                          bf16_t* __restrict__ out, int64_t n, int C, int64_t vocab) {
  const int cch = C >> 3;
  const int64_t total = n * cch;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cch;
    const int c = (int)(i - r * cch);
    int64_t id = ids[r];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    *reinterpret_cast<uint4*>(out + r * C + c * 8) = *reinterpret_cast<const uint4*>(table + id * C + c * 8);
  }
}

// T5LayerNorm: y = w * bf16(x * rsqrt(mean(x^2) + eps)), statistics in fp32 (modeling_t5.py T5LayerNorm.forward).  One wave per
// row, the row is held in registers (C <= 64 * 8 * MAXV).
template <int MAXV>
__global__ __launch_bounds__(256) void rms_norm_rows_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            bf16_t* __restrict__ y, int64_t rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunk = C >> 3;
  const bf16_t* xr = x + row * C;
  float v[MAXV][8];
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      unpack8(*reinterpret_cast<const uint4*>(xr + c * 8), v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) q += v[i][e] * v[i][e];
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      float g[8];
      unpack8(*reinterpret_cast<const uint4*>(w + c * 8), g);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = g[e] * bf2f(f2bf(v[i][e] * rstd));
      *reinterpret_cast<uint4*>(y + row * C + c * 8) = pack8(v[i]);
    }
  }
}

// T5DenseGatedActDense middle: out = bf16(gelu_new(a)) * b with [a | b] = the fused wi_0 | wi_1 GEMM output [rows, 2F]
__global__ __launch_bounds__(256) void geglu_kernel(const bf16_t* __restrict__ h, bf16_t* __restrict__ out, int64_t rows, int F) {
  const int cch = F >> 3;
  const int64_t total = rows * cch;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cch;
    const int c = (int)(i - r * cch);
    float a[8], b[8];
    unpack8(*reinterpret_cast<const uint4*>(h + r * 2 * F + c * 8), a);
    unpack8(*reinterpret_cast<const uint4*>(h + r * 2 * F + F + c * 8), b);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = bf2f(f2bf(gelu_tanh(a[e]))) * b[e];
    *reinterpret_cast<uint4*>(out + r * F + c * 8) = pack8(a);
  }
}

// T5 self-attention, d_kv = 64: softmax(q k^T + bias[h][j - i]) v over the first klen[b] keys (no 1/sqrt(d) scaling;
// T5Attention.forward).  Block = (32 query rows, head, sample): the head's K and V ([L, 64] bf16 each) sit in LDS with rows
// padded to 66 elements (33 dwords: lane j reading row j is conflict-free); a wave owns one query row at a time: lanes = keys
// for q k^T (each lane a 64-long dot product against LDS), lanes = output channels for p v.  fp32 throughout.
constexpr int T5_ROWS = 32;
constexpr int T5_KSTRIDE = 66;
__global__ __launch_bounds__(256) void t5_attention_kernel(const bf16_t* __restrict__ qkv, int64_t row_stride, int inner,
                                                           const float* __restrict__ relbias, const int* __restrict__ klen,
                                                           bf16_t* __restrict__ out, int64_t out_stride, int L) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Vs = Ks + (size_t)L * T5_KSTRIDE;
  float* qs = reinterpret_cast<float*>(Vs + (size_t)L * T5_KSTRIDE);  // [4 waves][64]
  float* ps = qs + 4 * 64;                                            // [4 waves][Lpad]
  const int Lpad = (L + 63) & ~63;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const int nk = klen[b] < L ? klen[b] : L;
  const bf16_t* base = qkv + (int64_t)b * L * row_stride + h * 64;
  // stage K and V of this head: item = (row, 4-byte pair)
  for (int i = tid; i < L * 32; i += 256) {
    const int r = i >> 5, c = i & 31;
    const uint32_t kk = *reinterpret_cast<const uint32_t*>(base + (int64_t)r * row_stride + inner + 2 * c);
    const uint32_t vv = *reinterpret_cast<const uint32_t*>(base + (int64_t)r * row_stride + 2 * inner + 2 * c);
    *reinterpret_cast<uint32_t*>(Ks + r * T5_KSTRIDE + 2 * c) = kk;
    *reinterpret_cast<uint32_t*>(Vs + r * T5_KSTRIDE + 2 * c) = vv;
  }
  __syncthreads();
  const float* bias_h = relbias + (int64_t)h * (2 * L - 1) + (L - 1);  // index j - i
  float* qw = qs + wave * 64;
  float* pw = ps + wave * Lpad;
  for (int rr = 0; rr < T5_ROWS / 4; ++rr) {
    const int i = blockIdx.x * T5_ROWS + wave * (T5_ROWS / 4) + rr;
    if (i >= L) break;  // wave-uniform
    qw[lane] = bf2f(base[(int64_t)i * row_stride + lane]);
    __builtin_amdgcn_wave_barrier();
    float mx = -3.0e38f;
    for (int j0 = 0; j0 < nk; j0 += 64) {
      const int j = j0 + lane;
      float s = -3.0e38f;
      if (j < nk) {
        const uint32_t* kr = reinterpret_cast<const uint32_t*>(Ks + j * T5_KSTRIDE);
        float acc = 0.f;
#pragma unroll 8
        for (int c = 0; c < 32; ++c) {
          const uint32_t u = kr[c];
          acc += qw[2 * c] * bflo(u) + qw[2 * c + 1] * bfhi(u);
        }
        s = acc + bias_h[j - i];
      }
      pw[j] = s;
      mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j0 = 0; j0 < nk; j0 += 64) {
      const int j = j0 + lane;
      const float e = j < nk ? __expf(pw[j] - mx) : 0.f;
      pw[j] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    __builtin_amdgcn_wave_barrier();
    float o = 0.f;
    for (int j = 0; j < nk; ++j) o += pw[j] * bf2f(Vs[j * T5_KSTRIDE + lane]);
    out[((int64_t)b * L + i) * out_stride + h * 64 + lane] = f2bf(sum > 0.f ? o / sum : 0.f);
    __builtin_amdgcn_wave_barrier();
  }
}

// Split-K epilogue of the skinny (few rows, weight-streaming) linears: the GEMM ran TRANSPOSED and in S slices of K —
// part[s][n][m] = sum over slice s of W[n][k] x[m][k] (fp32, conv_bf16.hip with the weight as the row operand so that a weight
// panel is fetched from HBM once) — and this kernel finishes it: out[m][n] = bf16(sum_s part[s][n][m]) (+ res[m][n],
// added after the rounding like the GEMM epilogues do).  Tile 64 n x 64 m through LDS: 16-byte reads along m, 16-byte writes along n.
__global__ __launch_bounds__(256) void splitk_reduce_t_kernel(const float* __restrict__ part, int S, int64_t slab, int ldp,
                                                              const bf16_t* __restrict__ res, int64_t ldr, bf16_t* __restrict__ out,
                                                              int64_t ldo, int M, int N) {
  __shared__ float tile[64][68];
  const int t = threadIdx.x;
  const int n0 = blockIdx.x * 64, m0 = blockIdx.y * 64;
  {
    const int mq = t & 15, nr = t >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int nl = nr + 16 * i;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n0 + nl < N && m0 + 4 * mq < ldp) {
        const float* src = part + (int64_t)(n0 + nl) * ldp + m0 + 4 * mq;
        for (int s = 0; s < S; ++s) {
          const float4 v = *reinterpret_cast<const float4*>(src + s * slab);
          a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
      }
      *reinterpret_cast<float4*>(&tile[nl][4 * mq]) = a;
    }
  }
  __syncthreads();
  {
    const int nc = t & 7, mr = t >> 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ml = mr + 32 * i;
      const int m = m0 + ml, n = n0 + 8 * nc;
      if (m < M && n < N) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tile[8 * nc + e][ml];
        if (res != nullptr) {
          float r[8];
          unpack8(*reinterpret_cast<const uint4*>(res + (int64_t)m * ldr + n), r);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = bf2f(f2bf(v[e])) + r[e];
        }
        *reinterpret_cast<uint4*>(out + (int64_t)m * ldo + n) = pack8(v);
      }
    }
  }
}

// The same finish for slices that are already [s][m][n] (launch_gemm2_slices): out[m][n] = bf16(sum_s part[s][m][n]) (+ res[m][n]);
// thread = 8 consecutive n.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int S, int64_t slab, int64_t ldp,
                                                            const bf16_t* __restrict__ res, int64_t ldr, bf16_t* __restrict__ out,
                                                            int64_t ldo, int M, int N) {
  const int nch = N >> 3;
  const int64_t total = (int64_t)M * nch;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / nch;
    const int n = (int)(i - m * nch) * 8;
    const float* src = part + m * ldp + n;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) {
      const float4 a = *reinterpret_cast<const float4*>(src + s * slab), b = *reinterpret_cast<const float4*>(src + s * slab + 4);
      v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
    if (res != nullptr) {
      float r[8];
      unpack8(*reinterpret_cast<const uint4*>(res + m * ldr + n), r);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = bf2f(f2bf(v[e])) + r[e];
    }
    *reinterpret_cast<uint4*>(out + m * ldo + n) = pack8(v);
  }
}

inline unsigned stream_grid(int64_t items) {
  int64_t b = (items + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

int launch_gather_rows(const bf16_t* table, const int64_t* ids, bf16_t* out, int64_t n, int C, int64_t vocab, hipStream_t stream) {
  if (n <= 0) return 0;
  if (C % 8 != 0 || vocab <= 0) return VSYS_ERR_SHAPE;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(stream_grid(n * (C >> 3))), dim3(256), 0, stream, table, ids, out, n, C, vocab);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_rms_norm_rows(const bf16_t* x, const bf16_t* w, bf16_t* y, int64_t rows, int C, float eps, hipStream_t stream) {
  if (rows <= 0) return 0;
  if (C % 8 != 0 || C > 64 * 8 * 16) return VSYS_ERR_SHAPE;
  const unsigned grid = (unsigned)((rows + 3) / 4);
  if (C <= 64 * 8 * 4) hipLaunchKernelGGL(rms_norm_rows_kernel<4>, dim3(grid), dim3(256), 0, stream, x, w, y, rows, C, eps);
  else if (C <= 64 * 8 * 8) hipLaunchKernelGGL(rms_norm_rows_kernel<8>, dim3(grid), dim3(256), 0, stream, x, w, y, rows, C, eps);
  else hipLaunchKernelGGL(rms_norm_rows_kernel<16>, dim3(grid), dim3(256), 0, stream, x, w, y, rows, C, eps);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_geglu(const bf16_t* h, bf16_t* out, int64_t rows, int F, hipStream_t stream) {
  if (rows <= 0) return 0;
  if (F % 8 != 0) return VSYS_ERR_SHAPE;
  hipLaunchKernelGGL(geglu_kernel, dim3(stream_grid(rows * (F >> 3))), dim3(256), 0, stream, h, out, rows, F);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_splitk_reduce_t(const float* part, int S, int64_t slab, int ldp, const bf16_t* res, int64_t ldr, bf16_t* out, int64_t ldo,
                           int M, int N, hipStream_t stream) {
  if (M <= 0 || N <= 0) return 0;
  if (S < 1 || S > 64 || N % 8 != 0 || ldp % 4 != 0 || ldp < M || slab < (int64_t)N * ldp || (ldo % 8) || (res && (ldr % 8)))
    return VSYS_ERR_SHAPE;
  hipLaunchKernelGGL(splitk_reduce_t_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, stream, part, S, slab, ldp, res, ldr, out,
                     ldo, M, N);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_splitk_reduce(const float* part, int S, int64_t slab, int64_t ldp, const bf16_t* res, int64_t ldr, bf16_t* out, int64_t ldo,
                         int M, int N, hipStream_t stream) {
  if (M <= 0 || N <= 0) return 0;
  if (S < 1 || S > 64 || N % 8 != 0 || ldp % 4 != 0 || ldp < N || slab < (int64_t)M * ldp || (ldo % 8) || (res && (ldr % 8)))
    return VSYS_ERR_SHAPE;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(stream_grid((int64_t)M * (N >> 3))), dim3(256), 0, stream, part, S, slab, ldp, res, ldr, out,
                     ldo, M, N);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_t5_attention(const bf16_t* qkv, int64_t row_stride, int inner, const float* relbias, const int* klen, bf16_t* out,
                        int64_t out_stride, int B, int L, int heads, hipStream_t stream) {
  if (B <= 0 || L <= 0) return 0;
  if (heads <= 0 || inner != heads * 64 || (row_stride % 2) || B > 65535 || heads > 65535) return VSYS_ERR_SHAPE;
  const int Lpad = (L + 63) & ~63;
  const size_t lds = (size_t)2 * L * T5_KSTRIDE * 2 + 4 * 64 * 4 + (size_t)4 * Lpad * 4;
  if (lds > 160 * 1024) return VSYS_ERR_SHAPE;  // L <= ~560 keys
  static size_t attr = 0;
  if (lds > attr) {
    (void)hipFuncSetAttribute((const void*)t5_attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = lds;
  }
  hipLaunchKernelGGL(t5_attention_kernel, dim3((L + T5_ROWS - 1) / T5_ROWS, heads, B), dim3(256), lds, stream, qkv, row_stride, inner,
                     relbias, klen, out, out_stride, L);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

}  // namespace vsys
