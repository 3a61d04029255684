// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of videosys_amd.
// Everything here is written for wave64 + MFMA on gfx950 only; there is no other backend.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vsys {

typedef uint16_t bf16_t;  // raw bfloat16 bits in HBM
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
__device__ __forceinline__ float bflo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bfhi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// 8 bf16 (one uint4) -> 8 floats
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = bflo(v.x); f[1] = bfhi(v.x); f[2] = bflo(v.y); f[3] = bfhi(v.y);
  f[4] = bflo(v.z); f[5] = bfhi(v.z); f[6] = bflo(v.w); f[7] = bfhi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]); v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// nn.GELU(approximate="tanh"): 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3).
// 0.5 (1 + tanh(u)) == sigmoid(2u) == 1 / (1 + exp(-2u)): one v_exp_f32 + one v_rcp_f32, no division sequence.
__device__ __forceinline__ float gelu_tanh(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float u = k0 * x * (1.0f + k1 * x * x);
  const float e = __builtin_amdgcn_exp2f(u * -2.8853900817779268f);  // exp(-2u); +inf for very negative x -> result -0
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

// ---- AdaLN fold: LayerNorm statistics as per-96-column partials (mean_b, M2_b) at st[b * ld + row] (vsys_internal.h GemmParams).
// (mu, rstd) of one row from its nb <= 12 partials.  Chan's combination around the first block's mean: no E[x^2] - mu^2 form
// anywhere, so a row mean that is many sigma away from zero costs nothing.
__device__ __forceinline__ void ln_combine(const float2* __restrict__ st, int64_t ld, int nb, int64_t row, float eps, float& mu,
                                           float& rstd) {
  float2 v[12];
#pragma unroll
  for (int b = 0; b < 12; ++b) v[b] = b < nb ? st[b * ld + row] : make_float2(0.f, 0.f);
  float sd = 0.f, sdd = 0.f, m2 = v[0].y;
#pragma unroll
  for (int b = 1; b < 12; ++b) {
    if (b < nb) {
      const float d = v[b].x - v[0].x;
      sd += d;
      sdd += d * d;
      m2 += v[b].y;
    }
  }
  const float inv = 1.0f / (float)nb;
  mu = v[0].x + sd * inv;
  m2 += 96.0f * (sdd - sd * sd * inv);
  rstd = rsqrtf(fmaxf(m2, 0.f) / (96.0f * (float)nb) + eps);
}
// Cooperative staging of what an EPI_LN_* epilogue needs for one output tile into LDS, so that the epilogue itself reads no
// global memory (a dependent global round trip per row pair / column block is ~1 us each under load and nothing covers it at
// one or two workgroups per CU): rows [row0, row0 + nrows) -> float2 (mu, rstd) at lds[r], columns [col0, col0 + ncols) ->
// float2 (cs, cv) at lds[nrows + c].  Every thread of the workgroup calls it; the caller orders the writes against the reads
// (lgkmcnt(0) + barrier).
__device__ __forceinline__ void ln_stage_tile(float2* __restrict__ lds, const float2* __restrict__ st, int64_t ld, int nb, float eps,
                                              int M, int row0, int nrows, const float* __restrict__ cs, const float* __restrict__ cv,
                                              int col0, int ncols, int tid, int nthreads) {
  // one row and one column per thread (nrows, ncols <= nthreads at every call site); the column loads are issued in front of
  // the row partials so that ONE round trip covers both
  const int c = nthreads - 1 - tid;
  float c_s = 0.f, c_v = 0.f;
  if (c < ncols) {
    c_s = cs[col0 + c];
    c_v = cv[col0 + c];
  }
  if (tid < nrows) {
    int grow = row0 + tid;
    grow = grow < M ? grow : M - 1;
    float mu, rstd;
    ln_combine(st, ld, nb, grow, eps, mu, rstd);
    lds[tid] = make_float2(mu, rstd);
  }
  if (c < ncols) lds[nrows + c] = make_float2(c_s, c_v);
}

// running (sum, sum of squares) of values taken relative to a pivot close to them -> (mean, M2) of n values
struct LnAcc {
  float p, s1, s2;
  __device__ __forceinline__ void init(float pivot) { p = pivot; s1 = 0.f; s2 = 0.f; }
  __device__ __forceinline__ void add(float x) { const float d = x - p; s1 += d; s2 = fmaf(d, d, s2); }
  __device__ __forceinline__ float2 finish(float n) const { return make_float2(p + s1 / n, fmaxf(s2 - s1 * s1 / n, 0.f)); }
};
// two equal-sized (n each) partials -> one
__device__ __forceinline__ float2 ln_merge_equal(float2 a, float2 b, float n) {
  const float d = b.x - a.x;
  return make_float2(a.x + 0.5f * d, a.y + b.y + d * d * (0.5f * n));
}

// XCD-aware bijective remap of a 1-D block id: block b runs on XCD b%8; give each XCD a contiguous
// chunk of logical tile ids so neighbouring tiles (sharing an operand panel) hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int NX = 8;
  int xcd = bid % NX, idx = bid / NX;
  int q = nblocks / NX, r = nblocks % NX;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// Tile raster of the GEMM kernels (gemm_bf16.hip, gemm2_bf16.hip).  The row panels are split into 8 contiguous groups, one per
// XCD chunk of xcd_remap; inside a group the order is  for panel-chunk (ph panels; 0 = all of the group's):  for column group
// (gw column tiles):  for panel:  for column.  The tiles an XCD runs at once are then a compact (panels x gw) rectangle whose W
// slab stays in the XCD's 4 MiB L2 while its panels sweep by.  gw >= nbn or gw <= 0: plain row-major.
__host__ __device__ __forceinline__ void gemm_raster(int tile, int nbm, int nbn, int gw, int ph, int& bm, int& bn) {
  if (gw <= 0 || gw >= nbn) {
    bm = tile / nbn;
    bn = tile - bm * nbn;
    return;
  }
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
  int pc0 = 0, npc = np;   // panel chunk: panels [pc0, pc0 + npc) of the group
  if (ph > 0 && ph < np) {
    const int per = ph * nbn;
    int c = off / per;
    const int nch = (np + ph - 1) / ph;
    c = c < nch - 1 ? c : nch - 1;
    pc0 = c * ph;
    npc = c < nch - 1 ? ph : np - pc0;
    off -= c * per;
  }
  const int ng = (nbn + gw - 1) / gw;
  int g = off / (npc * gw);
  g = g < ng - 1 ? g : ng - 1;
  const int off2 = off - g * npc * gw;
  const int width = g < ng - 1 ? gw : nbn - (ng - 1) * gw;
  const int pm = off2 / width;
  bm = p0 + pc0 + pm;
  bn = g * gw + (off2 - pm * width);
}

}  // namespace vsys
