// HBM-bound kernels around the implicit-GEMM convolutions of the VAE decoders:
// GroupNorm(32) statistics + normalise/affine/SiLU, grid-to-grid copies (re-pad, nearest 2x upsample), the temporal
// depth-to-space of the MAGVIT-style decoder, the small-channel first layer (im2col of a 4-channel latent), the planar
// extraction of the last layer and the row softmax of the mid-block attention.
//
// Every activation is a channels-last row matrix over a grid described by VaeGrid: sample n, frame t, pixel (h, w) lives at
// row  n*sample_rows + ((t + tf)*(H + 2 pad) + h + pad)*(W + 2 pad) + w + pad.  Kernels touch INTERIOR rows only; padded
// buffers are zero-filled once by the host and their borders are never written (that is what makes the tap-shifted conv
// in conv_bf16.hip see zero padding).
#include "common.h"
#include "vsys_internal.h"

namespace vsys {
namespace {

__device__ __forceinline__ int64_t grid_row(const VaeGrid& g, int n, int t, int h, int w) {
  return (int64_t)n * g.sample_rows + ((int64_t)(t + g.tf) * (g.H + 2 * g.pad) + h + g.pad) * (g.W + 2 * g.pad) + w + g.pad;
}

// ---- GroupNorm statistics, pass 1: block (blk, n) sums positions blk, blk + nblk, ... of sample n.  Thread = (position
// lane, 8-channel chunk); per chunk two half sums (4 channels each) so 4-channel groups (C = 128) need no special case.
// partial[n][blk][C/4][2] fp32 (sum, sum of squares).
__global__ __launch_bounds__(256) void gn_partial_kernel(const bf16_t* __restrict__ x, VaeGrid g, int C, float* __restrict__ partial) {
  __shared__ float red[256][4];
  const int nch = C >> 3;                 // chunks per row (16..128), a power of two <= 256 is not required
  const int lanes = 256 / nch;            // position lanes per block
  const int tid = threadIdx.x;
  const int ch = tid % nch, pl = tid / nch;
  const int n = blockIdx.y, nblk = gridDim.x;
  const int64_t P = (int64_t)g.T * g.H * g.W;
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
  if (pl < lanes) {
    for (int64_t pos = (int64_t)blockIdx.x * lanes + pl; pos < P; pos += (int64_t)nblk * lanes) {
      const int w = (int)(pos % g.W);
      const int64_t r1 = pos / g.W;
      const int h = (int)(r1 % g.H), t = (int)(r1 / g.H);
      const uint4 v = *reinterpret_cast<const uint4*>(x + grid_row(g, n, t, h, w) * C + ch * 8);
      float f[8];
      unpack8(v, f);
      s0 += (f[0] + f[1]) + (f[2] + f[3]);
      q0 += (f[0] * f[0] + f[1] * f[1]) + (f[2] * f[2] + f[3] * f[3]);
      s1 += (f[4] + f[5]) + (f[6] + f[7]);
      q1 += (f[4] * f[4] + f[5] * f[5]) + (f[6] * f[6] + f[7] * f[7]);
    }
  }
  red[tid][0] = s0; red[tid][1] = q0; red[tid][2] = s1; red[tid][3] = q1;
  __syncthreads();
  if (tid < nch) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int l = 0; l < lanes; ++l) {  // fixed order: deterministic
      a0 += red[l * nch + tid][0]; a1 += red[l * nch + tid][1]; a2 += red[l * nch + tid][2]; a3 += red[l * nch + tid][3];
    }
    float* o = partial + (((int64_t)n * nblk + blockIdx.x) * (C >> 2) + tid * 2) * 2;
    o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
  }
}

// pass 2: one wave per (n, group): fold the block partials in double (fixed lane assignment + butterfly: deterministic),
// emit mean and 1/sqrt(var + eps).  A lane's items are fetched eight at a time BEFORE they are added (the partials were written by
// another kernel a moment ago: every read is an L2 miss, and a load-add-load-add chain took 35-100 us per launch — 16 ms of a
// 64-frame decode — for a few KB of data; the order of the additions is unchanged).
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ partial, int nblk, int C, int groups,
                                                         int64_t count_per_channel, float eps, float* __restrict__ stats) {
  const int idx = blockIdx.x;
  const int n = idx / groups, grp = idx - n * groups;
  const int hpg = (C / groups) >> 2;  // 4-channel half chunks per group
  const int items = nblk * hpg;
  const float* base = partial + ((int64_t)n * nblk * (C >> 2) + grp * hpg) * 2;
  double s = 0.0, q = 0.0;
  for (int i0 = threadIdx.x; i0 < items; i0 += 64 * 8) {
    float2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + 64 * u;
      v[u] = make_float2(0.f, 0.f);
      if (i < items) {
        const int b = i / hpg, k = i - b * hpg;
        v[u] = *reinterpret_cast<const float2*>(base + ((int64_t)b * (C >> 2) + k) * 2);
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { s += v[u].x; q += v[u].y; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
  if (threadIdx.x == 0) {
    const double cnt = (double)count_per_channel * (C / groups);
    const double mean = s / cnt;
    double var = q / cnt - mean * mean;
    var = var > 0.0 ? var : 0.0;
    stats[2 * idx] = (float)mean;
    stats[2 * idx + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// normalise + affine (+ SiLU) from grid gs to grid gd (same T, H, W).  The reference rounds the GroupNorm output to bf16
// before the activation (two modules), so do we.  Block = one image row (n, t, h): W x C/8 16-byte items, 32-bit index math;
// a chunk of 8 channels lies in one group (C/groups >= 8) or two (C/groups = 4), so two stat pairs per item are enough.
template <int ACT>
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x, VaeGrid gs, bf16_t* __restrict__ y, VaeGrid gd,
                                                       int C, int groups, const float* __restrict__ stats,
                                                       const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta, int N) {
  const int nch = C >> 3;
  const int cg = C / groups;
  const int64_t nrows = (int64_t)N * gs.T * gs.H;
  for (int64_t br = blockIdx.x; br < nrows; br += gridDim.x) {
    const int h = (int)(br % gs.H);
    const int64_t r1 = br / gs.H;
    const int t = (int)(r1 % gs.T), n = (int)(r1 / gs.T);
    const bf16_t* xr = x + grid_row(gs, n, t, h, 0) * C;
    bf16_t* yr = y + grid_row(gd, n, t, h, 0) * C;
    const float* st = stats + 2 * n * groups;
    const int items = gs.W * nch;
    for (int i = threadIdx.x; i < items; i += 256) {
      const int w = i / nch, ch = i - w * nch;
      const uint4 v = *reinterpret_cast<const uint4*>(xr + (int64_t)w * C + ch * 8);
      const uint4 gm = *reinterpret_cast<const uint4*>(gamma + ch * 8);
      const uint4 bt = *reinterpret_cast<const uint4*>(beta + ch * 8);
      const int g0 = (ch * 8) / cg, g1 = (ch * 8 + 4) / cg;
      const float2 s0 = *reinterpret_cast<const float2*>(st + 2 * g0), s1 = *reinterpret_cast<const float2*>(st + 2 * g1);
      float f[8], ga[8], be[8];
      unpack8(v, f); unpack8(gm, ga); unpack8(bt, be);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float mean = e < 4 ? s0.x : s1.x, rstd = e < 4 ? s0.y : s1.y;
        float o = bf2f(f2bf((f[e] - mean) * rstd * ga[e] + be[e]));
        if (ACT) o = silu(o);
        f[e] = o;
      }
      *reinterpret_cast<uint4*>(yr + (int64_t)w * C + ch * 8) = pack8(f);
    }
  }
}

// grid-to-grid copy; (H, W) of the destination are (H << up, W << up) of the source: up = 1 is F.interpolate(nearest, 2x).
// tmode: 0 = same frame; 1 = destination frame t reads source frame t/2 (nearest 2x in time); 2 = CogVideoXUpsample3D with an
// odd frame count (modules/upsampling.py:42-49): frame 0 stays single, frame t >= 1 reads source frame 1 + (t-1)/2.
__global__ __launch_bounds__(256) void regrid_kernel(const bf16_t* __restrict__ x, VaeGrid gs, bf16_t* __restrict__ y, VaeGrid gd,
                                                     int C, int up, int tmode, int N) {
  const int nch = C >> 3;
  const int64_t total = (int64_t)N * gd.T * gd.H * gd.W * nch;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % nch);
    int64_t pos = i / nch;
    const int w = (int)(pos % gd.W); pos /= gd.W;
    const int h = (int)(pos % gd.H); pos /= gd.H;
    const int t = (int)(pos % gd.T);
    const int n = (int)(pos / gd.T);
    const int ts = tmode == 0 ? t : (tmode == 1 ? (t >> 1) : (t == 0 ? 0 : 1 + ((t - 1) >> 1)));
    *reinterpret_cast<uint4*>(y + grid_row(gd, n, t, h, w) * C + ch * 8) =
        *reinterpret_cast<const uint4*>(x + grid_row(gs, n, ts, h >> up, w >> up) * C + ch * 8);
  }
}

// strided pick: destination (t, h, w) = source (t * st + t0, h * ss + s0, w * ss + s0).  A strided convolution is the stride-1
// convolution sampled: diffusers Downsample2D (F.pad(0, 1, 0, 1) + 3 x 3 stride-2 conv, the SD VAE encoder) = the pad-1 stride-1
// conv at (2i + 1, 2j + 1); the causal stride-(2, 1, 1) CausalConv3d of the temporal encoder (one front pad frame,
// autoencoder_kl_open_sora.py:107-118) = the two-front-frames stride-1 conv at frame 2t + 1.
__global__ __launch_bounds__(256) void subsample_kernel(const bf16_t* __restrict__ x, VaeGrid gs, bf16_t* __restrict__ y, VaeGrid gd,
                                                        int C, int st, int ss, int t0, int s0, int N) {
  const int nch = C >> 3;
  const int64_t total = (int64_t)N * gd.T * gd.H * gd.W * nch;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % nch);
    int64_t pos = i / nch;
    const int w = (int)(pos % gd.W); pos /= gd.W;
    const int h = (int)(pos % gd.H); pos /= gd.H;
    const int t = (int)(pos % gd.T);
    const int n = (int)(pos / gd.T);
    *reinterpret_cast<uint4*>(y + grid_row(gd, n, t, h, w) * C + ch * 8) =
        *reinterpret_cast<const uint4*>(x + grid_row(gs, n, t * st + t0, h * ss + s0, w * ss + s0) * C + ch * 8);
  }
}

// "B (C ts) T H W -> B C (T ts) H W", ts = 2 (reference autoencoder_kl_open_sora.py:362-368): source channel 2c + s of frame
// t becomes channel c of frame 2t + s.  Thread = 16 source channels -> 8 channels of each of the two frames.
__global__ __launch_bounds__(256) void d2s_time_kernel(const bf16_t* __restrict__ x, VaeGrid gs, bf16_t* __restrict__ y, VaeGrid gd,
                                                       int Cout, int N) {
  const int nch = Cout >> 3;
  const int64_t total = (int64_t)N * gs.T * gs.H * gs.W * nch;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % nch);
    int64_t pos = i / nch;
    const int w = (int)(pos % gs.W); pos /= gs.W;
    const int h = (int)(pos % gs.H); pos /= gs.H;
    const int t = (int)(pos % gs.T);
    const int n = (int)(pos / gs.T);
    const bf16_t* src = x + grid_row(gs, n, t, h, w) * (2 * Cout) + ch * 16;
    const uint4 a = *reinterpret_cast<const uint4*>(src), b = *reinterpret_cast<const uint4*>(src + 8);
    uint4 e, o;  // even / odd source channels
    e.x = (a.x & 0xffffu) | (a.y << 16); e.y = (a.z & 0xffffu) | (a.w << 16);
    e.z = (b.x & 0xffffu) | (b.y << 16); e.w = (b.z & 0xffffu) | (b.w << 16);
    o.x = (a.x >> 16) | (a.y & 0xffff0000u); o.y = (a.z >> 16) | (a.w & 0xffff0000u);
    o.z = (b.x >> 16) | (b.y & 0xffff0000u); o.w = (b.z >> 16) | (b.w & 0xffff0000u);
    *reinterpret_cast<uint4*>(y + grid_row(gd, n, 2 * t, h, w) * Cout + ch * 8) = e;
    *reinterpret_cast<uint4*>(y + grid_row(gd, n, 2 * t + 1, h, w) * Cout + ch * 8) = o;
  }
}

// First layer of a decoder: planar 4-channel latent z[c][f][h][w] -> per-channel affine (pipeline scale/shift or 1/0.18215)
// -> 1x1(x1) post_quant_conv -> im2col rows [F*H*W, kcols] for the 3x3(x3) conv that follows (column = tap*4 + channel,
// zero outside the volume: spatial zero padding and the causal 2-frame front padding), all roundings to bf16 where the
// reference has a bf16 tensor.
struct FirstParams {
  float scale[4], shift[4], pq_w[16], pq_b[4];
};
__global__ __launch_bounds__(256) void first_im2col_kernel(const bf16_t* __restrict__ z, int F, int H, int W, int kt, int kcols,
                                                           FirstParams fp, bf16_t* __restrict__ out) {
  const int slots = kcols >> 2;
  const int taps = kt * 9;
  const int64_t total = (int64_t)F * H * W * slots;
  const int64_t plane = (int64_t)H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int slot = (int)(i % slots);
    int64_t pos = i / slots;
    const int w = (int)(pos % W); pos /= W;
    const int h = (int)(pos % H);
    const int f = (int)(pos / H);
    uint2 o = make_uint2(0, 0);
    if (slot < taps) {
      const int a = slot / 9, r = slot - a * 9, b = r / 3, c = r - b * 3;
      const int ff = f + a - (kt - 1), hh = h + b - 1, ww = w + c - 1;
      if (ff >= 0 && hh >= 0 && hh < H && ww >= 0 && ww < W) {
        float v[4], u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          v[k] = bf2f(f2bf(bf2f(z[((int64_t)k * F + ff) * plane + (int64_t)hh * W + ww]) * fp.scale[k] + fp.shift[k]));
#pragma unroll
        for (int k = 0; k < 4; ++k)
          u[k] = fp.pq_b[k] + fp.pq_w[4 * k] * v[0] + fp.pq_w[4 * k + 1] * v[1] + fp.pq_w[4 * k + 2] * v[2] + fp.pq_w[4 * k + 3] * v[3];
        o.x = pack2bf(u[0], u[1]);
        o.y = pack2bf(u[2], u[3]);
      }
    }
    *reinterpret_cast<uint2*>(out + (i / slots) * kcols + slot * 4) = o;
  }
}

// last layer: rows [*, ldx] (only the first nc channels are real) -> planar out[c][f0 + t - tskip][h][w]
__global__ __launch_bounds__(256) void extract_planar_kernel(const bf16_t* __restrict__ x, VaeGrid g, int ldx, int nc, int tskip,
                                                             bf16_t* __restrict__ out, int64_t Ftot, int f0, int N) {
  const int64_t total = (int64_t)N * g.T * g.H * g.W;
  const int64_t plane = (int64_t)g.H * g.W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t pos = i;
    const int w = (int)(pos % g.W); pos /= g.W;
    const int h = (int)(pos % g.H); pos /= g.H;
    const int t = (int)(pos % g.T);
    const int n = (int)(pos / g.T);
    const int fr = n * g.T + t;
    if (fr < tskip) continue;
    const bf16_t* src = x + grid_row(g, n, t, h, w) * ldx;
    const uint2 v = *reinterpret_cast<const uint2*>(src);
    const bf16_t c4[4] = {(bf16_t)(v.x & 0xffff), (bf16_t)(v.x >> 16), (bf16_t)(v.y & 0xffff), (bf16_t)(v.y >> 16)};
    for (int c = 0; c < nc; ++c) out[((int64_t)c * Ftot + f0 + fr - tskip) * plane + (int64_t)h * g.W + w] = c4[c];
  }
}

// row softmax over the first n of ld columns: fp32 scores [rows, ld] -> bf16 probabilities [rows, ld], columns n..ld-1 are
// written as 0 (padded keys); one 256-thread block per row, ld <= 256 * 32
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, bf16_t* __restrict__ p, int n, int ld) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const float* sr = s + row * ld;
  bf16_t* pr = p + row * ld;
  const int tid = threadIdx.x;
  float4 v[8];
  float mx = -3.0e38f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = tid * 4 + k * 1024;
    v[k] = make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
    if (c < n) v[k] = *reinterpret_cast<const float4*>(sr + c);
    mx = fmaxf(mx, fmaxf(fmaxf(v[k].x, v[k].y), fmaxf(v[k].z, v[k].w)));
  }
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {  // lanes past n hold -3e38: exp underflows to exactly 0
    v[k].x = __expf(v[k].x - mx); v[k].y = __expf(v[k].y - mx); v[k].z = __expf(v[k].z - mx); v[k].w = __expf(v[k].w - mx);
    sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  }
  sum = wave_sum(sum);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = tid * 4 + k * 1024;
    if (c < ld) {  // c >= n: v holds exp(-3e38 - mx) = 0
      uint2 o;
      o.x = pack2bf(v[k].x * inv, v[k].y * inv);
      o.y = pack2bf(v[k].z * inv, v[k].w * inv);
      *reinterpret_cast<uint2*>(pr + c) = o;
    }
  }
}

// CogVideoXSpatialNorm3D + SiLU (autoencoder_kl_cogvideox.py:165-178, :275-276): y = silu(GN(x) * Y[z] + B[z]) where [Y | B] =
// [conv_y(zq) | conv_b(zq)] were computed at LATENT resolution (1x1x1 convs commute with nearest-neighbour interpolation) as rows
// (zt, zh, zw) of 2C columns, and z = the latent voxel F.interpolate(zq, size=f.shape) maps (t, h, w) to — with the reference's
// first-frame split when the frame count is odd.  Every intermediate is rounded to bf16 like the reference's tensors.
__global__ __launch_bounds__(256) void spatial_norm_apply_kernel(const bf16_t* __restrict__ x, VaeGrid gs, bf16_t* __restrict__ y, VaeGrid gd,
                                                                 int C, int groups, const float* __restrict__ stats,
                                                                 const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta,
                                                                 const bf16_t* __restrict__ yb, int zT, int zH, int zW, int N) {
  const int nch = C >> 3;
  const int cg = C / groups;
  const int T = gs.T;
  const bool split = T > 1 && (T & 1);
  const int64_t nrows = (int64_t)N * T * gs.H;
  for (int64_t br = blockIdx.x; br < nrows; br += gridDim.x) {
    const int h = (int)(br % gs.H);
    const int64_t r1 = br / gs.H;
    const int t = (int)(r1 % T), n = (int)(r1 / T);
    const int zt = split ? (t == 0 ? 0 : 1 + (int)(((int64_t)(t - 1) * (zT - 1)) / (T - 1))) : (int)(((int64_t)t * zT) / T);
    const int zh = (int)(((int64_t)h * zH) / gs.H);
    const bf16_t* xr = x + grid_row(gs, n, t, h, 0) * C;
    bf16_t* yr = y + grid_row(gd, n, t, h, 0) * C;
    const bf16_t* zr = yb + (((int64_t)n * zT + zt) * zH + zh) * zW * (2 * C);
    const float* st = stats + 2 * n * groups;
    const int items = gs.W * nch;
    for (int i = threadIdx.x; i < items; i += 256) {
      const int w = i / nch, ch = i - w * nch;
      const int zw = (int)(((int64_t)w * zW) / gs.W);
      const uint4 v = *reinterpret_cast<const uint4*>(xr + (int64_t)w * C + ch * 8);
      const uint4 gm = *reinterpret_cast<const uint4*>(gamma + ch * 8);
      const uint4 bt = *reinterpret_cast<const uint4*>(beta + ch * 8);
      const uint4 yy = *reinterpret_cast<const uint4*>(zr + (int64_t)zw * 2 * C + ch * 8);
      const uint4 bb = *reinterpret_cast<const uint4*>(zr + (int64_t)zw * 2 * C + C + ch * 8);
      const int g0 = (ch * 8) / cg, g1 = (ch * 8 + 4) / cg;
      const float2 s0 = *reinterpret_cast<const float2*>(st + 2 * g0), s1 = *reinterpret_cast<const float2*>(st + 2 * g1);
      float f[8], ga[8], be[8], fy[8], fb[8];
      unpack8(v, f); unpack8(gm, ga); unpack8(bt, be); unpack8(yy, fy); unpack8(bb, fb);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float mean = e < 4 ? s0.x : s1.x, rstd = e < 4 ? s0.y : s1.y;
        const float nf = bf2f(f2bf((f[e] - mean) * rstd * ga[e] + be[e]));
        const float q = bf2f(f2bf(bf2f(f2bf(nf * fy[e])) + fb[e]));
        f[e] = silu(q);
      }
      *reinterpret_cast<uint4*>(yr + (int64_t)w * C + ch * 8) = pack8(f);
    }
  }
}

// tiled_decode's blend_v / blend_h (autoencoder_kl_cogvideox.py:1145-1159) on planar bf16 tiles [outer, H, W]:
// b[o, y, x] = bf16(bf16(a[o, Ha - ext + y, x] * (1 - y/ext)) + bf16(b[o, y, x] * (y/ext))) for y < ext (axis 0; axis 1: columns).
__global__ __launch_bounds__(256) void blend_edge_kernel(const bf16_t* __restrict__ a, bf16_t* __restrict__ b, int64_t outer, int Ha,
                                                         int Wa, int Hb, int Wb, int ext, int axis) {
  const int64_t per = axis == 0 ? (int64_t)ext * Wb : (int64_t)Hb * ext;
  const int64_t total = outer * per;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t o = i / per;
    const int64_t r = i - o * per;
    int yy, xx, k;
    int64_t ia;
    if (axis == 0) {
      yy = (int)(r / Wb); xx = (int)(r - (int64_t)yy * Wb); k = yy;
      ia = (o * Ha + (Ha - ext + yy)) * Wa + xx;
    } else {
      yy = (int)(r / ext); xx = (int)(r - (int64_t)yy * ext); k = xx;
      ia = (o * Ha + yy) * Wa + (Wa - ext + xx);
    }
    const int64_t ib = (o * Hb + yy) * Wb + xx;
    const float w1 = (float)(1.0 - (double)k / (double)ext), w2 = (float)((double)k / (double)ext);
    b[ib] = f2bf(bf2f(f2bf(bf2f(a[ia]) * w1)) + bf2f(f2bf(bf2f(b[ib]) * w2)));
  }
}

inline unsigned grid_for(int64_t work_items) {
  int64_t b = (work_items + 255) / 256;
  const int64_t cap = 256 * 32;  // grid-stride: 32 blocks per CU are plenty for streaming kernels
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

bool grid_ok(const VaeGrid& g) { return g.T > 0 && g.H > 0 && g.W > 0 && g.pad >= 0 && g.pad <= 1 && g.tf >= 0; }

}  // namespace

int launch_gn_stats(const bf16_t* x, const VaeGrid& g, int N, int C, int groups, float eps, float* partial, int nblk, float* stats,
                    hipStream_t stream) {
  if (N <= 0) return 0;
  if (!grid_ok(g) || C % 8 != 0 || C > 2048 || groups <= 0 || C % groups != 0 || (C / groups) % 4 != 0 || nblk <= 0 || N > 65535)
    return VSYS_ERR_SHAPE;
  hipLaunchKernelGGL(gn_partial_kernel, dim3(nblk, N), dim3(256), 0, stream, x, g, C, partial);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(N * groups), dim3(64), 0, stream, partial, nblk, C, groups,
                     (int64_t)g.T * g.H * g.W, eps, stats);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_gn_apply(const bf16_t* x, const VaeGrid& gs, bf16_t* y, const VaeGrid& gd, int N, int C, int groups, const float* stats,
                    const bf16_t* gamma, const bf16_t* beta, int act, hipStream_t stream) {
  if (N <= 0) return 0;
  if (!grid_ok(gs) || !grid_ok(gd) || gs.T != gd.T || gs.H != gd.H || gs.W != gd.W || C % 8 != 0 || groups <= 0 || C % groups != 0)
    return VSYS_ERR_SHAPE;
  if ((C / groups) % 4 != 0) return VSYS_ERR_SHAPE;
  const int64_t nrows = (int64_t)N * gs.T * gs.H;
  const unsigned grid = (unsigned)(nrows < 65536 ? nrows : 65536);
  if (act == ACT_SILU) hipLaunchKernelGGL(gn_apply_kernel<1>, dim3(grid), dim3(256), 0, stream, x, gs, y, gd, C, groups, stats, gamma, beta, N);
  else if (act == ACT_NONE) hipLaunchKernelGGL(gn_apply_kernel<0>, dim3(grid), dim3(256), 0, stream, x, gs, y, gd, C, groups, stats, gamma, beta, N);
  else return VSYS_ERR_ARG;
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_regrid(const bf16_t* x, const VaeGrid& gs, bf16_t* y, const VaeGrid& gd, int N, int C, int up, int tmode, hipStream_t stream) {
  if (N <= 0) return 0;
  const int want_t = tmode == 0 ? gs.T : (tmode == 1 ? 2 * gs.T : 2 * gs.T - 1);
  if (!grid_ok(gs) || !grid_ok(gd) || up < 0 || up > 1 || tmode < 0 || tmode > 2 || gd.T != want_t || (gs.H << up) != gd.H ||
      (gs.W << up) != gd.W || C % 8 != 0)
    return VSYS_ERR_SHAPE;
  hipLaunchKernelGGL(regrid_kernel, dim3(grid_for((int64_t)N * gd.T * gd.H * gd.W * (C >> 3))), dim3(256), 0, stream, x, gs, y, gd, C, up,
                     tmode, N);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_subsample(const bf16_t* x, const VaeGrid& gs, bf16_t* y, const VaeGrid& gd, int N, int C, int st, int ss, int t0, int s0,
                     hipStream_t stream) {
  if (N <= 0) return 0;
  if (!grid_ok(gs) || !grid_ok(gd) || C % 8 != 0 || st < 1 || ss < 1 || t0 < 0 || s0 < 0 || gd.T <= 0 || gd.H <= 0 || gd.W <= 0 ||
      (gd.T - 1) * st + t0 >= gs.T || (gd.H - 1) * ss + s0 >= gs.H || (gd.W - 1) * ss + s0 >= gs.W)
    return VSYS_ERR_SHAPE;
  hipLaunchKernelGGL(subsample_kernel, dim3(grid_for((int64_t)N * gd.T * gd.H * gd.W * (C >> 3))), dim3(256), 0, stream, x, gs, y, gd, C,
                     st, ss, t0, s0, N);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_spatial_norm_apply(const bf16_t* x, const VaeGrid& gs, bf16_t* y, const VaeGrid& gd, int N, int C, int groups,
                              const float* stats, const bf16_t* gamma, const bf16_t* beta, const bf16_t* yb, int zT, int zH, int zW,
                              hipStream_t stream) {
  if (N <= 0) return 0;
  if (!grid_ok(gs) || !grid_ok(gd) || gs.T != gd.T || gs.H != gd.H || gs.W != gd.W || C % 8 != 0 || groups <= 0 || C % groups != 0 ||
      (C / groups) % 4 != 0 || zT <= 0 || zH <= 0 || zW <= 0)
    return VSYS_ERR_SHAPE;
  const int64_t nrows = (int64_t)N * gs.T * gs.H;
  hipLaunchKernelGGL(spatial_norm_apply_kernel, dim3((unsigned)(nrows < 65536 ? nrows : 65536)), dim3(256), 0, stream, x, gs, y, gd, C,
                     groups, stats, gamma, beta, yb, zT, zH, zW, N);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_blend_edge(const bf16_t* a, bf16_t* b, int64_t outer, int Ha, int Wa, int Hb, int Wb, int ext, int axis, hipStream_t stream) {
  if (outer <= 0 || ext <= 0) return 0;
  if (axis < 0 || axis > 1 || Ha <= 0 || Wa <= 0 || Hb <= 0 || Wb <= 0) return VSYS_ERR_SHAPE;
  if (axis == 0 ? (ext > Ha || ext > Hb || Wa != Wb) : (ext > Wa || ext > Wb || Ha != Hb)) return VSYS_ERR_SHAPE;
  const int64_t total = outer * (axis == 0 ? (int64_t)ext * Wb : (int64_t)Hb * ext);
  hipLaunchKernelGGL(blend_edge_kernel, dim3(grid_for(total)), dim3(256), 0, stream, a, b, outer, Ha, Wa, Hb, Wb, ext, axis);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_d2s_time(const bf16_t* x, const VaeGrid& gs, bf16_t* y, const VaeGrid& gd, int N, int Cout, hipStream_t stream) {
  if (N <= 0) return 0;
  if (!grid_ok(gs) || !grid_ok(gd) || gd.T != 2 * gs.T || gs.H != gd.H || gs.W != gd.W || Cout % 8 != 0) return VSYS_ERR_SHAPE;
  hipLaunchKernelGGL(d2s_time_kernel, dim3(grid_for((int64_t)N * gs.T * gs.H * gs.W * (Cout >> 3))), dim3(256), 0, stream, x, gs, y, gd, Cout, N);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_vae_first_im2col(const bf16_t* z, int F, int H, int W, int kt, int kcols, const float* scale, const float* shift,
                            const float* pq_w, const float* pq_b, bf16_t* out, hipStream_t stream) {
  if (F <= 0) return 0;
  if (H <= 0 || W <= 0 || (kt != 1 && kt != 3) || kcols % 4 != 0 || kcols < kt * 36) return VSYS_ERR_SHAPE;
  FirstParams fp;
  for (int i = 0; i < 4; ++i) { fp.scale[i] = scale[i]; fp.shift[i] = shift[i]; fp.pq_b[i] = pq_b[i]; }
  for (int i = 0; i < 16; ++i) fp.pq_w[i] = pq_w[i];
  hipLaunchKernelGGL(first_im2col_kernel, dim3(grid_for((int64_t)F * H * W * (kcols >> 2))), dim3(256), 0, stream, z, F, H, W, kt, kcols, fp, out);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_extract_planar(const bf16_t* x, const VaeGrid& g, int N, int ldx, int nc, int tskip, bf16_t* out, int64_t Ftot, int f0,
                          hipStream_t stream) {
  if (N <= 0) return 0;
  if (!grid_ok(g) || nc < 1 || nc > 4 || ldx % 4 != 0 || tskip < 0) return VSYS_ERR_SHAPE;
  hipLaunchKernelGGL(extract_planar_kernel, dim3(grid_for((int64_t)N * g.T * g.H * g.W)), dim3(256), 0, stream, x, g, ldx, nc, tskip, out, Ftot, f0, N);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_softmax_rows(const float* s, bf16_t* p, int64_t rows, int n, int ld, hipStream_t stream) {
  if (rows <= 0) return 0;
  if (n <= 0 || n % 4 != 0 || ld < n || ld % 4 != 0 || ld > 8192 || rows > 0x7fffffff) return VSYS_ERR_SHAPE;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, stream, s, p, n, ld);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

}  // namespace vsys
