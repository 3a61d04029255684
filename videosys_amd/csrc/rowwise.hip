// HBM-bound row-wise / element-wise kernels of the STDiT3 denoise step (gfx950).
//
// Reference call sites replaced (/root/reference/videosys):
//   adaln_modulate      nn.LayerNorm(eps 1e-6, no affine) + t2i_modulate      open_sora_transformer_3d.py:47-48,117,196-197,260-261
//   mod_table           scale_shift_table[None] + t.reshape(B,6,C)            open_sora_transformer_3d.py:177-179 (all blocks at once)
//   timestep_embedding  TimestepEmbedder.timestep_embedding                    modules/embeddings.py:123-141
//   patch_embed         OpenSoraPatchEmbed3D (Conv3d k=s=(1,2,2)) + pos_emb    modules/embeddings.py:85-104; open_sora_transformer_3d.py:593-595
//   final_layer         T2IFinalLayer + unpatchify + fp32 cast                 open_sora_transformer_3d.py:75-87,622-630,634-658
//   cfg_euler_step      RFLOW.sample CFG combine + Euler update                schedulers/scheduling_rflow_open_sora.py:245-252
//   add_rows            x = x + last_attn / last_cross (PAB broadcast step)    open_sora_transformer_3d.py:192-193,228,234-235
//   copy_4d             DSP all-to-all pack/unpack (tensor_split/.contiguous/cat/pad/narrow)  core/distributed/comm.py:104-108,282-304
//
// All are one pass over their operands with 16-byte per-lane accesses; the algorithmic HBM bytes of each are the
// mandatory tensor reads + writes.
#include "common.h"

#include <cstdlib>
#include "vsys_internal.h"

namespace vsys {
namespace {

// ---------------------------------------------------------------------------------------------------------
// LayerNorm (no affine) + modulate: y = LN(x) * (1 + scale[b]) + shift[b].  One wave per row; the row lives in
// registers (C <= 64*8*MAXV elements), so x is read exactly once.  fp32 statistics (two-pass in registers).
// ---------------------------------------------------------------------------------------------------------
template <int MAXV>
__global__ __launch_bounds__(256) void adaln_modulate_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ shift,
                                                             const bf16_t* __restrict__ scale, bf16_t* __restrict__ y,
                                                             int64_t rows, int C, int64_t rows_per_sample,
                                                             int64_t mod_stride, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunk = C >> 3;
  const bf16_t* xr = x + row * C;
  float v[MAXV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      uint4 u = *reinterpret_cast<const uint4*>(xr + c * 8);
      unpack8(u, v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float d = v[i][e] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  const int64_t b = row / rows_per_sample;
  const bf16_t* sh = shift + b * mod_stride;
  const bf16_t* sc = scale + b * mod_stride;
  bf16_t* yr = y + row * C;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      float a[8], m[8], o[8];
      unpack8(*reinterpret_cast<const uint4*>(sh + c * 8), a);
      unpack8(*reinterpret_cast<const uint4*>(sc + c * 8), m);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * (1.0f + m[e]) + a[e];
      *reinterpret_cast<uint4*>(yr + c * 8) = pack8(o);
    }
  }
}

// RPW rows per wave (all of one sample): the row loads of a wave are in flight together and the shift / scale vectors of the
// sample are fetched once per wave instead of once per row (they are 2 x the bytes of a row, from L2).  C <= 1536.
template <int RPW>
__global__ __launch_bounds__(256) void adaln_modulate_rows_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ shift,
                                                                  const bf16_t* __restrict__ scale, bf16_t* __restrict__ y,
                                                                  int64_t rows, int C, int64_t rows_per_sample,
                                                                  int64_t mod_stride, float eps) {
  constexpr int MAXV = 3;
  const int lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
  if (row0 >= rows) return;
  const int nchunk = C >> 3;
  uint4 u[RPW][MAXV];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int64_t row = row0 + r < rows ? row0 + r : rows - 1;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      u[r][i] = c < nchunk ? *reinterpret_cast<const uint4*>(x + row * C + c * 8) : make_uint4(0, 0, 0, 0);
    }
  }
  const int64_t b = row0 / rows_per_sample;   // the launcher guarantees rows_per_sample % RPW == 0
  uint4 ush[MAXV], usc[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    ush[i] = c < nchunk ? *reinterpret_cast<const uint4*>(shift + b * mod_stride + c * 8) : make_uint4(0, 0, 0, 0);
    usc[i] = c < nchunk ? *reinterpret_cast<const uint4*>(scale + b * mod_stride + c * 8) : make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    float v[MAXV][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      unpack8(u[r][i], v[i]);
      if (lane + 64 * i < nchunk) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[i][e];
      }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      if (lane + 64 * i < nchunk) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float d = v[i][e] - mean;
          q += d * d;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    if (row0 + r < rows) {
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
          float a[8], m[8], o[8];
          unpack8(ush[i], a);
          unpack8(usc[i], m);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * (1.0f + m[e]) + a[e];
          *reinterpret_cast<uint4*>(y + (row0 + r) * C + c * 8) = pack8(o);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// General LayerNorm + modulate for the CogVideoX blocks (modules/normalization.py:36-60 CogVideoXLayerNormZero, :62-114
// AdaLayerNorm, nn.LayerNorm norm_final): y = (LN(x) * w + b) * (1 + scale) + shift with optional affine (w, b), optional
// modulation (shift/scale NULL = plain LayerNorm) and TWO row segments per sample: rows whose position inside the sample
// is < seg_split (the text tokens of the joint [text | video] sequence) read their shift/scale mod_alt elements further on.
// ---------------------------------------------------------------------------------------------------------
template <int MAXV>
__global__ __launch_bounds__(256) void ln_modulate_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ ln_w,
                                                          const bf16_t* __restrict__ ln_b, const bf16_t* __restrict__ shift,
                                                          const bf16_t* __restrict__ scale, bf16_t* __restrict__ y, int64_t rows,
                                                          int C, int64_t rows_per_sample, int64_t mod_stride, int64_t seg_split,
                                                          int64_t mod_alt, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunk = C >> 3;
  const bf16_t* xr = x + row * C;
  float v[MAXV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      unpack8(*reinterpret_cast<const uint4*>(xr + c * 8), v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[i][e] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  const int64_t b = row / rows_per_sample;
  const int64_t alt = (seg_split > 0 && row - b * rows_per_sample < seg_split) ? mod_alt : 0;
  bf16_t* yr = y + row * C;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd;
      if (ln_w != nullptr) {
        float w[8], bb[8];
        unpack8(*reinterpret_cast<const uint4*>(ln_w + c * 8), w);
        unpack8(*reinterpret_cast<const uint4*>(ln_b + c * 8), bb);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = o[e] * w[e] + bb[e];
      }
      if (shift != nullptr) {
        float a[8], m[8];
        unpack8(*reinterpret_cast<const uint4*>(shift + b * mod_stride + alt + c * 8), a);
        unpack8(*reinterpret_cast<const uint4*>(scale + b * mod_stride + alt + c * 8), m);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = o[e] * (1.0f + m[e]) + a[e];
      }
      *reinterpret_cast<uint4*>(yr + c * 8) = pack8(o);
    }
  }
}

// x[r] += gate[sample(r)] * y[r]   (CogVideoX PAB: the cached un-gated attention output is re-gated every step,
// cogvideox_transformer_3d.py:288-289); two gate segments per sample like ln_modulate.
__global__ void gate_add_rows_kernel(bf16_t* __restrict__ x, const bf16_t* __restrict__ y, const bf16_t* __restrict__ gate,
                                     int64_t rows, int C8, int64_t rows_per_sample, int64_t gate_stride, int64_t seg_split,
                                     int64_t gate_alt) {
  const int64_t total = rows * C8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / C8;
    const int c = (int)(i - r * C8);
    const int64_t b = r / rows_per_sample;
    const int64_t alt = (seg_split > 0 && r - b * rows_per_sample < seg_split) ? gate_alt : 0;
    float a[8], v[8], g[8];
    unpack8(reinterpret_cast<const uint4*>(x)[i], a);
    unpack8(reinterpret_cast<const uint4*>(y)[i], v);
    unpack8(*reinterpret_cast<const uint4*>(gate + b * gate_stride + alt + c * 8), g);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] += bf2f(f2bf(g[k] * v[k]));
    reinterpret_cast<uint4*>(x)[i] = pack8(a);
  }
}

// CogVideoXPatchEmbed's Conv2d(k = s = p) as a GEMM operand (modules/embeddings.py:14-51): out[(b, f, hp, wp)][(c, dy, dx)]
// = bf16(z[b % Bz][f][c][hp*p + dy][wp*p + dx]), z fp32 [Bz, F, Cin, H, W]; the weight [C, Cin, p, p] flattens to the same
// (c, dy, dx) order.  K = Cin*p*p must be a multiple of 8 (64 for CogVideoX).
__global__ void im2col_patch_kernel(const float* __restrict__ z, int Bz, bf16_t* __restrict__ out, int B, int F, int Cin, int H,
                                    int W, int p) {
  const int Hp = H / p, Wp = W / p, K = Cin * p * p;
  const int64_t total = (int64_t)B * F * Hp * Wp * K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % K);
    const int64_t tok = i / K;
    const int wp = (int)(tok % Wp), hp = (int)((tok / Wp) % Hp), f = (int)((tok / ((int64_t)Wp * Hp)) % F);
    const int b = (int)(tok / ((int64_t)Wp * Hp * F));
    const int dx = k % p, dy = (k / p) % p, c = k / (p * p);
    out[i] = f2bf(z[((((int64_t)(b % Bz) * F + f) * Cin + c) * H + hp * p + dy) * W + wp * p + dx]);
  }
}

// CogVideoX unpatchify (cogvideox_transformer_3d.py:581-583): x [(b, f, hp, wp)][>= Cout*p*p] bf16, channel order (c, dy, dx)
// -> out fp32 [B, F, Cout, Hp*p, Wp*p]
__global__ void unpatchify_cvx_kernel(const bf16_t* __restrict__ x, int64_t ldx, float* __restrict__ out, int B, int F, int Hp,
                                      int Wp, int Cout, int p) {
  const int H = Hp * p, W = Wp * p;
  const int64_t total = (int64_t)B * F * Cout * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int w = (int)(i % W), h = (int)((i / W) % H), c = (int)((i / ((int64_t)W * H)) % Cout);
    const int64_t bf = i / ((int64_t)W * H * Cout);
    const int64_t tok = (bf * Hp + h / p) * Wp + w / p;
    out[i] = bf2f(x[tok * ldx + (c * p + h % p) * p + w % p]);
  }
}

// mod[blk][b][6][C] = bf16(table[blk][6][C] + t_mlp[b][6*C])   (bf16 add, as the reference's bf16 tensors do)
__global__ void mod_table_kernel(const bf16_t* __restrict__ table, const bf16_t* __restrict__ t_mlp, bf16_t* __restrict__ out,
                                 int nblk, int B, int C6) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)nblk * B * C6;
  if (i >= total) return;
  const int c = (int)(i % C6);
  const int b = (int)((i / C6) % B);
  const int k = (int)(i / ((int64_t)C6 * B));
  out[i] = f2bf(bf2f(table[(int64_t)k * C6 + c]) + bf2f(t_mlp[(int64_t)b * C6 + c]));
}

// out[b][0:half] = cos(t*f_i), out[b][half:] = sin(t*f_i), f_i = exp(-ln(10000) i/half)   (dim even)
__global__ void timestep_embedding_kernel(const float* __restrict__ t, bf16_t* __restrict__ out, int B, int dim) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = dim / 2;
  if (i >= B * half) return;
  const int b = i / half, k = i - b * half;
  const float f = expf(-9.210340371976184f * (float)k / (float)half);
  const float a = t[b] * f;
  out[(int64_t)b * dim + k] = f2bf(cosf(a));
  out[(int64_t)b * dim + half + k] = f2bf(sinf(a));
}

// Patch embed: fp32 latent z [Bz, Cin, T, H, W] (sample b reads z[b % Bz]: the CFG duplicate torch.cat([z, z]) of
// scheduling_rflow_open_sora.py:239 is never materialised; x.to(bf16) of open_sora_transformer_3d.py:561 happens at
// the load) -> out [B, T, Hp*Wp, C] = conv(k=s=(1,ph,pw)) + bias + pos[s][C].
// K = Cin*ph*pw (16 for STDiT3) is far too small for MFMA: each thread produces 8 output channels of one token
// from K inputs (broadcast within the token) and K*8 weights; the store side (N*C bf16) is the HBM term.
__global__ __launch_bounds__(256) void patch_embed_kernel(const float* __restrict__ x, int Bz, const bf16_t* __restrict__ w,
                                                          const bf16_t* __restrict__ bias, const bf16_t* __restrict__ pos,
                                                          bf16_t* __restrict__ out, int B, int Cin, int T, int H, int W,
                                                          int ph, int pw, int C, int s0, int Sl) {
  // out rows are the tokens s0 .. s0 + Sl - 1 of every (b, t) (the whole frame: s0 = 0, Sl = S); tokens past S are zero rows
  // (the zero padding of split_sequence, comm.py:148-167)
  const int Hp = (H + ph - 1) / ph, Wp = (W + pw - 1) / pw;
  const int S = Hp * Wp;
  const int cchunks = C >> 3;
  const int64_t total = (int64_t)B * T * Sl * cchunks;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int cc = (int)(i % cchunks);
  const int64_t tok = i / cchunks;
  const int s = s0 + (int)(tok % Sl);
  const int t = (int)((tok / Sl) % T);
  const int b = (int)(tok / ((int64_t)Sl * T));
  const int bz = b % Bz;
  if (s >= S) {
    *reinterpret_cast<uint4*>(out + tok * C + cc * 8) = make_uint4(0, 0, 0, 0);
    return;
  }
  const int hp = s / Wp, wp = s - hp * Wp;
  float acc[8];
  {
    float bb[8];
    unpack8(*reinterpret_cast<const uint4*>(bias + cc * 8), bb);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = bb[e];
  }
  const int K = Cin * ph * pw;
  if (K == 16) {
    // STDiT3 geometry (Cin 4, patch 2x2): gather the 16 inputs once, then two 16-byte weight loads per output channel
    float xin[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int ci = k >> 2, dy = (k >> 1) & 1, dx = k & 1;
      const int hh = hp * 2 + dy, ww = wp * 2 + dx;
      xin[k] = 0.f;
      if (ph == 2 && pw == 2 && hh < H && ww < W) xin[k] = bf2f(f2bf(x[((((int64_t)bz * Cin + ci) * T + t) * H + hh) * W + ww]));
    }
    if (ph == 2 && pw == 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bf16_t* wr = w + (int64_t)(cc * 8 + e) * 16;
        float wa[8], wb[8];
        unpack8(*reinterpret_cast<const uint4*>(wr), wa);
        unpack8(*reinterpret_cast<const uint4*>(wr + 8), wb);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[e] += xin[k] * wa[k] + xin[k + 8] * wb[k];
      }
    }
  }
  if (!(K == 16 && ph == 2 && pw == 2)) {
    for (int ci = 0; ci < Cin; ++ci)
      for (int dy = 0; dy < ph; ++dy)
        for (int dx = 0; dx < pw; ++dx) {
          const int hh = hp * ph + dy, ww = wp * pw + dx;
          float xv = 0.f;  // zero padding of odd H/W (F.pad in the reference)
          if (hh < H && ww < W) xv = bf2f(f2bf(x[((((int64_t)bz * Cin + ci) * T + t) * H + hh) * W + ww]));
          const int k = (ci * ph + dy) * pw + dx;
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += xv * bf2f(w[(int64_t)(cc * 8 + e) * K + k]);
        }
  }
  float pp[8];
  unpack8(*reinterpret_cast<const uint4*>(pos + (int64_t)s * C + cc * 8), pp);
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = bf2f(f2bf(acc[e])) + pp[e];  // conv output is a bf16 tensor before "+ pos_emb"
  *reinterpret_cast<uint4*>(out + tok * C + cc * 8) = pack8(acc);
}

// Final layer: y = Linear(LN(x)*(1+scale_b)+shift_b) with (shift,scale) = table[2,C] + t[b,C]; then unpatchify into
// out[B, Cout, T, H, W] fp32 (cropped to the un-padded latent size).  One wave per token row; NOUT = ph*pw*Cout <= 64.
template <int MAXV>
__global__ __launch_bounds__(256) void final_layer_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ table,
                                                          const bf16_t* __restrict__ tvec, const bf16_t* __restrict__ w,
                                                          const bf16_t* __restrict__ bias, float* __restrict__ out,
                                                          int B, int T, int Hp, int Wp, int H, int W, int ph, int pw,
                                                          int Cout, int C, float eps, int Sl, float* __restrict__ tokens) {
  // x rows: [B, T, Sl, C] (Sl = S for the whole sequence).  tokens != null: the NOUT values of a row go to tokens[row][NOUT]
  // (the S-shard of a sequence-parallel rank; vsys_unpatchify_tokens scatters the gathered rows) instead of to the pixels.
  const int lane = threadIdx.x & 63;
  const int S = Sl;
  const int64_t rows = (int64_t)B * T * S;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunk = C >> 3;
  const bf16_t* xr = x + row * C;
  float v[MAXV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      unpack8(*reinterpret_cast<const uint4*>(xr + c * 8), v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float d = v[i][e] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  const int b = (int)(row / ((int64_t)T * S));
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      float sh[8], sc[8], tt[8];
      unpack8(*reinterpret_cast<const uint4*>(table + c * 8), sh);
      unpack8(*reinterpret_cast<const uint4*>(table + C + c * 8), sc);
      unpack8(*reinterpret_cast<const uint4*>(tvec + (int64_t)b * C + c * 8), tt);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float shift = bf2f(f2bf(sh[e] + tt[e])), scale = bf2f(f2bf(sc[e] + tt[e]));
        v[i][e] = bf2f(f2bf((v[i][e] - mean) * rstd * (1.0f + scale) + shift));  // bf16 activation into the Linear
      }
    }
  }
  const int NOUT = ph * pw * Cout;
  const int64_t tok = row % ((int64_t)T * S);
  const int t = (int)(tok / S), sidx = (int)(tok % S);
  const int hp = sidx / Wp, wp = sidx - hp * Wp;
  for (int n = 0; n < NOUT; ++n) {
    const bf16_t* wr = w + (int64_t)n * C;
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        float ww[8];
        unpack8(*reinterpret_cast<const uint4*>(wr + c * 8), ww);
#pragma unroll
        for (int e = 0; e < 8; ++e) d += v[i][e] * ww[e];
      }
    }
    d = wave_sum(d);
    if (lane == 0) {
      d = bf2f(f2bf(d + bf2f(bias[n])));  // Linear output is bf16, then .to(float32)
      if (tokens != nullptr) {
        tokens[row * NOUT + n] = d;
        continue;
      }
      // "(T_p H_p W_p C_out)" ordering of the channel axis, T_p = 1
      const int co = n % Cout;
      const int dx = (n / Cout) % pw, dy = n / (Cout * pw);
      const int hh = hp * ph + dy, ww2 = wp * pw + dx;
      if (hh < H && ww2 < W) out[((((int64_t)b * Cout + co) * T + t) * H + hh) * W + ww2] = d;
    }
  }
}

// unpatchify of gathered token rows: tok [P][B][T][Sl][NOUT] fp32 (rank r holds tokens r*Sl .. of every (b, t); tokens >= Hp*Wp
// are padding) -> out[B, Cout, T, H, W] fp32, "(T_p H_p W_p C_out)" channel order, cropped (open_sora_transformer_3d.py:634-658)
__global__ void unpatchify_tokens_kernel(const float* __restrict__ tok, float* __restrict__ out, int P, int B, int T, int Sl,
                                         int Hp, int Wp, int H, int W, int ph, int pw, int Cout) {
  const int NOUT = ph * pw * Cout;
  const int64_t total = (int64_t)B * Cout * T * H * W;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ww = (int)(i % W), hh = (int)((i / W) % H), t = (int)((i / ((int64_t)W * H)) % T);
  const int co = (int)((i / ((int64_t)W * H * T)) % Cout), b = (int)(i / ((int64_t)W * H * T * Cout));
  const int hp = hh / ph, dy = hh - hp * ph, wp = ww / pw, dx = ww - wp * pw;
  const int s = hp * Wp + wp;
  const int r = s / Sl, j = s - r * Sl;
  const int n = (dy * pw + dx) * Cout + co;
  out[i] = tok[((((int64_t)r * B + b) * T + t) * Sl + j) * NOUT + n];
}

// RFLOW step: pred = model_out[:, :Cin] ; v = uncond + g*(cond - uncond) ; z += v*dt    (cond = batch half 0)
__global__ void cfg_euler_kernel(float* __restrict__ z, const float* __restrict__ model_out, int Bz, int Cin, int Cout,
                                 int64_t thw, float guidance, float dt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)Bz * Cin * thw;
  if (i >= total) return;
  const int64_t sp = i % thw;
  const int c = (int)((i / thw) % Cin);
  const int b = (int)(i / (thw * Cin));
  const float cond = model_out[((int64_t)b * Cout + c) * thw + sp];
  const float unc = model_out[((int64_t)(b + Bz) * Cout + c) * thw + sp];
  z[i] = z[i] + (unc + guidance * (cond - unc)) * dt;
}

// Generic CFG + linear scheduler step (DDIM eta = 0 and friends): eps = uncond + g (cond - uncond) on the first Cin of
// Cout channels, z = c_z z + c_eps eps.  cond_first selects which batch half of model_out holds the conditional
// prediction (RFLOW: first; Latte / CogVideoX pipelines put the negative prompt first).
__global__ void cfg_axpby_kernel(float* __restrict__ z, const float* __restrict__ model_out, int Bz, int Cin, int Cout,
                                 int64_t thw, float guidance, float c_z, float c_eps, int cond_first) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)Bz * Cin * thw;
  if (i >= total) return;
  const int64_t sp = i % thw;
  const int c = (int)((i / thw) % Cin);
  const int b = (int)(i / (thw * Cin));
  const float h0 = model_out[((int64_t)b * Cout + c) * thw + sp];
  const float h1 = model_out[((int64_t)(b + Bz) * Cout + c) * thw + sp];
  const float cond = cond_first ? h0 : h1, unc = cond_first ? h1 : h0;
  z[i] = c_z * z[i] + c_eps * (unc + guidance * (cond - unc));
}

// x[r][:] += e[(r / group) % period][:]  (Latte temporal position embedding: rows ordered (b, f, s), group = S, period = F)
__global__ void add_bcast_rows_kernel(bf16_t* __restrict__ x, const bf16_t* __restrict__ e, int64_t rows, int C8, int64_t group,
                                      int64_t period) {
  const int64_t total = rows * C8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / C8;
    const int c = (int)(i - r * C8);
    const int64_t er = (r / group) % period;
    float a[8], b[8];
    unpack8(reinterpret_cast<const uint4*>(x)[i], a);
    unpack8(reinterpret_cast<const uint4*>(e)[er * C8 + c], b);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] += b[k];
    reinterpret_cast<uint4*>(x)[i] = pack8(a);
  }
}

// x[i] = bf16(x[i] + y[i]) over n8 16-byte chunks
__global__ void add_rows_kernel(bf16_t* __restrict__ x, const bf16_t* __restrict__ y, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float a[8], b[8];
    unpack8(reinterpret_cast<const uint4*>(x)[i], a);
    unpack8(reinterpret_cast<const uint4*>(y)[i], b);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += b[e];
    reinterpret_cast<uint4*>(x)[i] = pack8(a);
  }
}

// Generic 4-D strided copy of rows of C bf16 (C % 8 == 0): dst[i0][i1][i2][:] = src[i0][i1][i2][:] or zero when
// (i1 >= n1_valid || i2 >= n2_valid) — the zero-pad of all_to_all_with_pad.  Strides in elements.
__global__ void copy_4d_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int n0, int n1, int n2, int C,
                               int64_t ss0, int64_t ss1, int64_t ss2, int64_t ds0, int64_t ds1, int64_t ds2,
                               int n1_valid, int n2_valid) {
  const int cch = C >> 3;
  const int64_t total = (int64_t)n0 * n1 * n2 * cch;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cch);
    int64_t r = i / cch;
    const int i2 = (int)(r % n2);
    r /= n2;
    const int i1 = (int)(r % n1);
    const int i0 = (int)(r / n1);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (i1 < n1_valid && i2 < n2_valid) v = *reinterpret_cast<const uint4*>(src + i0 * ss0 + i1 * ss1 + i2 * ss2 + c * 8);
    *reinterpret_cast<uint4*>(dst + i0 * ds0 + i1 * ds1 + i2 * ds2 + c * 8) = v;
  }
}

// several copy_4d problems over the same (src, dst) pair in ONE launch (blockIdx.y = problem): the P pack (or unpack) pieces
// of a DSP / Ulysses all-to-all
struct CopyBatch {
  int nops;
  CopyDesc d[VSYS_COPY_BATCH_MAX];
};
__global__ void copy_4d_batch_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, CopyBatch b) {
  const CopyDesc& o = b.d[blockIdx.y];
  const int cch = o.C >> 3;
  const int64_t total = (int64_t)o.n0 * o.n1 * o.n2 * cch;
  const bf16_t* s = src + o.src_off;
  bf16_t* d = dst + o.dst_off;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cch);
    int64_t r = i / cch;
    const int i2 = (int)(r % o.n2);
    r /= o.n2;
    const int i1 = (int)(r % o.n1);
    const int i0 = (int)(r / o.n1);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (i1 < o.n1_valid && i2 < o.n2_valid) v = *reinterpret_cast<const uint4*>(s + i0 * o.ss0 + i1 * o.ss1 + i2 * o.ss2 + c * 8);
    *reinterpret_cast<uint4*>(d + i0 * o.ds0 + i1 * o.ds1 + i2 * o.ds2 + c * 8) = v;
  }
}

}  // namespace

int launch_copy_4d_batch(const bf16_t* src, bf16_t* dst, const CopyDesc* ops, int nops, hipStream_t stream) {
  if (nops <= 0) return 0;
  if (nops > VSYS_COPY_BATCH_MAX) return VSYS_ERR_SHAPE;
  CopyBatch b;
  b.nops = nops;
  int64_t most = 0;
  for (int i = 0; i < nops; ++i) {
    if (ops[i].C % 8 || ops[i].n0 < 0 || ops[i].n1 < 0 || ops[i].n2 < 0) return VSYS_ERR_SHAPE;
    b.d[i] = ops[i];
    const int64_t t = (int64_t)ops[i].n0 * ops[i].n1 * ops[i].n2 * (ops[i].C / 8);
    most = t > most ? t : most;
  }
  if (most <= 0) return 0;
  int64_t grid = (most + 255) / 256;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(copy_4d_batch_kernel, dim3((unsigned)grid, (unsigned)nops), dim3(256), 0, stream, src, dst, b);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_adaln_modulate(const bf16_t* x, const bf16_t* shift, const bf16_t* scale, bf16_t* y, int64_t rows, int C,
                          int64_t rows_per_sample, int64_t mod_stride, float eps, hipStream_t stream) {
  if (rows <= 0) return 0;
  if (C % 8 != 0 || C > 64 * 8 * 4 || rows_per_sample <= 0 || (mod_stride % 8)) return VSYS_ERR_SHAPE;
  // two rows per wave at the STDiT3 / Latte width (measured at 38912 x 1152: 31.2 us against 33.2 for one row and 34-35 for four)
  if (C > 64 * 8 * 2 && C <= 64 * 8 * 3 && rows_per_sample % 2 == 0) {
    hipLaunchKernelGGL(adaln_modulate_rows_kernel<2>, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, stream, x, shift, scale, y, rows, C,
                       rows_per_sample, mod_stride, eps);
    return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
  }
  const unsigned grid = (unsigned)((rows + 3) / 4);
  if (C <= 64 * 8 * 2)
    hipLaunchKernelGGL(adaln_modulate_kernel<2>, dim3(grid), dim3(256), 0, stream, x, shift, scale, y, rows, C,
                       rows_per_sample, mod_stride, eps);
  else if (C <= 64 * 8 * 3)
    hipLaunchKernelGGL(adaln_modulate_kernel<3>, dim3(grid), dim3(256), 0, stream, x, shift, scale, y, rows, C,
                       rows_per_sample, mod_stride, eps);
  else
    hipLaunchKernelGGL(adaln_modulate_kernel<4>, dim3(grid), dim3(256), 0, stream, x, shift, scale, y, rows, C,
                       rows_per_sample, mod_stride, eps);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_ln_modulate(const bf16_t* x, const bf16_t* ln_w, const bf16_t* ln_b, const bf16_t* shift, const bf16_t* scale,
                       bf16_t* y, int64_t rows, int C, int64_t rows_per_sample, int64_t mod_stride, int64_t seg_split,
                       int64_t mod_alt, float eps, hipStream_t stream) {
  if (rows <= 0) return 0;
  if (C % 8 != 0 || C > 64 * 8 * 6 || rows_per_sample <= 0 || (mod_stride % 8) || (mod_alt % 8)) return VSYS_ERR_SHAPE;
  if ((ln_w == nullptr) != (ln_b == nullptr) || (shift == nullptr) != (scale == nullptr)) return VSYS_ERR_ARG;
  const unsigned grid = (unsigned)((rows + 3) / 4);
#define VSYS_LNMOD(V)                                                                                                      \
  hipLaunchKernelGGL(ln_modulate_kernel<V>, dim3(grid), dim3(256), 0, stream, x, ln_w, ln_b, shift, scale, y, rows, C,       \
                     rows_per_sample, mod_stride, seg_split, mod_alt, eps)
  if (C <= 64 * 8 * 2) VSYS_LNMOD(2);
  else if (C <= 64 * 8 * 3) VSYS_LNMOD(3);
  else if (C <= 64 * 8 * 4) VSYS_LNMOD(4);
  else VSYS_LNMOD(6);
#undef VSYS_LNMOD
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_gate_add_rows(bf16_t* x, const bf16_t* y, const bf16_t* gate, int64_t rows, int C, int64_t rows_per_sample,
                         int64_t gate_stride, int64_t seg_split, int64_t gate_alt, hipStream_t stream) {
  if (rows <= 0) return 0;
  if (C % 8 || rows_per_sample <= 0 || (gate_stride % 8) || (gate_alt % 8)) return VSYS_ERR_SHAPE;
  int64_t grid = (rows * (C / 8) + 255) / 256;
  if (grid > 2048 * 4) grid = 2048 * 4;
  hipLaunchKernelGGL(gate_add_rows_kernel, dim3((unsigned)grid), dim3(256), 0, stream, x, y, gate, rows, C / 8, rows_per_sample,
                     gate_stride, seg_split, gate_alt);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_im2col_patch(const float* z, int Bz, bf16_t* out, int B, int F, int Cin, int H, int W, int p, hipStream_t stream) {
  if (p <= 0 || H % p || W % p || Bz <= 0) return VSYS_ERR_SHAPE;
  const int64_t total = (int64_t)B * F * (H / p) * (W / p) * Cin * p * p;
  if (total <= 0) return 0;
  int64_t grid = (total + 255) / 256;
  if (grid > 2048 * 8) grid = 2048 * 8;
  hipLaunchKernelGGL(im2col_patch_kernel, dim3((unsigned)grid), dim3(256), 0, stream, z, Bz, out, B, F, Cin, H, W, p);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_unpatchify_cvx(const bf16_t* x, int64_t ldx, float* out, int B, int F, int Hp, int Wp, int Cout, int p,
                          hipStream_t stream) {
  const int64_t total = (int64_t)B * F * Cout * Hp * p * Wp * p;
  if (total <= 0) return 0;
  int64_t grid = (total + 255) / 256;
  if (grid > 2048 * 8) grid = 2048 * 8;
  hipLaunchKernelGGL(unpatchify_cvx_kernel, dim3((unsigned)grid), dim3(256), 0, stream, x, ldx, out, B, F, Hp, Wp, Cout, p);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_mod_table(const bf16_t* table, const bf16_t* t_mlp, bf16_t* out, int nblk, int B, int C6, hipStream_t stream) {
  const int64_t total = (int64_t)nblk * B * C6;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(mod_table_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, table, t_mlp, out, nblk,
                     B, C6);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_timestep_embedding(const float* t, bf16_t* out, int B, int dim, hipStream_t stream) {
  if (dim % 2) return VSYS_ERR_SHAPE;
  const int total = B * (dim / 2);
  if (total <= 0) return 0;
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((total + 127) / 128), dim3(128), 0, stream, t, out, B, dim);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_patch_embed(const float* x, int Bz, const bf16_t* w, const bf16_t* bias, const bf16_t* pos, bf16_t* out, int B,
                       int Cin, int T, int H, int W, int ph, int pw, int C, int s0, int Sl, hipStream_t stream) {
  if (C % 8 || Bz <= 0 || s0 < 0) return VSYS_ERR_SHAPE;
  const int Hp = (H + ph - 1) / ph, Wp = (W + pw - 1) / pw;
  if (Sl < 0) Sl = Hp * Wp;   // the whole frame
  const int64_t total = (int64_t)B * T * Sl * (C / 8);
  if (total <= 0) return 0;
  if ((total + 255) / 256 > 0x7fffffff) return VSYS_ERR_SHAPE;
  hipLaunchKernelGGL(patch_embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, Bz, w, bias, pos, out,
                     B, Cin, T, H, W, ph, pw, C, s0, Sl);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_unpatchify_tokens(const float* tok, float* out, int P, int B, int T, int Sl, int Hp, int Wp, int H, int W, int ph,
                             int pw, int Cout, hipStream_t stream) {
  if (P <= 0 || Sl <= 0 || (int64_t)P * Sl < (int64_t)Hp * Wp) return VSYS_ERR_SHAPE;
  const int64_t total = (int64_t)B * Cout * T * H * W;
  if (total <= 0) return 0;
  if ((total + 255) / 256 > 0x7fffffff) return VSYS_ERR_SHAPE;
  hipLaunchKernelGGL(unpatchify_tokens_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, tok, out, P, B, T, Sl,
                     Hp, Wp, H, W, ph, pw, Cout);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_final_layer(const bf16_t* x, const bf16_t* table, const bf16_t* tvec, const bf16_t* w, const bf16_t* bias,
                       float* out, int B, int T, int Hp, int Wp, int H, int W, int ph, int pw, int Cout, int C, float eps,
                       int Sl, float* tokens, hipStream_t stream) {
  if (C % 8 != 0 || C > 64 * 8 * 4) return VSYS_ERR_SHAPE;
  if (Sl < 0) Sl = Hp * Wp;
  if (tokens == nullptr && Sl != Hp * Wp) return VSYS_ERR_ARG;   // pixels can only be addressed from whole frames
  const int64_t rows = (int64_t)B * T * Sl;
  if (rows <= 0) return 0;
  const unsigned grid = (unsigned)((rows + 3) / 4);
  if (C <= 64 * 8 * 2)
    hipLaunchKernelGGL(final_layer_kernel<2>, dim3(grid), dim3(256), 0, stream, x, table, tvec, w, bias, out, B, T, Hp, Wp,
                       H, W, ph, pw, Cout, C, eps, Sl, tokens);
  else if (C <= 64 * 8 * 3)
    hipLaunchKernelGGL(final_layer_kernel<3>, dim3(grid), dim3(256), 0, stream, x, table, tvec, w, bias, out, B, T, Hp, Wp,
                       H, W, ph, pw, Cout, C, eps, Sl, tokens);
  else
    hipLaunchKernelGGL(final_layer_kernel<4>, dim3(grid), dim3(256), 0, stream, x, table, tvec, w, bias, out, B, T, Hp, Wp,
                       H, W, ph, pw, Cout, C, eps, Sl, tokens);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_cfg_euler(float* z, const float* model_out, int Bz, int Cin, int Cout, int64_t thw, float guidance, float dt,
                     hipStream_t stream) {
  const int64_t total = (int64_t)Bz * Cin * thw;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(cfg_euler_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, z, model_out, Bz, Cin,
                     Cout, thw, guidance, dt);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_cfg_axpby(float* z, const float* model_out, int Bz, int Cin, int Cout, int64_t thw, float guidance, float c_z,
                     float c_eps, int cond_first, hipStream_t stream) {
  const int64_t total = (int64_t)Bz * Cin * thw;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(cfg_axpby_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, z, model_out, Bz, Cin,
                     Cout, thw, guidance, c_z, c_eps, cond_first);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_add_bcast_rows(bf16_t* x, const bf16_t* e, int64_t rows, int C, int64_t group, int64_t period, hipStream_t stream) {
  if (rows <= 0) return 0;
  if (C % 8 || group <= 0 || period <= 0) return VSYS_ERR_SHAPE;
  int64_t grid = (rows * (C / 8) + 255) / 256;
  if (grid > 2048 * 4) grid = 2048 * 4;
  hipLaunchKernelGGL(add_bcast_rows_kernel, dim3((unsigned)grid), dim3(256), 0, stream, x, e, rows, C / 8, group, period);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_add_rows(bf16_t* x, const bf16_t* y, int64_t n, hipStream_t stream) {
  if (n <= 0) return 0;
  if (n % 8) return VSYS_ERR_SHAPE;
  const int64_t n8 = n / 8;
  int64_t grid = (n8 + 255) / 256;
  if (grid > 2048 * 4) grid = 2048 * 4;
  hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)grid), dim3(256), 0, stream, x, y, n8);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_copy_4d(const bf16_t* src, bf16_t* dst, int n0, int n1, int n2, int C, int64_t ss0, int64_t ss1, int64_t ss2,
                   int64_t ds0, int64_t ds1, int64_t ds2, int n1_valid, int n2_valid, hipStream_t stream) {
  if (C % 8) return VSYS_ERR_SHAPE;
  const int64_t total = (int64_t)n0 * n1 * n2 * (C / 8);
  if (total <= 0) return 0;
  int64_t grid = (total + 255) / 256;
  if (grid > 2048 * 4) grid = 2048 * 4;
  hipLaunchKernelGGL(copy_4d_kernel, dim3((unsigned)grid), dim3(256), 0, stream, src, dst, n0, n1, n2, C, ss0, ss1, ss2, ds0,
                     ds1, ds2, n1_valid, n2_valid);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

}  // namespace vsys
