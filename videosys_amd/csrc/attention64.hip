// Attention kernels for head_dim 64 (CogVideoX: 30 x 64 = 1920 / 48 x 64 = 3072) on gfx950.
//
// Reference call site replaced (/root/reference/videosys): CogVideoXAttnProcessor2_0.__call__
//   models/transformers/cogvideox_transformer_3d.py:93-175 — joint [text | video] self-attention: LayerNorm qk-norm
//   (diffusers Attention(qk_norm="layer_norm", eps=1e-6): affine LayerNorm over the 64 head dims), rotary embedding on the
//   video slice of q and k (modules/embeddings.py:358-412, interleaved pairs), F.scaled_dot_product_attention.
//
// Same structure as the d72 flash kernel of attention.hip (swapped MFMA forms so a query's softmax row is lane-local,
// LDS-DMA K/V staging, running max as the MFMA C operand, deferred rescale, exp2-ready logits because the softmax scale
// rides on K) with the d64 geometry: 4 QK^T chunks of 16, 2 PV tiles of 32 (no padding rows, so the softmax denominator
// is summed on the VALU), K rows of 128 bytes whose 16-byte chunks are XOR-swizzled by (row>>1)&7 — the swizzle is
// written by attn_prep_kv64 into the HBM image, so the tile stays one contiguous 8 KiB LDS-DMA.
#include <cstdlib>

#include "common.h"
#include "vsys_internal.h"

namespace vsys {
namespace {

constexpr int HD = 64;
constexpr int KROW = 128;                      // bytes per K row (64 bf16), chunks swizzled
constexpr int VROW = 128;                      // bytes per Vt row in LDS (64 keys), slots swizzled
constexpr int K_TILE_BYTES = 64 * KROW;        // 8192
constexpr int V_TILE_BYTES = HD * VROW;        // 8192
constexpr int KV_STAGE = K_TILE_BYTES + V_TILE_BYTES;  // 16384
constexpr float NEG_BIG = -1.0e30f;

// LayerNorm (affine, biased variance) over 64 values held 16 per lane by 4 consecutive lanes; bf16 result like
// F.layer_norm on a bf16 tensor.  ln_w == nullptr: identity.
__device__ __forceinline__ void ln64_quad(float* x, const bf16_t* __restrict__ ln_w, const bf16_t* __restrict__ ln_b, int part,
                                          float eps) {
  if (ln_w == nullptr) return;
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) s += x[e];
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  const float mean = s * (1.0f / 64.0f);
  float v = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) v += (x[e] - mean) * (x[e] - mean);
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  const float rstd = rsqrtf(v * (1.0f / 64.0f) + eps);
#pragma unroll
  for (int e = 0; e < 16; ++e)
    x[e] = bf2f(f2bf((x[e] - mean) * rstd * bf2f(ln_w[part * 16 + e]) + (ln_b ? bf2f(ln_b[part * 16 + e]) : 0.f)));
}

// ---------------------------------------------------------------------------------------------------------
// attn_prep_kv64: k, v rows (strided, heads interleaved) ->
//   Kp[batch][H][kv_pad][64]  LayerNorm + RoPE + softmax scale, 16-byte chunks of a row stored at chunk ^ ((row>>1)&7)
//   Vt[batch][H][64][kv_pad]  transposed
// grid: (kv_pad/64, batch*H); block 256 = 64 token rows x 4 lanes (16 dims each).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_prep_kv64_kernel(const bf16_t* __restrict__ k, int64_t k_stride,
                                                             const bf16_t* __restrict__ v, int64_t v_stride,
                                                             const bf16_t* __restrict__ ln_w, const bf16_t* __restrict__ ln_b,
                                                             const float* __restrict__ rope_cos,
                                                             const float* __restrict__ rope_sin, int rope_start, int rope_len,
                                                             bf16_t* __restrict__ kp, bf16_t* __restrict__ vt, int heads,
                                                             int kv_len, int kv_pad, float eps, float kscale) {
  __shared__ __attribute__((aligned(16))) bf16_t vs[64][HD + 8];  // [token][d], 144-byte rows
  const int tid = threadIdx.x;
  const int s0 = blockIdx.x * 64;
  const int bh = blockIdx.y;
  const int b = bh / heads, h = bh - b * heads;
  const int r = tid >> 2, part = tid & 3;
  const int s = s0 + r;
  // ---- V rows -> LDS (each thread 16 dims of one token)
  {
    uint4 a = make_uint4(0, 0, 0, 0), c = a;
    if (s < kv_len) {
      const bf16_t* src = v + ((int64_t)b * kv_len + s) * v_stride + h * HD + part * 16;
      a = *reinterpret_cast<const uint4*>(src);
      c = *reinterpret_cast<const uint4*>(src + 8);
    }
    *reinterpret_cast<uint4*>(&vs[r][part * 16]) = a;
    *reinterpret_cast<uint4*>(&vs[r][part * 16 + 8]) = c;
  }
  // ---- K row: norm, rope, scale, swizzled store
  {
    float x[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = 0.f;
    if (s < kv_len) {
      const bf16_t* src = k + ((int64_t)b * kv_len + s) * k_stride + h * HD + part * 16;
      unpack8(*reinterpret_cast<const uint4*>(src), x);
      unpack8(*reinterpret_cast<const uint4*>(src + 8), x + 8);
    }
    ln64_quad(x, ln_w, ln_b, part, eps);
    const int rp = s - rope_start;
    if (rope_cos != nullptr && rp >= 0 && rp < rope_len) {
      const float* cs = rope_cos + (int64_t)rp * HD + part * 16;
      const float* sn = rope_sin + (int64_t)rp * HD + part * 16;
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const float a = x[e], bb = x[e + 1];
        x[e] = a * cs[e] - bb * sn[e];
        x[e + 1] = bb * cs[e + 1] + a * sn[e + 1];
      }
    }
    if (s >= kv_len) {
#pragma unroll
      for (int e = 0; e < 16; ++e) x[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] *= kscale;  // softmax scale * log2(e) rides on K (single rounding below)
    const int sw = (r >> 1) & 7;
    bf16_t* dst = kp + ((int64_t)bh * kv_pad + s) * HD;
    *reinterpret_cast<uint4*>(dst + (((2 * part) ^ sw) << 3)) = pack8(x);
    *reinterpret_cast<uint4*>(dst + (((2 * part + 1) ^ sw) << 3)) = pack8(x + 8);
  }
  __syncthreads();
  // ---- Vt rows: thread -> (d, 8-token chunk), 512 chunks
  for (int q = tid; q < HD * 8; q += 256) {
    const int d = q >> 3, c = q & 7;
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = (uint32_t)vs[c * 8 + 2 * e][d] | ((uint32_t)vs[c * 8 + 2 * e + 1][d] << 16);
    uint4 o;
    o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
    *reinterpret_cast<uint4*>(vt + ((int64_t)bh * HD + d) * kv_pad + s0 + c * 8) = o;
  }
}

struct Flash64Params {
  const bf16_t* q; int64_t q_stride;      // q(b, s, h) at q + (b*q_len + s)*q_stride + h*64
  const bf16_t* ln_w; const bf16_t* ln_b;  // q LayerNorm weight / bias [64] or null
  const float* rope_cos; const float* rope_sin; int rope_start, rope_len;
  const bf16_t* kp;                        // [batch][H][kv_pad][64] (swizzled chunks)
  const bf16_t* vt;                        // [batch][H][64][kv_pad]
  bf16_t* out; int64_t out_stride;
  int heads, q_len, kv_len, kv_pad, nqb;
  float eps;
  // BIAS kernels only: additive logit bias that depends on (head, key - query) — T5's relative-position bias — as an fp32
  // table in the exp2 domain, entry of (h, j - i) at bias[h * bias_ld + bias_center + j - i]
  const float* bias; int bias_ld, bias_center;
};

// grid: ceil(q_len/128) * batch * heads workgroups of 4 waves x 32 query rows (1-D, XCD-remapped so the q-blocks of one
// (batch, head) share an L2).
// NS = stages of the K/V ring: tile t + NS - 1 is requested at the top of tile t; the wait at the end of tile t is COUNTED — a wave
// issues four 1-KiB LDS-DMA pieces per tile, so vmcnt(4 k) means "everything but my newest k tiles" = tile t + 1 has landed — and
// the barrier is the raw s_barrier (__syncthreads() would drain the DMA queue).  An LDS-DMA piece lands ~1.1 us after issue
// (MI355X_MICROARCH.md), longer than a wave-tile of compute: with NS = 2 (request t + 1, drain at the end of tile t) every tile
// ended in a wait.  Measured on CogVideoX-5B (config 5, 42 blocks, one GPU; profiles/r03_cogvideox_ring.json): box A 659.8 ms per
// step with NS = 2 vs 636.6 / 632.3 with NS = 3; box B 671.0 vs 678.4 / 676.8 (and 688 with NS = 4): the part is power-capped and
// the boxes of the pool differ by more than the effect — NS = 3 ships on the sum of the two (-1.4 %), NS = 2 stays selectable.
// NS x 16 KiB of LDS per workgroup, two workgroups per CU.
template <int NS, bool BIAS = false>
__global__ __launch_bounds__(256, 2) void flash_attn_d64_kernel(Flash64Params p) {
#if __HIP_DEVICE_COMPILE__  // buffer-resource types exist in the device pass only
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int tile_id = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = tile_id / p.nqb;
  const int qb = tile_id - bh * p.nqb;
  const int b = bh / p.heads, h = bh - b * p.heads;
  const int q0 = qb * 128 + wave * 32;

  // ---- Q fragment (B operand): lane holds Q[q0 + l31][16c + 8hi .. +8], c = 0..3; LayerNorm + RoPE applied here
  bf16x8 qf[4];
  {
    int qs = q0 + l31;
    qs = qs < p.q_len ? qs : p.q_len - 1;
    const bf16_t* qrow = p.q + ((int64_t)b * p.q_len + qs) * p.q_stride + h * HD;
    float x[4][8];
#pragma unroll
    for (int c = 0; c < 4; ++c) unpack8(*reinterpret_cast<const uint4*>(qrow + 16 * c + 8 * hi), x[c]);
    if (p.ln_w != nullptr) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += x[c][e];
      s += __shfl_xor(s, 32, 64);
      const float mean = s * (1.0f / 64.0f);
      float v = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) v += (x[c][e] - mean) * (x[c][e] - mean);
      v += __shfl_xor(v, 32, 64);
      const float rstd = rsqrtf(v * (1.0f / 64.0f) + p.eps);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float w[8], bb[8];
        unpack8(*reinterpret_cast<const uint4*>(p.ln_w + 16 * c + 8 * hi), w);
        if (p.ln_b != nullptr) {
          unpack8(*reinterpret_cast<const uint4*>(p.ln_b + 16 * c + 8 * hi), bb);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) bb[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) x[c][e] = bf2f(f2bf((x[c][e] - mean) * rstd * w[e] + bb[e]));
      }
    }
    const int rp = qs - p.rope_start;
    if (p.rope_cos != nullptr && rp >= 0 && rp < p.rope_len) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4* cs = reinterpret_cast<const float4*>(p.rope_cos + (int64_t)rp * HD + 16 * c + 8 * hi);
        const float4* sn = reinterpret_cast<const float4*>(p.rope_sin + (int64_t)rp * HD + 16 * c + 8 * hi);
        const float4 c0 = cs[0], c1 = cs[1], s0 = sn[0], s1 = sn[1];
        const float cv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const float a = x[c][e], bb = x[c][e + 1];
          x[c][e] = a * cv[e] - bb * sv[e];
          x[c][e + 1] = bb * cv[e + 1] + a * sv[e + 1];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[c][e] = (__bf16)x[c][e];
  }

  // ---- K/V staging by LDS-DMA: 8 K pieces + 8 Vt pieces of 1 KiB per tile; wave w issues pieces w, w+4, w+8, w+12
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const bf16_t* kbase = p.kp + (int64_t)bh * p.kv_pad * HD;
  const bf16_t* vbase = p.vt + (int64_t)bh * HD * p.kv_pad;
  const auto rsrc_k = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, p.kv_pad * HD * 2, 0x00020000);
  const auto rsrc_v = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, HD * p.kv_pad * 2, 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int k_voff = lane * 16;
  // Vt piece j: lane -> row 8j + (lane>>3), physical slot lane&7 holds logical slot (lane&7) ^ ((row>>1)&7);
  // (row>>1)&7 = ((j&1)<<2) | (lane>>4), so odd pieces differ from even ones by XOR 64 in the byte offset
  const int v_voff = (lane >> 3) * p.kv_pad * 2 + (((lane & 7) ^ (lane >> 4)) << 4);
  auto stage = [&](int t, int buf) {
    char* base = smem + buf * KV_STAGE;
#pragma unroll
    for (int idx = 0; idx < 4; ++idx) {
      const int piece = wave_u + 4 * idx;
      if (piece < 8) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_k, (lds_ptr_t)(base + piece * 1024), 16, k_voff, t * K_TILE_BYTES + piece * 1024, 0, 0);
      } else {
        const int j = piece - 8;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_v, (lds_ptr_t)(base + K_TILE_BYTES + j * 1024), 16, v_voff ^ ((j & 1) << 6),
                                                 j * 8 * p.kv_pad * 2 + t * 128, 0, 0);
      }
    }
  };
  // permuted K row for MFMA row i = l31: lane's 16 acc regs <-> 16 consecutive keys (16*hi + reg)
  const int krow = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
  // K fragment read: row kt*32 + krow, logical chunk 2cc + hi at physical chunk ^ ((row>>1)&7)  ->  + kt*4096, ^ (cc << 5)
  const int k_roff = krow * KROW + ((hi ^ ((krow >> 1) & 7)) << 4);
  // Vt fragment read: row dt*32 + l31, logical slot kt*4 + 2hi + cc  ->  + dt*4096, ^ ((kt*4 + cc) << 4)
  const int v_roff = K_TILE_BYTES + l31 * VROW + (((2 * hi) ^ ((l31 >> 1) & 7)) << 4);

  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  // negated running max (exp2 domain), splatted: the C operand of the first QK^T MFMA of every 32-key tile
  f32x16 minit;
#pragma unroll
  for (int r = 0; r < 16; ++r) minit[r] = 0.f;
  float l_run = 0.f;

  const int ntiles = (p.kv_len + 63) / 64;
  constexpr int LEAD = NS - 1;   // tiles requested ahead of the one being computed
  // wait until at most ``k`` of this wave's tile requests (4 pieces each) are outstanding; k is wave-uniform
  auto wait_pending = [&](int k) {
    if (k <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (k == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (k == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  };
  stage(0, 0);
#pragma unroll
  for (int j = 1; j < LEAD; ++j)
    if (j < ntiles) stage(j, j);                        // fly under tile 0 (no other vector-memory operation is outstanding: Q was consumed)
  wait_pending(LEAD - 1 < ntiles - 1 ? LEAD - 1 : ntiles - 1);
  __builtin_amdgcn_s_barrier();
  const float defer_thr = 8.0f;

  auto tile = [&](int t, int cur, const bool masked) {
    const char* sk = smem + cur * KV_STAGE;
    bf16x8 kf0[4], kf1[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) kf0[cc] = *reinterpret_cast<const bf16x8*>(sk + (k_roff ^ (cc << 5)));
    __builtin_amdgcn_sched_barrier(0);
    f32x16 s[2];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) kf1[cc] = *reinterpret_cast<const bf16x8*>(sk + 32 * KROW + (k_roff ^ (cc << 5)));
    // D != C on purpose (the builtin ties them and hipcc would first copy the 16 minit registers into s)
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(s[0]) : "v"(kf0[0]), "v"(qf[0]), "v"(minit));
#pragma unroll
    for (int cc = 1; cc < 4; ++cc) s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0[cc], qf[cc], s[0], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(s[1]) : "v"(kf1[0]), "v"(qf[0]), "v"(minit));
#pragma unroll
    for (int cc = 1; cc < 4; ++cc) s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1[cc], qf[cc], s[1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 vf0[2][2], vf1[2][2];
#define FLASH64_VREAD(dst_, kt_)                                                                                 \
  _Pragma("unroll") for (int cc = 0; cc < 2; ++cc) _Pragma("unroll") for (int dt = 0; dt < 2; ++dt)              \
    dst_[cc][dt] = *reinterpret_cast<const bf16x8*>(sk + dt * 32 * VROW + (v_roff ^ (((kt_) * 4 + cc) << 4)))
    FLASH64_VREAD(vf0, 0);
    __builtin_amdgcn_sched_barrier(0);

    if constexpr (BIAS) {
      // element (kt, r) of this lane is key t*64 + kt*32 + 16*hi + r against query q0 + l31: 16 consecutive table entries per kt.
      // The BIAS caller leaves K UNSCALED (T5 logits are large — no 1/sqrt(d) — and a log2(e) folded into bf16 K would cost 2^-9
      // of each of them): the accumulators are q k - m in natural units (minit is kept in natural units too) and move to the
      // exp2 domain here, in the same fma that adds the bias.
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int idx = p.bias_center + (t * 64 + kt * 32 + 16 * hi + r) - (q0 + l31);
          s[kt][r] = __builtin_fmaf(s[kt][r], 1.4426950408889634f, p.bias[(int64_t)h * p.bias_ld + idx]);
        }
    }
    if (masked) {
      const int lim = p.kv_len - (t * 64 + 16 * hi);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt * 32 + r >= lim) s[kt][r] = NEG_BIG;
    }
    float mx = s[0][0];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    if (t == 0 || __builtin_amdgcn_ballot_w64(mx > defer_thr) != 0) {  // wave-uniform
      asm volatile("; rescale path (rare): kept out of line" ::: "memory");
      const float delta = t == 0 ? mx : fmaxf(mx, 0.f);
      const float alpha = __builtin_amdgcn_exp2f(-delta);
      l_run *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) minit[r] -= BIAS ? delta * 0.6931471805599453f : delta;   // (BIAS: natural units, see above)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] -= delta;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
    bf16x8 pf0[2], pf1[2];
    float lsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = __builtin_amdgcn_exp2f(s[0][r]);
      lsum += pv;
      pf0[r >> 3][r & 7] = (__bf16)pv;
    }
    FLASH64_VREAD(vf1, 1);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf0[cc][dt], pf0[cc], o[dt], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = __builtin_amdgcn_exp2f(s[1][r]);
      lsum += pv;
      pf1[r >> 3][r & 7] = (__bf16)pv;
    }
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf1[cc][dt], pf1[cc], o[dt], 0, 0, 0);
    l_run += lsum;
#undef FLASH64_VREAD
  };

  {
    int cur = 0, nxt = LEAD % NS;   // buffer of tile t, buffer of tile t + LEAD (= the one tile t - 1 left)
    for (int t = 0; t < ntiles - 1; ++t) {
      if (t + LEAD < ntiles) stage(t + LEAD, nxt);   // last read during tile t - 1: every wave passed that tile's barrier
      __builtin_amdgcn_sched_barrier(0);
      tile(t, cur, false);
      __builtin_amdgcn_sched_barrier(0);
      // tiles t + 2 .. min(t + LEAD, ntiles - 1) may stay in flight; tile t + 1 must have landed ...
      wait_pending(LEAD - 1 < ntiles - t - 2 ? LEAD - 1 : ntiles - t - 2);
      __builtin_amdgcn_s_barrier();   // ... for every wave
      cur = cur == NS - 1 ? 0 : cur + 1;
      nxt = nxt == NS - 1 ? 0 : nxt + 1;
    }
    if (p.kv_len & 63) tile(ntiles - 1, cur, true);
    else tile(ntiles - 1, cur, false);
  }

  // ---- epilogue: O[q][d] = O^T[d][q] / l ; lane holds d = 32dt + (r&3) + 8(r>>2) + 4hi
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int qs = q0 + l31;
  if (qs < p.q_len) {
    bf16_t* orow = p.out + ((int64_t)b * p.q_len + qs) * p.out_stride + h * HD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + 8 * g + 4 * hi;
        uint2 w;
        w.x = pack2bf(o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv);
        w.y = pack2bf(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(orow + d) = w;
      }
  }
#endif
}

}  // namespace

static int prep_kv64(const bf16_t* k, int64_t k_stride, const bf16_t* v, int64_t v_stride, const bf16_t* ln_w, const bf16_t* ln_b,
                     const float* rope_cos, const float* rope_sin, int rope_start, int rope_len, bf16_t* kp, bf16_t* vt, int batch,
                     int heads, int kv_len, int kv_pad, float eps, float kscale, hipStream_t stream) {
  if (batch <= 0 || heads <= 0 || kv_len <= 0) return 0;
  if (kv_pad % 64 != 0 || kv_pad < kv_len || (k_stride % 8) || (v_stride % 8)) return VSYS_ERR_SHAPE;
  if ((rope_cos == nullptr) != (rope_sin == nullptr)) return VSYS_ERR_ARG;
  dim3 grid(kv_pad / 64, batch * heads);
  hipLaunchKernelGGL(attn_prep_kv64_kernel, grid, dim3(256), 0, stream, k, k_stride, v, v_stride, ln_w, ln_b, rope_cos, rope_sin,
                     rope_start, rope_len, kp, vt, heads, kv_len, kv_pad, eps, kscale);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_attn_prep_kv64(const bf16_t* k, int64_t k_stride, const bf16_t* v, int64_t v_stride, const bf16_t* ln_w,
                          const bf16_t* ln_b, const float* rope_cos, const float* rope_sin, int rope_start, int rope_len,
                          bf16_t* kp, bf16_t* vt, int batch, int heads, int kv_len, int kv_pad, float eps, hipStream_t stream) {
  return prep_kv64(k, k_stride, v, v_stride, ln_w, ln_b, rope_cos, rope_sin, rope_start, rope_len, kp, vt, batch, heads, kv_len, kv_pad,
                   eps, 0.125f * 1.4426950408889634f /* 64^-0.5 * log2(e) */, stream);
}

// T5 self-attention (transformers T5Attention.forward, third-party; the encoder every pipeline calls once per prompt): no
// 1/sqrt(d) scaling, additive relative-position bias, keys >= kv_len masked.  One sample per call (its own key length): the K/V
// layouts of its heads are written by the d64 prep kernel (K unscaled), the d64 flash kernel runs with the BIAS hook.
// bias: fp32 [heads, bias_ld] in the exp2 domain, entry of (h, key - query) at bias_center + key - query; the caller pads the
// table so that every (key < kv_pad, query < 128 * ceil(L / 128)) index is inside it.
int launch_t5_attention_mfma(const bf16_t* qkv, int64_t row_stride, int inner, const float* bias, int bias_ld, int bias_center,
                             int kv_len, bf16_t* kp, bf16_t* vt, bf16_t* out, int64_t out_stride, int L, int heads,
                             hipStream_t stream) {
  if (L <= 0 || heads <= 0) return 0;
  if (inner != heads * 64 || kv_len <= 0 || kv_len > L || (row_stride % 8) || (out_stride % 4) || bias == nullptr) return VSYS_ERR_SHAPE;
  const int kv_pad = (L + 63) / 64 * 64, qpad = (L + 127) / 128 * 128;
  if (bias_center < qpad - 1 || bias_ld < bias_center + kv_pad) return VSYS_ERR_SHAPE;
  const int rc = prep_kv64(qkv + inner, row_stride, qkv + 2 * inner, row_stride, nullptr, nullptr, nullptr, nullptr, 0, 0, kp, vt, 1, heads,
                           L, kv_pad, 0.f, 1.0f, stream);
  if (rc != 0) return rc;
  Flash64Params p;
  p.q = qkv; p.q_stride = row_stride; p.ln_w = nullptr; p.ln_b = nullptr; p.rope_cos = nullptr; p.rope_sin = nullptr;
  p.rope_start = 0; p.rope_len = 0; p.kp = kp; p.vt = vt; p.out = out; p.out_stride = out_stride;
  p.heads = heads; p.q_len = L; p.kv_len = kv_len; p.kv_pad = kv_pad; p.eps = 0.f;
  p.nqb = (L + 127) / 128;
  p.bias = bias; p.bias_ld = bias_ld; p.bias_center = bias_center;
  hipLaunchKernelGGL((flash_attn_d64_kernel<3, true>), dim3((unsigned)(p.nqb * heads)), dim3(256), 3 * KV_STAGE, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_flash_attn_d64(const bf16_t* q, int64_t q_stride, const bf16_t* ln_w, const bf16_t* ln_b, const float* rope_cos,
                          const float* rope_sin, int rope_start, int rope_len, const bf16_t* kp, const bf16_t* vt, bf16_t* out,
                          int64_t out_stride, int batch, int heads, int q_len, int kv_len, int kv_pad, float eps, float k_bound,
                          hipStream_t stream) {
  if (batch <= 0 || heads <= 0 || q_len <= 0) return 0;
  if (kv_len <= 0 || kv_pad % 64 != 0 || kv_pad < kv_len || (q_stride % 8) || (out_stride % 4)) return VSYS_ERR_SHAPE;
  if ((rope_cos == nullptr) != (rope_sin == nullptr)) return VSYS_ERR_ARG;
  Flash64Params p;
  p.q = q; p.q_stride = q_stride; p.ln_w = ln_w; p.ln_b = ln_b; p.rope_cos = rope_cos; p.rope_sin = rope_sin;
  p.rope_start = rope_start; p.rope_len = rope_len; p.kp = kp; p.vt = vt; p.out = out; p.out_stride = out_stride;
  p.heads = heads; p.q_len = q_len; p.kv_len = kv_len; p.kv_pad = kv_pad; p.eps = eps;
  p.nqb = (q_len + 127) / 128;
  p.bias = nullptr; p.bias_ld = 0; p.bias_center = 0;
  const int64_t nblk = (int64_t)p.nqb * batch * heads;
  if (nblk > 0x7fffffff) return VSYS_ERR_SHAPE;
  const int fv = get_flash_variant();
  // long key sequences (CogVideoX: 17 776): 64 query rows per wave, one wave per SIMD, hand-allocated instruction stream
  // (attention64_w64.hip); 15 = never (the A/B id of the measurement tools), 14 = wherever it is supported
  static const bool w64_off = [] { const char* e = getenv("VSYS_FLASH_W64"); return e && e[0] == '0'; }();
  constexpr int W64_DEFAULT_VAR = 4;   // 141 / 144 select placement variant 1 / 4 (4: +0.5-1 %, profiles/r04_flash64_w64_cvx5b.json)
  // k_bound > 0 (vsys_flash_attn_d64_kb): the statement without the running max; 17 forces it, 19 ignores the promise
  static const bool static_ok = [] { const char* e = getenv("VSYS_FLASH_STATIC"); return !(e && e[0] == '0'); }();
  const bool bounded = k_bound > 0.f && ln_w != nullptr && static_ok && fv != 19;   // no q / k LayerNorm: the bound is ignored (as d72 does)
  if (((fv == 0 && kv_len >= 2048 && !w64_off) || fv == 14 || fv == 17 || fv == 141 || fv == 144) && flash64_w64_supports(q_len, kv_len)) {
    int var = fv >= 140 ? fv - 140 : W64_DEFAULT_VAR;
    if (bounded && (fv == 0 || fv == 17)) var = 5;
    if (fv == 17 && var != 5) return VSYS_ERR_ARG;
    return launch_flash_attn_d64_w64(q, q_stride, ln_w, ln_b, rope_cos, rope_sin, rope_start, rope_len, kp, vt, out, out_stride, batch,
                                     heads, q_len, kv_len, kv_pad, eps, var, k_bound, stream);
  }
  if (fv == 12)        // A/B id (see set_flash_variant): the two-stage ring
    hipLaunchKernelGGL(flash_attn_d64_kernel<2>, dim3((unsigned)nblk), dim3(256), 2 * KV_STAGE, stream, p);
  else
    hipLaunchKernelGGL(flash_attn_d64_kernel<3>, dim3((unsigned)nblk), dim3(256), 3 * KV_STAGE, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

}  // namespace vsys
