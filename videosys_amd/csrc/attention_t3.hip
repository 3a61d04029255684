// attn_temporal_d72_v3: temporal self-attention (sequence = the T <= 32 frames of one pixel token, head dim 72) on the matrix
// pipe, operands loaded from HBM straight into MFMA fragment layout.
//
// Replaces (/root/reference/videosys): models/modules/attentions.py:55-120 for the temporal blocks — the two `(B S) T C` rearranges
// (open_sora_transformer_3d.py:204,206), LlamaRMSNorm on q and k (modules/normalization.py:19-33), RotaryEmbedding on q and k
// (rotary_embedding_torch, attentions.py:76-78) and native_attention (:111-120: q * scale in the activation dtype, fp32 softmax,
// probabilities cast back to the activation dtype before P V).  Same contract as attn_temporal_d72 (attention.hip); Latte calls it
// without norm / RoPE.
//
// Why: attn_temporal_d72_v2 parks K / V in LDS as fp32 and every lane re-reads all of it (228 KiB of LDS reads and ~2600 VALU
// instructions per (b, s, h) problem): it runs at 2.0-2.3 TB/s of algorithmic traffic, LDS- and VALU-bound
// (profiles/r01_run4_pmc_hot_kernels.txt).  Here a problem is two tiny matrix products:
//   S^T = K Q^T   5 x v_mfma_f32_32x32x16_bf16 (d = 72 -> 80; A = K rows, B = Q rows: a lane's 16-byte loads ARE the fragments)
//   O^T = V^T P^T 6 x v_mfma_f32_32x32x16_bf16 (3 blocks of 32 output dims x 2 chunks of 16 keys)
// The key order of the A operand of the first product is permuted (row m of the MFMA holds key 8*((m>>2)&1) + r + 8*(r>=8),
// r = (m&3) + 4*(m>>3)) so that the 16 accumulators of a lane are keys 8hi..8hi+7 and 16+8hi..16+8hi+7: packed to bf16 they are
// exactly the B fragments of the second product — the probabilities never cross lanes.  V is the one operand whose natural layout
// (dims contiguous) is not a fragment layout: each lane writes its row transposed into a 7.5 KiB wave-private LDS image [d][key]
// (80-byte pitch: conflict-free ds_read_b128) and reads the V^T fragments back.  RoPE angles come from a compact [T][36] table in
// LDS (the reference's table repeats every value twice), norm weights likewise: ~30 KiB of LDS reads per problem instead of 228.
//
// One workgroup = 4 waves = one pixel token (b, s); wave w walks heads w, w+4, ...: the 16 heads of a token read the same 6.9 KB
// qkv rows, so the rows stay in L1 / L2 between waves.  HBM-bound: 10.9 KB per problem in and out.
#include "common.h"
#include "vsys_internal.h"

namespace vsys {
namespace {

constexpr int HD = 72;
constexpr int VT_PITCH = 80;               // bytes per output-dim row of the V^T image: 32 keys x 2 B + 16 (5 r + chunk mod 16 distinct)
constexpr int VT_BYTES = 96 * VT_PITCH;    // 7680
constexpr int TAB_ROW = 36;                // floats per position in the compact cos / sin tables
constexpr int TAB_BYTES = 32 * TAB_ROW * 4;  // 4608
constexpr int W_BYTES = 160;               // one norm-weight vector (72 bf16, padded)
constexpr int WF_BYTES = 320;              // the same vector as fp32 (72 floats, padded): the fused path multiplies unpacked pairs by it
constexpr int LDS_HEAD = 2 * TAB_BYTES + 2 * W_BYTES + 2 * WF_BYTES;  // 10176
constexpr float NEG_BIG = -1e30f;

typedef __attribute__((ext_vector_type(2))) float t3_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 t3_bf16x2;
// two fp32 -> one dword of two bf16 (round to nearest even): ONE v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(t3_f32x2{lo, hi}, t3_bf16x2));
}

// One 16-byte piece (8 dims = 4 rotary pairs) of a q / k row: RMS-norm, weight, rotation, q scale -> the fragment dwords.
// RP = true keeps every rounding point of the reference's bf16 run (x * rstd -> bf16 (normalization.py:32), * weight -> bf16 (:33),
// rotation -> bf16 (rotary_embedding_torch, attentions.py:76-78), q * scale -> bf16 (:113)): each stage unpacks its pairs and repacks
// them, 17 (k) / 21 (q) VALU instructions per dword — 340 of the 653 VALU instructions of a head iteration were these conversions
// (profiles/r03_isa_mix_attention_t3.txt), and the kernel is VALU-bound.  RP = false (default) carries the pair in fp32 from the
// unpack to ONE rounding in front of the matrix product: x * (rstd [* scale]) * w, rotated, as four packed fp32 instructions + one
// v_cvt_pk_bf16_f32 = 7 per dword.  The chain is linear, so the q scale rides on rstd.  Fewer roundings than the reference: the
// distance to the fp32 oracle shrinks; same tolerance, not the same bits as RP (flash variant 21 selects RP for A/B).
// MODE: 0 = RP with run-time norm / rope flags; fused forms with the flags at COMPILE time (as run-time conditions hipcc computes both
// sides and selects: 14 instructions per dword): 1 = norm + rope (Open-Sora), 2 = neither (Latte), 3 = norm only, 4 = rope only.
template <int MODE>
__device__ __forceinline__ void t3_piece(uint32_t (&u)[4], const bf16_t* w, const float* wf, const float* cosrow, const float* sinrow,
                                         int c, int hi, float rstd, float scale, bool has_norm, bool has_rope, bool scaled) {
  if constexpr (MODE == 0) {
    if (has_norm) {
      const uint4 wv = *reinterpret_cast<const uint4*>(w + 16 * c + 8 * hi);
      const uint32_t wu[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t n1 = pk_bf16(bflo(u[e]) * rstd, bfhi(u[e]) * rstd);           // .to(input_dtype) (normalization.py:32)
        u[e] = pk_bf16(bflo(n1) * bflo(wu[e]), bfhi(n1) * bfhi(wu[e]));              // weight * (...)   (:33)
      }
    }
    if (has_rope) {
      const float4 cs = *reinterpret_cast<const float4*>(cosrow + 8 * c + 4 * hi);
      const float4 sn = *reinterpret_cast<const float4*>(sinrow + 8 * c + 4 * hi);
      const float cv[4] = {cs.x, cs.y, cs.z, cs.w}, sv[4] = {sn.x, sn.y, sn.z, sn.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {   // (x0, x1) * c + (-x1, x0) * s as one packed multiply + one packed FMA
        const float x0 = bflo(u[e]), x1 = bfhi(u[e]);
        const t3_f32x2 r = __builtin_elementwise_fma(t3_f32x2{-x1, x0}, t3_f32x2{sv[e], sv[e]}, t3_f32x2{x0, x1} * t3_f32x2{cv[e], cv[e]});
        u[e] = pk_bf16(r.x, r.y);
      }
    }
    if (scaled) {
#pragma unroll
      for (int e = 0; e < 4; ++e) u[e] = pk_bf16(bflo(u[e]) * scale, bfhi(u[e]) * scale);   // q = q * self.scale (attentions.py:113)
    }
  } else {
    constexpr bool NORM = MODE == 1 || MODE == 3, ROPE = MODE == 1 || MODE == 4;
    if (!NORM && !ROPE && !scaled) return;     // (Latte keys: the raw bf16 pairs are the fragment)
    const float rs = scaled ? rstd * scale : rstd;
    float wv[8], cv[4], sv[4];
    if constexpr (NORM) {
      const float4 wa = *reinterpret_cast<const float4*>(wf + 16 * c + 8 * hi), wb = *reinterpret_cast<const float4*>(wf + 16 * c + 8 * hi + 4);
      wv[0] = wa.x; wv[1] = wa.y; wv[2] = wa.z; wv[3] = wa.w; wv[4] = wb.x; wv[5] = wb.y; wv[6] = wb.z; wv[7] = wb.w;
    }
    if constexpr (ROPE) {
      const float4 cs = *reinterpret_cast<const float4*>(cosrow + 8 * c + 4 * hi);
      const float4 sn = *reinterpret_cast<const float4*>(sinrow + 8 * c + 4 * hi);
      cv[0] = cs.x; cv[1] = cs.y; cv[2] = cs.z; cv[3] = cs.w; sv[0] = sn.x; sv[1] = sn.y; sv[2] = sn.z; sv[3] = sn.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      t3_f32x2 x = t3_f32x2{bflo(u[e]), bfhi(u[e])} * t3_f32x2{rs, rs};
      if constexpr (NORM) x = x * t3_f32x2{wv[2 * e], wv[2 * e + 1]};
      if constexpr (ROPE) {
        // (x0, x1) c + (-x1, x0) s in ONE packed FMA: source 0 is read with its halves swapped (op_sel) and the low result's copy
        // negated (neg_lo) — written out by hand, hipcc lowers the swapped pair to a v_xor + v_mov in front of the FMA
        // The sine comes in as the register PAIR the table read delivered (elements e & ~1, e | 1) and op_sel / op_sel_hi pick the
        // same half for both results: duplicating it into a pair of its own cost two v_mov per dword (80 of the ~750 VALU
        // instructions of a head).
        const t3_f32x2 t = x * t3_f32x2{cv[e], cv[e]}, s2 = t3_f32x2{sv[e & ~1], sv[e | 1]};
        t3_f32x2 r;
        if (e & 1) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(x), "v"(s2), "v"(t));
        else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(x), "v"(s2), "v"(t));
        x = r;
      }
      u[e] = pk_bf16(x.x, x.y);
    }
  }
}

// The fused forms of the two production combinations (MODE 1: Open-Sora, 2: Latte) fit 128 registers: FOUR workgroups per CU when the
// launch's LDS allows it (the RoPE tables are sized by T, not by the 32-row tile: 37 KB per workgroup at T = 19) — the kernel waits
// on one HBM round trip per head with no prefetch, so resident waves are what covers it.
template <int MODE>
__global__ __launch_bounds__(256, (MODE == 1 || MODE == 2) ? 4 : 3) void attn_temporal_d72_v3_kernel(
    const bf16_t* __restrict__ qkv, int64_t row_stride, int C, const bf16_t* __restrict__ q_norm_w, const bf16_t* __restrict__ k_norm_w,
    const float* __restrict__ rope_cos, const float* __restrict__ rope_sin, bf16_t* __restrict__ out, int64_t out_stride, int B, int T,
    int S, int heads, float eps, float scale, int hsplit, int tab_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  float* cosc = reinterpret_cast<float*>(smem);
  // (tab_bytes = the bytes of one compact RoPE table for THIS launch's T, a multiple of 16; TAB_BYTES is the 32-row maximum)
  float* sinc = reinterpret_cast<float*>(smem + tab_bytes);
  bf16_t* qw = reinterpret_cast<bf16_t*>(smem + 2 * tab_bytes);
  bf16_t* kw = reinterpret_cast<bf16_t*>(smem + 2 * tab_bytes + W_BYTES);
  float* qwf = reinterpret_cast<float*>(smem + 2 * tab_bytes + 2 * W_BYTES);
  float* kwf = reinterpret_cast<float*>(smem + 2 * tab_bytes + 2 * W_BYTES + WF_BYTES);
  char* vt = smem + 2 * tab_bytes + 2 * W_BYTES + 2 * WF_BYTES + wave * VT_BYTES;
  const bool has_norm = q_norm_w != nullptr, has_rope = rope_cos != nullptr;

  // ---- per-workgroup tables (compact RoPE angles, norm weights) and the zeroed V^T image (key columns >= T stay zero for good:
  // they meet probability 0 in the second product and must be finite)
  if (has_rope) {
    for (int i = tid; i < T * TAB_ROW; i += 256) {
      const int t = i / TAB_ROW, j = i - t * TAB_ROW;
      cosc[i] = rope_cos[t * HD + 2 * j];
      sinc[i] = rope_sin[t * HD + 2 * j];
    }
  }
  if (has_norm && tid < HD) {
    qw[tid] = q_norm_w[tid];
    kw[tid] = k_norm_w[tid];
    qwf[tid] = bf2f(q_norm_w[tid]);
    kwf[tid] = bf2f(k_norm_w[tid]);
  }
  for (int i = lane * 16; i < VT_BYTES; i += 64 * 16) *reinterpret_cast<uint4*>(vt + i) = make_uint4(0, 0, 0, 0);
  __syncthreads();

  // hsplit workgroups share a pixel token: workgroup (bs, hg) owns the heads hg*hpg .. hg*hpg + hpg - 1 (small grids: one rank
  // of a sequence-parallel run has B*S = 256 tokens, one workgroup per CU would leave a wave per SIMD walking four heads serially)
  const int64_t bs = blockIdx.x / hsplit;
  const int hg = (int)(blockIdx.x - bs * hsplit);
  const int hpg = (heads + hsplit - 1) / hsplit;
  const int h_end = (hg + 1) * hpg < heads ? (hg + 1) * hpg : heads;
  const int s = (int)(bs % S), b = (int)(bs / S);
  // A-operand row m = l31 of the first product holds this key (see header); query / value rows are natural
  const int r_of_m = (l31 & 3) + 4 * (l31 >> 3);
  const int krow = 8 * ((l31 >> 2) & 1) + r_of_m + (r_of_m >= 8 ? 8 : 0);
  const bool q_ok = l31 < T, k_ok = krow < T;
  const int qrow_c = q_ok ? l31 : T - 1, krow_c = k_ok ? krow : T - 1;
  const bf16_t* qbase = qkv + (((int64_t)b * T + qrow_c) * S + s) * row_stride + 8 * hi;          // q and v of row l31
  const bf16_t* kbase = qkv + (((int64_t)b * T + krow_c) * S + s) * row_stride + C + 8 * hi;      // k of row key(l31)
  bf16_t* obase = out + (((int64_t)b * T + qrow_c) * S + s) * out_stride + 8 * hi;

  // norm + RoPE (+ q scale) of one row's pieces held by this lane, on PACKED bf16 pairs: every rounding point of the reference
  // (x * rstd -> bf16, * weight -> bf16, rotation -> bf16, q * scale -> bf16) is one v_cvt_pk_bf16_f32 on a pair, and the last
  // one IS the fragment dword.  raw[c] = 8 values at dims 16c + 8hi .. +7 (piece 4 of the hi half is padding).
  auto make_frags = [&](const uint4 (&raw)[5], bf16x8 (&frag)[5], const bf16_t* w, const float* wf, int pos, bool zero_row, bool scaled) {
    float rstd = 1.f;
    if (has_norm) {
      // sum of squares on the packed pairs: v_dot2c_f32_bf16 (lo*lo + hi*hi + acc in fp32) — one instruction per dword instead of
      // two unpacks and two FMAs
      float ss = 0.f;
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        const uint32_t u[4] = {raw[c].x, raw[c].y, raw[c].z, raw[c].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const t3_bf16x2 pr = __builtin_bit_cast(t3_bf16x2, u[e]);
          ss = __builtin_amdgcn_fdot2_f32_bf16(pr, pr, ss, false);
        }
      }
      ss += __shfl_xor(ss, 32, 64);
      rstd = rsqrtf(ss * (1.0f / (float)HD) + eps);   // (a multiply, not the division sequence)
    }
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      uint32_t u[4] = {raw[c].x, raw[c].y, raw[c].z, raw[c].w};
      if (c < 4 || hi == 0)
        t3_piece<MODE>(u, w, wf, cosc + pos * TAB_ROW, sinc + pos * TAB_ROW, c, hi, rstd, scale, has_norm, has_rope, scaled);
      // (the fused path reads 16 table floats per piece: left alone, hipcc hoists the reads of all five pieces of q AND k to the top
      //  and the kernel spills at its three-waves-per-SIMD budget; one piece's reads at a time keep it at the old footprint)
      if constexpr (MODE != 0) __builtin_amdgcn_sched_barrier(0);
      if (zero_row) u[0] = u[1] = u[2] = u[3] = 0u;
      frag[c] = __builtin_bit_cast(bf16x8, make_uint4(u[0], u[1], u[2], u[3]));
    }
  };

  for (int h = hg * hpg + wave; h < h_end; h += 4) {
    // ---- one round trip: the 4 / 5 sixteen-byte pieces of this lane's q, k and v rows
    uint4 rq[5], rk[5], rv[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      rq[c] = rk[c] = rv[c] = make_uint4(0, 0, 0, 0);
      if (c < 4 || hi == 0) {
        rk[c] = *reinterpret_cast<const uint4*>(kbase + h * HD + 16 * c);
        rv[c] = *reinterpret_cast<const uint4*>(qbase + 2 * C + h * HD + 16 * c);
        rq[c] = *reinterpret_cast<const uint4*>(qbase + h * HD + 16 * c);
      }
    }
    // ---- V: transposed into the wave's LDS image (lanes of real frames only)
    if (q_ok) {
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        if (c < 4 || hi == 0) {
          const uint32_t u[4] = {rv[c].x, rv[c].y, rv[c].z, rv[c].w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const bf16_t val = (bf16_t)((e & 1) ? (u[e >> 1] >> 16) : (u[e >> 1] & 0xffffu));
            *reinterpret_cast<bf16_t*>(vt + (16 * c + 8 * hi + e) * VT_PITCH + l31 * 2) = val;
          }
        }
      }
    }
    // ---- K, then Q: norm, RoPE, (q: scale) -> fragments
    bf16x8 kf[5], qf[5];
    make_frags(rk, kf, kw, kwf, krow_c, !k_ok, false);
    make_frags(rq, qf, qw, qwf, qrow_c, false, true);
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int c = 0; c < 5; ++c) sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[c], qf[c], sacc, 0, 0, 0);
    // ---- softmax over the keys of this lane's query (16 here, 16 in lane ^ 32); accumulator r <-> key 8hi + r + 8 (r >= 8)
    float m = NEG_BIG;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = 8 * hi + r + (r >= 8 ? 8 : 0);
      if (key >= T) sacc[r] = NEG_BIG;
      m = fmaxf(m, sacc[r]);
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float p[16], l = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = __builtin_amdgcn_exp2f((sacc[r] - m) * 1.4426950408889634f);
      l += p[r];
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = __builtin_amdgcn_rcpf(l);   // v_rcp_f32 (1 ulp) instead of the IEEE division sequence
    bf16x8 p0, p1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      p0[e] = (__bf16)(p[e] * inv);        // attn.to(dtype) before attn @ v (:117)
      p1[e] = (__bf16)(p[8 + e] * inv);
    }
    // ---- O^T = V^T P^T
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    f32x16 oacc[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(vt + (d * 32 + l31) * VT_PITCH + hi * 16);
      const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(vt + (d * 32 + l31) * VT_PITCH + 32 + hi * 16);
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
      oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, p0, oacc[d], 0, 0, 0);
      oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, p1, oacc[d], 0, 0, 0);
    }
    // the image is rewritten by the next head: its reads above must have been issued (in-order LDS) — and the compiler must not
    // move the next iteration's writes up
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // ---- store: accumulator 4g + r' of block d <-> dim 32d + 8g + 4hi + r'; swap pairs of groups -> 16 contiguous bytes per lane
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      uint2 o[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        o[g].x = pack2bf(oacc[d][4 * g], oacc[d][4 * g + 1]);
        o[g].y = pack2bf(oacc[d][4 * g + 2], oacc[d][4 * g + 3]);
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        auto sx = __builtin_amdgcn_permlane32_swap(o[2 * k].x, o[2 * k + 1].x, false, false);
        auto sy = __builtin_amdgcn_permlane32_swap(o[2 * k].y, o[2 * k + 1].y, false, false);
        const int dim0 = 32 * d + 16 * k;   // + 8 hi inside obase
        if (q_ok && dim0 + 8 * hi + 8 <= HD)
          *reinterpret_cast<uint4*>(obase + h * HD + dim0) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// attn_temporal_d72_v4: the same formulation for 32 < T <= 64 frames (720p x 128f: T = 38, which attn_temporal_d72_v3 cannot hold in
// one 32-row tile and which ran on the VALU kernel at 1.0 TB/s, 8 % of that step: profiles/r04_720p128f_kernel_stats.txt).
// Two key blocks and two query blocks of 32: K fragments of both key blocks are built once per head, then per query block
//   S^T = K Q^T     2 x 5 v_mfma_f32_32x32x16_bf16  (key block kb: accumulator r of a lane <-> key 32 kb + 8 hi + r + 8 (r >= 8))
//   O^T = V^T P^T   3 x 4                           (four 16-key chunks: chunk 2 kb + j = accumulators 8 j .. 8 j + 7 of key block kb)
// V^T image per wave: [96 dims][64 keys], 144-byte pitch (13.5 KiB); compact RoPE table for 64 positions: two workgroups per CU.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int VT4_PITCH = 144;
constexpr int VT4_BYTES = 96 * VT4_PITCH;    // 13824
constexpr int TAB4_BYTES = 64 * TAB_ROW * 4;  // 9216
constexpr int LDS4_HEAD = 2 * TAB4_BYTES + 2 * W_BYTES + 2 * WF_BYTES;

template <int MODE>
__global__ __launch_bounds__(256, 2) void attn_temporal_d72_v4_kernel(
    const bf16_t* __restrict__ qkv, int64_t row_stride, int C, const bf16_t* __restrict__ q_norm_w, const bf16_t* __restrict__ k_norm_w,
    const float* __restrict__ rope_cos, const float* __restrict__ rope_sin, bf16_t* __restrict__ out, int64_t out_stride, int B, int T,
    int S, int heads, float eps, float scale, int hsplit) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  float* cosc = reinterpret_cast<float*>(smem);
  float* sinc = reinterpret_cast<float*>(smem + TAB4_BYTES);
  bf16_t* qw = reinterpret_cast<bf16_t*>(smem + 2 * TAB4_BYTES);
  bf16_t* kw = reinterpret_cast<bf16_t*>(smem + 2 * TAB4_BYTES + W_BYTES);
  float* qwf = reinterpret_cast<float*>(smem + 2 * TAB4_BYTES + 2 * W_BYTES);
  float* kwf = reinterpret_cast<float*>(smem + 2 * TAB4_BYTES + 2 * W_BYTES + WF_BYTES);
  char* vt = smem + LDS4_HEAD + wave * VT4_BYTES;
  const bool has_norm = q_norm_w != nullptr, has_rope = rope_cos != nullptr;
  if (has_rope) {
    for (int i = tid; i < T * TAB_ROW; i += 256) {
      const int t = i / TAB_ROW, j = i - t * TAB_ROW;
      cosc[i] = rope_cos[t * HD + 2 * j];
      sinc[i] = rope_sin[t * HD + 2 * j];
    }
  }
  if (has_norm && tid < HD) {
    qw[tid] = q_norm_w[tid];
    kw[tid] = k_norm_w[tid];
    qwf[tid] = bf2f(q_norm_w[tid]);
    kwf[tid] = bf2f(k_norm_w[tid]);
  }
  for (int i = lane * 16; i < VT4_BYTES; i += 64 * 16) *reinterpret_cast<uint4*>(vt + i) = make_uint4(0, 0, 0, 0);
  __syncthreads();

  const int64_t bs = blockIdx.x / hsplit;
  const int hg = (int)(blockIdx.x - bs * hsplit);
  const int hpg = (heads + hsplit - 1) / hsplit;
  const int h_end = (hg + 1) * hpg < heads ? (hg + 1) * hpg : heads;
  const int s = (int)(bs % S), b = (int)(bs / S);
  const int r_of_m = (l31 & 3) + 4 * (l31 >> 3);
  const int krow0 = 8 * ((l31 >> 2) & 1) + r_of_m + (r_of_m >= 8 ? 8 : 0);   // key of MFMA row l31 inside a 32-key block
  int qrow[2], krow[2];
  bool q_ok[2], k_ok[2];
  const bf16_t* qbase[2];
  const bf16_t* kbase[2];
  bf16_t* obase[2];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    q_ok[blk] = 32 * blk + l31 < T;
    k_ok[blk] = 32 * blk + krow0 < T;
    qrow[blk] = q_ok[blk] ? 32 * blk + l31 : T - 1;
    krow[blk] = k_ok[blk] ? 32 * blk + krow0 : T - 1;
    qbase[blk] = qkv + (((int64_t)b * T + qrow[blk]) * S + s) * row_stride + 8 * hi;
    kbase[blk] = qkv + (((int64_t)b * T + krow[blk]) * S + s) * row_stride + C + 8 * hi;
    obase[blk] = out + (((int64_t)b * T + qrow[blk]) * S + s) * out_stride + 8 * hi;
  }

  // norm + RoPE (+ q scale) of one row's pieces on packed bf16 pairs: as attn_temporal_d72_v3_kernel::make_frags
  auto make_frags = [&](const uint4 (&raw)[5], bf16x8 (&frag)[5], const bf16_t* w, const float* wf, int pos, bool zero_row, bool scaled) {
    float rstd = 1.f;
    if (has_norm) {
      float ss = 0.f;
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        const uint32_t u[4] = {raw[c].x, raw[c].y, raw[c].z, raw[c].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const t3_bf16x2 pr = __builtin_bit_cast(t3_bf16x2, u[e]);
          ss = __builtin_amdgcn_fdot2_f32_bf16(pr, pr, ss, false);
        }
      }
      ss += __shfl_xor(ss, 32, 64);
      rstd = rsqrtf(ss * (1.0f / (float)HD) + eps);   // (a multiply, not the division sequence)
    }
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      uint32_t u[4] = {raw[c].x, raw[c].y, raw[c].z, raw[c].w};
      if (c < 4 || hi == 0)
        t3_piece<MODE>(u, w, wf, cosc + pos * TAB_ROW, sinc + pos * TAB_ROW, c, hi, rstd, scale, has_norm, has_rope, scaled);
      // (the fused path reads 16 table floats per piece: left alone, hipcc hoists the reads of all five pieces of q AND k to the top
      //  and the kernel spills at its three-waves-per-SIMD budget; one piece's reads at a time keep it at the old footprint)
      if constexpr (MODE != 0) __builtin_amdgcn_sched_barrier(0);
      if (zero_row) u[0] = u[1] = u[2] = u[3] = 0u;
      frag[c] = __builtin_bit_cast(bf16x8, make_uint4(u[0], u[1], u[2], u[3]));
    }
  };

  for (int h = hg * hpg + wave; h < h_end; h += 4) {
    // ---- one round trip: q, k, v pieces of both 32-row blocks
    uint4 rq[2][5], rk[2][5], rv[2][5];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        rq[blk][c] = rk[blk][c] = rv[blk][c] = make_uint4(0, 0, 0, 0);
        if (c < 4 || hi == 0) {
          rk[blk][c] = *reinterpret_cast<const uint4*>(kbase[blk] + h * HD + 16 * c);
          rv[blk][c] = *reinterpret_cast<const uint4*>(qbase[blk] + 2 * C + h * HD + 16 * c);
          rq[blk][c] = *reinterpret_cast<const uint4*>(qbase[blk] + h * HD + 16 * c);
        }
      }
    // ---- V: transposed into the wave's LDS image [dim][key] (real frames only)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      if (q_ok[blk]) {
#pragma unroll
        for (int c = 0; c < 5; ++c) {
          if (c < 4 || hi == 0) {
            const uint32_t u[4] = {rv[blk][c].x, rv[blk][c].y, rv[blk][c].z, rv[blk][c].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const bf16_t val = (bf16_t)((e & 1) ? (u[e >> 1] >> 16) : (u[e >> 1] & 0xffffu));
              *reinterpret_cast<bf16_t*>(vt + (16 * c + 8 * hi + e) * VT4_PITCH + (32 * blk + l31) * 2) = val;
            }
          }
        }
      }
    }
    // ---- K fragments of both key blocks
    bf16x8 kf[2][5];
    make_frags(rk[0], kf[0], kw, kwf, krow[0], !k_ok[0], false);
    make_frags(rk[1], kf[1], kw, kwf, krow[1], !k_ok[1], false);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      if (32 * qb >= T) break;          // (wave-uniform)
      bf16x8 qf[5];
      make_frags(rq[qb], qf, qw, qwf, qrow[qb], false, true);
      f32x16 sacc[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
        for (int c = 0; c < 5; ++c) sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb][c], qf[c], sacc[kb], 0, 0, 0);
      }
      // ---- softmax over the 64 keys of this lane's query (32 here, 32 in lane ^ 32)
      float m = NEG_BIG;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = 32 * kb + 8 * hi + r + (r >= 8 ? 8 : 0);
          if (key >= T) sacc[kb][r] = NEG_BIG;
          m = fmaxf(m, sacc[kb][r]);
        }
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      float l = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          sacc[kb][r] = __builtin_amdgcn_exp2f((sacc[kb][r] - m) * 1.4426950408889634f);
          l += sacc[kb][r];
        }
      l += __shfl_xor(l, 32, 64);
      const float inv = __builtin_amdgcn_rcpf(l);   // v_rcp_f32 (1 ulp) instead of the IEEE division sequence
      bf16x8 pf[4];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          pf[2 * kb][e] = (__bf16)(sacc[kb][e] * inv);          // attn.to(dtype) before attn @ v (attentions.py:117)
          pf[2 * kb + 1][e] = (__bf16)(sacc[kb][8 + e] * inv);
        }
      // ---- O^T = V^T P^T
      f32x16 oacc[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
          const bf16x8 vfrag = *reinterpret_cast<const bf16x8*>(vt + (d * 32 + l31) * VT4_PITCH + ch * 32 + hi * 16);
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfrag, pf[ch], oacc[d], 0, 0, 0);
        }
      }
      // ---- store (as attn_temporal_d72_v3_kernel)
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        uint2 o[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          o[g].x = pack2bf(oacc[d][4 * g], oacc[d][4 * g + 1]);
          o[g].y = pack2bf(oacc[d][4 * g + 2], oacc[d][4 * g + 3]);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          auto sx = __builtin_amdgcn_permlane32_swap(o[2 * k].x, o[2 * k + 1].x, false, false);
          auto sy = __builtin_amdgcn_permlane32_swap(o[2 * k].y, o[2 * k + 1].y, false, false);
          const int dim0 = 32 * d + 16 * k;
          if (q_ok[qb] && dim0 + 8 * hi + 8 <= HD)
            *reinterpret_cast<uint4*>(obase[qb] + h * HD + dim0) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
      }
    }
    // the image is rewritten by the next head: its reads above must have been issued, and the compiler must not move the next
    // iteration's writes up
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// attn_temporal_d72_v5 (round 6): the same arithmetic as attn_temporal_d72_v3 (same fragments, same MFMA order: same bits) with the
// HBM side rebuilt.  v3 loads every operand straight into fragment layout: a lane owns one frame, so ONE wave load touches T = 19
// rows that lie S * row_stride * 2 = 7 MB apart and takes 32 bytes from each — 15 loads + 9 stores per head, every one of them 19-38
// cache-line touches in the CU's texture path for 0.6-1.2 KB of payload.  r05 finding (f): -31 % VALU bought -3 % time; the kernel is
// bound by that load shape (3.4 TB/s).  Here a workgroup owns (pixel token, group of FOUR heads): per frame the q, k and v bytes of
// four heads are 576 contiguous bytes each, and the whole operand set of an item (3 T segments) comes in by LDS-DMA
// (buffer_load_dwordx4 ... lds) as ONE lane-linear image: LDS piece L = 16-byte piece (L % 37) of segment row (L / 37), 37 = 36 real
// pieces + 1 pad, i.e. a 592-byte row pitch (conflict-free for the 16-byte fragment reads: 592 / 4 = 148 = 20 mod 64 banks per row).
// 33 wave instructions of 64 pieces bring in the 32.8 KB of an item at T = 19 (v3: 60 instructions for the same four heads), each
// touching ~2 rows.  The waves then read their K / Q pieces as ds_read_b128 (the registers v3 got from HBM), gather the V^T fragments
// with 16-bit LDS reads straight from the row-major V rows (no transposed image), and the outputs go back through the (dead) Q rows
// of the image so that the workgroup stores 576-byte row segments with 16-byte stores.
// Rows of an image: [V: 0 .. T-1][K: T .. 2T-1][Q / O: 2T .. 3T-1]; the V^T gather of keys >= T runs on into the K / Q rows (finite
// values, met by probability 0); below 11 frames the rows behind 3T are zero-filled once.
// Workgroups are persistent (items blockIdx, + gridDim, ...: the tables are staged once) and four fit a CU at T = 19 (39.9 KB).
// What was measured and not kept (profiles/r06_temporal_v5_probe.txt; probe = tools/temporal_probe.py, v3 110-112 us on those boxes):
//   * the phases switched off one by one (lab builds, VSYS_T5_ABLATE): 100 us = LDS-DMA alone 50 (6.1 TB/s) / + stores 84 / arithmetic
//     alone 73: memory and arithmetic each fill ~80 % of the launch, and every workgroup of the launch sits in the same phase at
//     the same time (the DMA phase ends for everybody when HBM has delivered), so they overlap only partly;
//   * two images per workgroup with the next item's LDS-DMA under the current item's arithmetic (two workgroups per CU): 110 us —
//     the arithmetic alone takes 84 us at two waves per SIMD; a start-up stagger of the workgroups of a CU: 94-98 us (inside the noise);
//   * eight heads per workgroup (1152-byte segments = nine whole 128-byte lines, eight waves): 102 us, no gain over four;
//   * ds_read_u16_d16_hi for the V^T gather: the register's other half is NOT preserved on this part (SRAM ECC) — wrong fragments.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int V5_PIECES = 37;               // 16-byte pieces per image row (36 + 1 pad)
constexpr int V5_PITCH = V5_PIECES * 16;    // 592

template <int MODE, int NIW>
__global__ __launch_bounds__(256, 4) void attn_temporal_d72_v5_kernel(
    const bf16_t* __restrict__ qkv, int64_t row_stride, int C, const bf16_t* __restrict__ q_norm_w, const bf16_t* __restrict__ k_norm_w,
    const float* __restrict__ rope_cos, const float* __restrict__ rope_sin, bf16_t* __restrict__ out, int64_t out_stride, int B, int T,
    int S, int heads, float eps, float scale, int ninstr, int data_bytes, int tab_bytes, int nitems, int ablate) {
#if __HIP_DEVICE_COMPILE__
#ifdef VSYS_LAB
#define V5_ABL ablate
#else
#define V5_ABL 0
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HPG = 4, NTH = 256, NIMG = 1;   // heads (= waves) per workgroup, threads, images
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  // the image, then the tables
  float* cosc = reinterpret_cast<float*>(smem + data_bytes);
  float* sinc = reinterpret_cast<float*>(smem + data_bytes + tab_bytes);
  float* qwf = reinterpret_cast<float*>(smem + data_bytes + 2 * tab_bytes);
  float* kwf = reinterpret_cast<float*>(smem + data_bytes + 2 * tab_bytes + WF_BYTES);
  constexpr bool NORM = MODE == 1 || MODE == 3, ROPE = MODE == 1 || MODE == 4;

  // ---- item-independent addressing.  LDS-DMA: instruction ii = wave + HPG i of the workgroup fills pieces 64 ii .. 64 ii + 63
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int frame_bytes = S * (int)row_stride * 2;       // consecutive frames of one pixel token (launcher: T * this < 2^31)
  int voff[NIW];
#pragma unroll
  for (int i = 0; i < NIW; ++i) {
    const int L = lane + 64 * (wave_u + HPG * i);
    int row = L / V5_PIECES, piece = L - row * V5_PIECES;
    piece = piece < V5_PIECES - 1 ? piece : V5_PIECES - 2;                     // the pad slot receives a copy of the last piece
    row = row < 3 * T ? row : 3 * T - 1;                 // pieces behind the image land in the spare tail of the data region
    const int seg = row / T, t = row - seg * T;          // 0 = V, 1 = K, 2 = Q
    voff[i] = t * frame_bytes + (seg == 0 ? 2 * C : (seg == 1 ? C : 0)) * 2 + piece * 16;
  }
  const int HG = heads / HPG;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  // LDS-DMA of one item into image ``img`` (this wave owns instructions wave, wave + 4, ... of the workgroup's ninstr)
  auto dma_item = [&](int item, int) {
    const int bs = item / HG, hg = item - bs * HG;
    const int s = bs % S, b = bs / S;
    const bf16_t* ibase = qkv + ((int64_t)b * T * S + s) * row_stride + hg * (HPG * HD);
    const int ibytes = (T - 1) * frame_bytes + (2 * C + HPG * HD) * 2;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ibase, 0, ibytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < NIW; ++i) {
      const int ii = wave_u + HPG * i;
      if (ii < ninstr && !(V5_ABL & 1))
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(smem + ii * 1024), 16, voff[i], 0, 0, 0);
    }
  };
  const int lw = xcd_remap(blockIdx.x, gridDim.x);       // neighbouring items (head groups of one token, then the next token) share an XCD's L2
  if (lw < nitems) dma_item(lw, 0);                       // the first image travels while the tables are built

  if (ROPE) {
    for (int i = tid; i < T * TAB_ROW; i += NTH) {        // compact table: entry i = (t, j) <-> element 2 i of the [T][72] table
      cosc[i] = rope_cos[2 * i];
      sinc[i] = rope_sin[2 * i];
    }
  }
  if (NORM && tid < HD) {
    qwf[tid] = bf2f(q_norm_w[tid]);
    kwf[tid] = bf2f(k_norm_w[tid]);
  }
  // rows behind the 3 T real ones (reached by the V^T gather below 11 frames) are never written by a real piece: zero them once
  for (int i = 3 * T * V5_PITCH + tid * 16; i < data_bytes; i += NTH * 16) *reinterpret_cast<uint4*>(smem + i) = make_uint4(0, 0, 0, 0);

  // A-operand row m = l31 of the first product holds this key (attn_temporal_d72_v3_kernel); query / value rows are natural
  const int r_of_m = (l31 & 3) + 4 * (l31 >> 3);
  const int krow = 8 * ((l31 >> 2) & 1) + r_of_m + (r_of_m >= 8 ? 8 : 0);
  const bool q_ok = l31 < T, k_ok = krow < T;
  const int qrow_c = q_ok ? l31 : T - 1, krow_c = k_ok ? krow : T - 1;
  const int hoff = wave * (HD * 2);                      // this wave's head inside the 576-byte segments
  const int koff = (T + krow_c) * V5_PITCH + hoff + 16 * hi;
  const int qoff = (2 * T + qrow_c) * V5_PITCH + hoff + 16 * hi;      // Q in, O out
  const int voffl = hoff + l31 * 2 + hi * 8 * V5_PITCH;  // + 64 d + (16 ch + j) * 592: V^T element (dim 32 d + l31, key 16 ch + 8 hi + j)
  const int npieces_out = T * (V5_PIECES - 1);

  auto make_frags = [&](const uint4 (&raw)[5], bf16x8 (&frag)[5], const float* wf, int pos, bool scaled) {
    float rstd = 1.f;
    if (NORM) {
      float ss = 0.f;
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        const uint32_t u[4] = {raw[c].x, raw[c].y, raw[c].z, raw[c].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const t3_bf16x2 pr = __builtin_bit_cast(t3_bf16x2, u[e]);
          ss = __builtin_amdgcn_fdot2_f32_bf16(pr, pr, ss, false);
        }
      }
      ss += __shfl_xor(ss, 32, 64);
      rstd = rsqrtf(ss * (1.0f / (float)HD) + eps);   // (a multiply, not the division sequence)
    }
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      uint32_t u[4] = {raw[c].x, raw[c].y, raw[c].z, raw[c].w};
      if (c < 4 || hi == 0)
        t3_piece<MODE>(u, nullptr, wf, cosc + pos * TAB_ROW, sinc + pos * TAB_ROW, c, hi, rstd, scale, NORM, ROPE, scaled);
      __builtin_amdgcn_sched_barrier(0);   // (128 registers: one piece's table reads at a time)
      frag[c] = __builtin_bit_cast(bf16x8, make_uint4(u[0], u[1], u[2], u[3]));
    }
  };

  // the store phase of an item: the workgroup stores T row segments of 576 bytes out of the Q / O rows of ``img``
  auto store_item = [&](int item, const char* img) {
    const int bs = item / HG, hg = item - bs * HG;
    const int s = bs % S, b = bs / S;
    bf16_t* obase = out + ((int64_t)b * T * S + s) * out_stride + hg * (HPG * HD);
    for (int Lp = tid; Lp < npieces_out; Lp += NTH) {
      const int row = Lp / (V5_PIECES - 1), piece = Lp - row * (V5_PIECES - 1);
      const uint4 v = *reinterpret_cast<const uint4*>(img + (2 * T + row) * V5_PITCH + piece * 16);
      if (!(V5_ABL & 4)) *reinterpret_cast<uint4*>(obase + (int64_t)row * S * out_stride + piece * 8) = v;
    }
  };
  for (int item = lw; item < nitems; item += gridDim.x) {
    const int next = item + (int)gridDim.x;
    char* data = smem;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // this item's image has landed everywhere
    const char* kpiece = data + koff;
    char* qpiece = data + qoff;
    const char* vbase = data + voffl;

    if (!(V5_ABL & 2)) {   // (lab builds, VSYS_T5_ABLATE: 1 no LDS-DMA, 2 no arithmetic, 4 no stores — wrong output)
    // ---- this wave's head: K and Q pieces (the registers v3 loads from HBM)
    bf16x8 kf[5], qf[5];
    {
      uint4 rk[5];
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        rk[c] = make_uint4(0, 0, 0, 0);
        if (c < 4 || hi == 0) rk[c] = *reinterpret_cast<const uint4*>(kpiece + 32 * c);
      }
      make_frags(rk, kf, kwf, krow_c, false);   // (rows of keys >= T are the clamped last frame's: finite, and masked below)
    }
    {
      uint4 rq[5];
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        rq[c] = make_uint4(0, 0, 0, 0);
        if (c < 4 || hi == 0) rq[c] = *reinterpret_cast<const uint4*>(qpiece + 32 * c);
      }
      make_frags(rq, qf, qwf, qrow_c, true);
    }
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int c = 0; c < 5; ++c) sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[c], qf[c], sacc, 0, 0, 0);
    float m = NEG_BIG;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = 8 * hi + r + (r >= 8 ? 8 : 0);
      if (key >= T) sacc[r] = NEG_BIG;
      m = fmaxf(m, sacc[r]);
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float p[16], l = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = __builtin_amdgcn_exp2f((sacc[r] - m) * 1.4426950408889634f);
      l += p[r];
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = __builtin_amdgcn_rcpf(l);   // v_rcp_f32 (1 ulp) instead of the IEEE division sequence
    bf16x8 p0, p1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      p0[e] = (__bf16)(p[e] * inv);
      p1[e] = (__bf16)(p[8 + e] * inv);
    }
    f32x16 oacc[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      // V^T fragments gathered from the row-major V rows: element j <-> key 16 ch + 8 hi + j of dim 32 d + l31.  (Plain 16-bit LDS
      // reads + one v_perm per dword: ds_read_u16_d16_hi does NOT keep the other half of its register on this part — SRAM ECC —
      // so the in-place form gives wrong fragments; measured, reverted.)
      bf16x8 v0, v1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v0[j] = *reinterpret_cast<const __bf16*>(vbase + 64 * d + j * V5_PITCH);
        v1[j] = *reinterpret_cast<const __bf16*>(vbase + 64 * d + (16 + j) * V5_PITCH);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
      oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, p0, oacc[d], 0, 0, 0);
      oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, p1, oacc[d], 0, 0, 0);
    }
    // ---- outputs into this wave's own bytes of the Q rows (its Q pieces are in registers; no other wave touches that slice)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      uint2 o[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        o[g].x = pack2bf(oacc[d][4 * g], oacc[d][4 * g + 1]);
        o[g].y = pack2bf(oacc[d][4 * g + 2], oacc[d][4 * g + 3]);
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        auto sx = __builtin_amdgcn_permlane32_swap(o[2 * k].x, o[2 * k + 1].x, false, false);
        auto sy = __builtin_amdgcn_permlane32_swap(o[2 * k].y, o[2 * k + 1].y, false, false);
        const int dim0 = 32 * d + 16 * k;   // + 8 hi: the 16 hi bytes inside qpiece
        if (q_ok && dim0 + 8 * hi + 8 <= HD)
          *reinterpret_cast<uint4*>(qpiece + dim0 * 2) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
      }
    }
    }
    __syncthreads();
    store_item(item, data);
    __syncthreads();   // (every output piece has been read: the next item's image may land)
    if (next < nitems) dma_item(next, 0);
  }
#undef V5_ABL
#endif
}

}  // namespace

int launch_attn_temporal_d72_v3(const bf16_t* qkv, int64_t row_stride, int C, const bf16_t* q_norm_w, const bf16_t* k_norm_w,
                                const float* rope_cos, const float* rope_sin, bf16_t* out, int64_t out_stride, int B, int T, int S,
                                int heads, float eps, float scale, hipStream_t stream, bool ref_rounding, bool no_v5) {
  if (T > 64 || T < 1) return VSYS_ERR_SHAPE;
  // (t3_piece) 0 = the reference's rounding points with run-time flags; else the fused single-rounding form of this norm / rope combination
  const int mode = ref_rounding ? 0 : (q_norm_w != nullptr ? (rope_cos != nullptr ? 1 : 3) : (rope_cos != nullptr ? 4 : 2));
  if (T > 32) {   // two key / query blocks (attn_temporal_d72_v4_kernel)
    int hsplit4 = 1;
    const int64_t slots4 = 2LL * cu_count_this_device();
    while (hsplit4 < 4 && (int64_t)B * S * hsplit4 < slots4 && heads % (hsplit4 * 2 * 4) == 0) hsplit4 *= 2;
    const int64_t grid4 = (int64_t)B * S * hsplit4;
    if (grid4 > 0x7fffffff) return VSYS_ERR_SHAPE;
    const int lds4 = LDS4_HEAD + 4 * VT4_BYTES;   // 74688
    static std::atomic<unsigned long long> attr_seen{0};
    for (DeviceOnce once(attr_seen); once.todo(); once.done()) {
      (void)hipFuncSetAttribute((const void*)attn_temporal_d72_v4_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
      (void)hipFuncSetAttribute((const void*)attn_temporal_d72_v4_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
      (void)hipFuncSetAttribute((const void*)attn_temporal_d72_v4_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
      (void)hipFuncSetAttribute((const void*)attn_temporal_d72_v4_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
      (void)hipFuncSetAttribute((const void*)attn_temporal_d72_v4_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
    }
#define T3_LAUNCH4(M_)                                                                                                          \
  hipLaunchKernelGGL(attn_temporal_d72_v4_kernel<M_>, dim3((unsigned)grid4), dim3(256), lds4, stream, qkv, row_stride, C, q_norm_w, \
                     k_norm_w, rope_cos, rope_sin, out, out_stride, B, T, S, heads, eps, scale, hsplit4)
    switch (mode) {
      case 0: T3_LAUNCH4(0); break;
      case 1: T3_LAUNCH4(1); break;
      case 2: T3_LAUNCH4(2); break;
      case 3: T3_LAUNCH4(3); break;
      default: T3_LAUNCH4(4); break;
    }
#undef T3_LAUNCH4
    return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
  }
  // ---- round 6: the cooperative LDS-DMA form (attn_temporal_d72_v5_kernel) for the fused-rounding modes whenever the heads come in
  // groups of four and a pixel token's frames are addressable with 32-bit offsets (VSYS_TEMPORAL_V5=0 / flash variant 22: the v3 kernel)
  static const bool v5_env_ok = [] { const char* e = getenv("VSYS_TEMPORAL_V5"); return !(e && e[0] == '0'); }();
  const int64_t span = (int64_t)T * S * row_stride * 2;
  if (mode != 0 && mode != 3 /* norm without RoPE: nobody runs it, and it spills at four workgroups per CU */ && v5_env_ok && !no_v5 && heads % 4 == 0 && C % 8 == 0 && span < 0x7fffffff && (int64_t)T * S * out_stride * 2 < 0x7fffffff &&
      (int64_t)B * S * (heads / 4) <= 0x7fffffff) {
    const int rows = 3 * T > 35 ? 3 * T : 35;   // (V^T gather reaches image row 31 + the 144-byte overrun of the last head)
    const int ninstr = (3 * T * V5_PIECES + 63) / 64;
    const int ninstr_region = (rows * V5_PIECES + 63) / 64;
    const int data_bytes = (ninstr_region > ninstr ? ninstr_region : ninstr) * 1024;
    const int tabb = rope_cos != nullptr ? T * TAB_ROW * 4 : 0;
    const int lds5 = data_bytes + 2 * tabb + 2 * WF_BYTES;
    const int nitems = B * S * (heads / 4);
    int per_cu = 160 * 1024 / lds5;
    per_cu = per_cu > 4 ? 4 : per_cu;
    int64_t grid5 = (int64_t)per_cu * cu_count_this_device();
    grid5 = grid5 < nitems ? grid5 : nitems;
    const bool small = (ninstr + 3) / 4 <= 9;
    int ablate = 0;
#ifdef VSYS_LAB
    static const int ablate_env = [] { const char* e = getenv("VSYS_T5_ABLATE"); return e ? atoi(e) : 0; }();   // measurement only (wrong output)
    ablate = ablate_env;
#endif
    static std::atomic<unsigned long long> attr5_seen{0};
    for (DeviceOnce once(attr5_seen); once.todo(); once.done()) {
#define T3_ATTR5(M_)                                                                                                                    \
  (void)hipFuncSetAttribute((const void*)attn_temporal_d72_v5_kernel<M_, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);     \
  (void)hipFuncSetAttribute((const void*)attn_temporal_d72_v5_kernel<M_, 14>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)
      T3_ATTR5(1); T3_ATTR5(2); T3_ATTR5(4);
#undef T3_ATTR5
    }
#define T3_LAUNCH5_(M_, N_)                                                                                                               \
  hipLaunchKernelGGL((attn_temporal_d72_v5_kernel<M_, N_>), dim3((unsigned)grid5), dim3(256), lds5, stream, qkv, row_stride, C, q_norm_w,  \
                     k_norm_w, rope_cos, rope_sin, out, out_stride, B, T, S, heads, eps, scale, ninstr, data_bytes, tabb, nitems, ablate)
#define T3_LAUNCH5(M_) do { if (small) T3_LAUNCH5_(M_, 9); else T3_LAUNCH5_(M_, 14); } while (0)
    switch (mode) {
      case 1: T3_LAUNCH5(1); break;
      case 2: T3_LAUNCH5(2); break;
      default: T3_LAUNCH5(4); break;
    }
#undef T3_LAUNCH5
#undef T3_LAUNCH5_
    return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
  }
  // fewer than ~3 workgroups per CU: split the heads of a token over 2 or 4 workgroups (every wave still owns whole heads)
  int hsplit = 1;
  const int64_t slots = 4LL * cu_count_this_device();
  while (hsplit < 4 && (int64_t)B * S * hsplit < slots && heads % (hsplit * 2 * 4) == 0) hsplit *= 2;
  const int64_t grid = (int64_t)B * S * hsplit;
  if (grid > 0x7fffffff) return VSYS_ERR_SHAPE;
  const int tab_bytes = rope_cos != nullptr ? T * TAB_ROW * 4 : 0;   // (T x 144 B: a multiple of 16)
  const int lds = 2 * tab_bytes + 2 * W_BYTES + 2 * WF_BYTES + 4 * VT_BYTES;   // 37152 at T = 19 (four workgroups per CU), <= 40896
#define T3_LAUNCH3(M_)                                                                                                        \
  hipLaunchKernelGGL(attn_temporal_d72_v3_kernel<M_>, dim3((unsigned)grid), dim3(256), lds, stream, qkv, row_stride, C, q_norm_w, \
                     k_norm_w, rope_cos, rope_sin, out, out_stride, B, T, S, heads, eps, scale, hsplit, tab_bytes)
  switch (mode) {
    case 0: T3_LAUNCH3(0); break;
    case 1: T3_LAUNCH3(1); break;
    case 2: T3_LAUNCH3(2); break;
    case 3: T3_LAUNCH3(3); break;
    default: T3_LAUNCH3(4); break;
  }
#undef T3_LAUNCH3
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

}  // namespace vsys
