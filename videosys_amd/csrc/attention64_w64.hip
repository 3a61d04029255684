// flash_attn_d64_w64: joint [text | video] self-attention of CogVideoX (head_dim 64) with 64 query rows per wave and ONE wave per SIMD.
//
// Replaces (/root/reference/videosys): CogVideoXAttnProcessor2_0.__call__, models/transformers/cogvideox_transformer_3d.py:93-175 —
// LayerNorm qk-norm, rotary embedding on the video slice, F.scaled_dot_product_attention — the same contract and layouts (Kp / Vt of
// attn_prep_kv64) as flash_attn_d64_kernel (attention64.hip), for long key sequences without a bias table.
//
// The tile loop is the hand-allocated instruction stream of attention_w64.hip generated for this head size
// (csrc/gen/flash72_gen.py generate(d64=True) -> FLASH64_W64_ASM): 4 QK^T chunks instead of 5, K rows of 128 bytes with their
// chunks XOR-swizzled (four fragment address registers), 8 + 8 LDS-DMA pieces per tile (four per wave), and the THIRD 32-row block of
// the Vt image — in the d72 kernel the padding that carries the ones rows — kept constant in LDS (rows 72 / 76 = 1.0, the rest 0):
// the row sum of P rides on the matrix pipe, which has the slack here (40 MFMAs against ~190 VALU issues per tile), where
// flash_attn_d64_kernel adds it on the VALU.  At CogVideoX's 17 776 keys an item walks 278 tiles, so the one-item-per-workgroup
// form's seams (attention_w64.hip) do not matter.
#include "common.h"
#include "vsys_internal.h"

#include "flash72_w64_asm.inc"

namespace vsys {
namespace {

constexpr int HD = 64, KROW = 128, VROW = 128;
constexpr int K_TILE_BYTES = 64 * KROW;              // 8192
constexpr int KV_STAGE = K_TILE_BYTES + 96 * VROW;   // 20480: K tile + Vt image of 96 rows (64 fetched + 32 constant)
constexpr int W64_STAGES = 4;

struct Flash64W64Params {
  const bf16_t* q; int64_t q_stride;
  const bf16_t* ln_w; const bf16_t* ln_b;
  const float* rope_cos; const float* rope_sin; int rope_start, rope_len;
  const bf16_t* kp;
  const bf16_t* vt;
  bf16_t* out; int64_t out_stride;
  int heads, q_len, kv_len, kv_pad, nqb;
  float eps;
  float k_bound;   // VAR 5: upper bound on the norm of every Kp row (FlashW64Params::k_bound, attention_w64.hip)
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 make_rsrc64(const void* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  u32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}

// VAR: placement variant of the generated stream (flash72_gen.py body(): 1 = as the d72 kernel, 4 = one more P unit in the PV phase)
template <int VAR>
__global__ __launch_bounds__(256, 1) void flash_attn_d64_w64_kernel(Flash64W64Params p) {
#if __HIP_DEVICE_COMPILE__
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int tile_id = xcd_remap(blockIdx.x, gridDim.x);   // the q-blocks of one (batch, head) are consecutive on one XCD
  const int bh = tile_id / p.nqb, qb = tile_id - bh * p.nqb;
  const int b = bh / p.heads, h = bh - b * p.heads;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int q0 = qb * 256 + wave_u * 64;

  // ---- LDS-DMA assignment (as flash_attn_d64_kernel): 8 K pieces + 8 Vt pieces of 1 KiB per tile; wave w issues K pieces w, w + 4 and
  // Vt pieces w, w + 4 (rows 8 j + (lane >> 3), slot (lane & 7) holding logical slot (lane & 7) ^ ((row >> 1) & 7))
  const bf16_t* kbase = p.kp + (int64_t)bh * p.kv_pad * HD;
  const bf16_t* vbase = p.vt + (int64_t)bh * HD * p.kv_pad;
  const u32x4 rsrc_k = make_rsrc64(kbase, (unsigned)(p.kv_pad * HD * 2)), rsrc_v = make_rsrc64(vbase, (unsigned)(HD * p.kv_pad * 2));
  const int k_voff = lane * 16;
  const int v_voff = ((lane >> 3) * p.kv_pad * 2 + (((lane & 7) ^ (lane >> 4)) << 4)) ^ ((wave_u & 1) << 6);
  const int wl = wave_u * 1024;
  const int sv0 = wave_u * 8 * p.kv_pad * 2, sv1 = (wave_u + 4) * 8 * p.kv_pad * 2;
  const int lb = (int)(unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
  const int ntiles = (p.kv_len + 63) >> 6;
  const int lim = p.kv_len - (ntiles - 1) * 64 - 16 * hi;

  // rows 64..95 of every stage's Vt image: constant — row 72 (read by the hi = 0 half) and 76 (hi = 1) are ones, so accumulator 4 of
  // the third PV block is sum_k P[k][q]; the rest is MFMA padding
  for (int u = tid; u < W64_STAGES * 256; u += 256) {
    const int row = 64 + ((u & 255) >> 3);
    const unsigned w1 = (row == 72 || row == 76) ? 0x3f803f80u : 0u;
    *reinterpret_cast<uint4*>(smem + (u >> 8) * KV_STAGE + K_TILE_BYTES + row * VROW + (u & 7) * 16) = make_uint4(w1, w1, w1, w1);
  }

  // ---- fragment read offsets inside a stage
  const int krow = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
  const int k_roff = krow * KROW + ((hi ^ ((krow >> 1) & 7)) << 4);   // logical chunk 2 cc + hi at physical chunk ^ ((row >> 1) & 7)
  const int kfa0 = k_roff, kfa1 = k_roff ^ (1 << 5), kfa2 = k_roff ^ (2 << 5), kfa3 = k_roff ^ (3 << 5);
  const int v_roff = K_TILE_BYTES + l31 * VROW + (((2 * hi) ^ ((l31 >> 1) & 7)) << 4);
  const int vfa0 = v_roff ^ (0 << 4), vfa1 = v_roff ^ (1 << 4), vfa2 = v_roff ^ (4 << 4), vfa3 = v_roff ^ (5 << 4);

  // ---- Q fragments of the two 32-row blocks (B operand: lane holds Q[row][16c + 8hi .. +8], c = 0..3): LayerNorm + RoPE as
  // flash_attn_d64_kernel's prologue
  unsigned qw[2][16];
  float nmv[2] = {0.f, 0.f};   // VAR 5: -(row bound) of this lane's query row in block A / B
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    int qs = q0 + 32 * blk + l31;
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
    if constexpr (VAR == 5) {
      float qn2 = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) qn2 += x[c][e] * x[c][e];
      qn2 += __shfl_xor(qn2, 32, 64);
      nmv[blk] = -(sqrtf(qn2) * p.k_bound * 1.015625f);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint4 pk = pack8(x[c]);
      qw[blk][4 * c + 0] = pk.x; qw[blk][4 * c + 1] = pk.y; qw[blk][4 * c + 2] = pk.z; qw[blk][4 * c + 3] = pk.w;
    }
  }
  // Q -> a[96:111] (block A), a[116:131] (block B)
#define QW(b_, i_) "v"(qw[b_][i_])
  asm volatile(
      "v_accvgpr_write_b32 a96, %0\n\tv_accvgpr_write_b32 a97, %1\n\tv_accvgpr_write_b32 a98, %2\n\tv_accvgpr_write_b32 a99, %3\n\t"
      "v_accvgpr_write_b32 a100, %4\n\tv_accvgpr_write_b32 a101, %5\n\tv_accvgpr_write_b32 a102, %6\n\tv_accvgpr_write_b32 a103, %7\n\t"
      "v_accvgpr_write_b32 a104, %8\n\tv_accvgpr_write_b32 a105, %9\n\tv_accvgpr_write_b32 a106, %10\n\tv_accvgpr_write_b32 a107, %11\n\t"
      "v_accvgpr_write_b32 a108, %12\n\tv_accvgpr_write_b32 a109, %13\n\tv_accvgpr_write_b32 a110, %14\n\tv_accvgpr_write_b32 a111, %15\n\t"
      :
      : QW(0, 0), QW(0, 1), QW(0, 2), QW(0, 3), QW(0, 4), QW(0, 5), QW(0, 6), QW(0, 7), QW(0, 8), QW(0, 9), QW(0, 10), QW(0, 11), QW(0, 12), QW(0, 13), QW(0, 14), QW(0, 15)
      : "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111");
  asm volatile(
      "v_accvgpr_write_b32 a116, %0\n\tv_accvgpr_write_b32 a117, %1\n\tv_accvgpr_write_b32 a118, %2\n\tv_accvgpr_write_b32 a119, %3\n\t"
      "v_accvgpr_write_b32 a120, %4\n\tv_accvgpr_write_b32 a121, %5\n\tv_accvgpr_write_b32 a122, %6\n\tv_accvgpr_write_b32 a123, %7\n\t"
      "v_accvgpr_write_b32 a124, %8\n\tv_accvgpr_write_b32 a125, %9\n\tv_accvgpr_write_b32 a126, %10\n\tv_accvgpr_write_b32 a127, %11\n\t"
      "v_accvgpr_write_b32 a128, %12\n\tv_accvgpr_write_b32 a129, %13\n\tv_accvgpr_write_b32 a130, %14\n\tv_accvgpr_write_b32 a131, %15\n\t"
      "s_nop 1\n\t"
      :
      : QW(1, 0), QW(1, 1), QW(1, 2), QW(1, 3), QW(1, 4), QW(1, 5), QW(1, 6), QW(1, 7), QW(1, 8), QW(1, 9), QW(1, 10), QW(1, 11), QW(1, 12), QW(1, 13), QW(1, 14), QW(1, 15)
      : "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131");
#undef QW

  // ---- the tile loop
#define W64_LOOP(TEXT_)                                                                                                          \
  asm volatile(TEXT_                                                                                                             \
               :                                                                                                                 \
               : [rk] "s"(rsrc_k), [rv] "s"(rsrc_v), [wl] "s"(wl), [sv0] "s"(sv0), [sv1] "s"(sv1), [lb] "s"(lb), [nt] "s"(ntiles), \
                 [lim] "v"(lim), [kvo] "v"(k_voff), [vvo] "v"(v_voff), [kfa0] "v"(kfa0), [kfa1] "v"(kfa1), [kfa2] "v"(kfa2),       \
                 [kfa3] "v"(kfa3), [vfa0] "v"(vfa0), [vfa1] "v"(vfa1), [vfa2] "v"(vfa2), [vfa3] "v"(vfa3)                         \
               : FLASH64_W64_CLOBBERS)
  if constexpr (VAR == 5) {   // no running max: the row bounds ride in as the -m splat
    asm volatile(FLASH64_W64_ASM_V5
                 :
                 : [rk] "s"(rsrc_k), [rv] "s"(rsrc_v), [wl] "s"(wl), [sv0] "s"(sv0), [sv1] "s"(sv1), [lb] "s"(lb), [nt] "s"(ntiles),
                   [lim] "v"(lim), [kvo] "v"(k_voff), [vvo] "v"(v_voff), [kfa0] "v"(kfa0), [kfa1] "v"(kfa1), [kfa2] "v"(kfa2),
                   [kfa3] "v"(kfa3), [vfa0] "v"(vfa0), [vfa1] "v"(vfa1), [vfa2] "v"(vfa2), [vfa3] "v"(vfa3), [nma] "v"(nmv[0]),
                   [nmb] "v"(nmv[1])
                 : FLASH64_W64_CLOBBERS);
  }
  else if constexpr (VAR == 4) W64_LOOP(FLASH64_W64_ASM_V4);
  else W64_LOOP(FLASH64_W64_ASM_V1);
#undef W64_LOOP

  // ---- epilogue: O[q][d] = O^T[d][q] / l ; lane holds d = 32dt + (r&3) + 8(r>>2) + 4hi; 16-byte stores through v_permlane32_swap
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    float o[2][16], lsum;
#define RDO(dt_, r_, reg_) asm volatile("v_accvgpr_read_b32 %0, " reg_ : "=v"(o[dt_][r_]))
    if (blk == 0) {
      RDO(0, 0, "a0"); RDO(0, 1, "a1"); RDO(0, 2, "a2"); RDO(0, 3, "a3"); RDO(0, 4, "a4"); RDO(0, 5, "a5"); RDO(0, 6, "a6"); RDO(0, 7, "a7");
      RDO(0, 8, "a8"); RDO(0, 9, "a9"); RDO(0, 10, "a10"); RDO(0, 11, "a11"); RDO(0, 12, "a12"); RDO(0, 13, "a13"); RDO(0, 14, "a14"); RDO(0, 15, "a15");
      RDO(1, 0, "a16"); RDO(1, 1, "a17"); RDO(1, 2, "a18"); RDO(1, 3, "a19"); RDO(1, 4, "a20"); RDO(1, 5, "a21"); RDO(1, 6, "a22"); RDO(1, 7, "a23");
      RDO(1, 8, "a24"); RDO(1, 9, "a25"); RDO(1, 10, "a26"); RDO(1, 11, "a27"); RDO(1, 12, "a28"); RDO(1, 13, "a29"); RDO(1, 14, "a30"); RDO(1, 15, "a31");
      asm volatile("v_accvgpr_read_b32 %0, a36" : "=v"(lsum));
    } else {
      RDO(0, 0, "a48"); RDO(0, 1, "a49"); RDO(0, 2, "a50"); RDO(0, 3, "a51"); RDO(0, 4, "a52"); RDO(0, 5, "a53"); RDO(0, 6, "a54"); RDO(0, 7, "a55");
      RDO(0, 8, "a56"); RDO(0, 9, "a57"); RDO(0, 10, "a58"); RDO(0, 11, "a59"); RDO(0, 12, "a60"); RDO(0, 13, "a61"); RDO(0, 14, "a62"); RDO(0, 15, "a63");
      RDO(1, 0, "a64"); RDO(1, 1, "a65"); RDO(1, 2, "a66"); RDO(1, 3, "a67"); RDO(1, 4, "a68"); RDO(1, 5, "a69"); RDO(1, 6, "a70"); RDO(1, 7, "a71");
      RDO(1, 8, "a72"); RDO(1, 9, "a73"); RDO(1, 10, "a74"); RDO(1, 11, "a75"); RDO(1, 12, "a76"); RDO(1, 13, "a77"); RDO(1, 14, "a78"); RDO(1, 15, "a79");
      asm volatile("v_accvgpr_read_b32 %0, a84" : "=v"(lsum));
    }
#undef RDO
    const float inv = 1.0f / lsum;   // row 72 (hi = 0) / 76 (hi = 1) of the Vt image: sum_k P[k][q]
    const int qs = q0 + 32 * blk + l31;
    bf16_t* orow = p.out + ((int64_t)b * p.q_len + (qs < p.q_len ? qs : p.q_len - 1)) * p.out_stride + h * HD + 8 * hi;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      uint2 w[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        w[g].x = pack2bf(o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv);
        w[g].y = pack2bf(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const auto sx = __builtin_amdgcn_permlane32_swap(w[2 * k].x, w[2 * k + 1].x, false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(w[2 * k].y, w[2 * k + 1].y, false, false);
        if (qs < p.q_len) *reinterpret_cast<uint4*>(orow + dt * 32 + 16 * k) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
      }
    }
  }
#endif
}

}  // namespace

// true when the w64 kernel takes the problem (launch_flash_attn_d64 falls back to flash_attn_d64_kernel otherwise)
bool flash64_w64_supports(int q_len, int kv_len) { return kv_len >= 256 && q_len >= 256; }

int launch_flash_attn_d64_w64(const bf16_t* q, int64_t q_stride, const bf16_t* ln_w, const bf16_t* ln_b, const float* rope_cos,
                              const float* rope_sin, int rope_start, int rope_len, const bf16_t* kp, const bf16_t* vt, bf16_t* out,
                              int64_t out_stride, int batch, int heads, int q_len, int kv_len, int kv_pad, float eps, int var,
                              float k_bound, hipStream_t stream) {
  Flash64W64Params p;
  p.k_bound = k_bound;
  p.q = q; p.q_stride = q_stride; p.ln_w = ln_w; p.ln_b = ln_b; p.rope_cos = rope_cos; p.rope_sin = rope_sin;
  p.rope_start = rope_start; p.rope_len = rope_len; p.kp = kp; p.vt = vt; p.out = out; p.out_stride = out_stride;
  p.heads = heads; p.q_len = q_len; p.kv_len = kv_len; p.kv_pad = kv_pad; p.eps = eps;
  p.nqb = (q_len + 255) / 256;
  const int64_t nblk = (int64_t)p.nqb * batch * heads;
  if (nblk > 0x7fffffff || (int64_t)kv_pad * HD * 2 >= 0x7fffffff) return VSYS_ERR_SHAPE;
  const size_t lds = (size_t)W64_STAGES * KV_STAGE;   // 81920
  static std::atomic<unsigned long long> attr_seen{0};
  for (DeviceOnce once(attr_seen); once.todo(); once.done()) {
    (void)hipFuncSetAttribute((const void*)flash_attn_d64_w64_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)flash_attn_d64_w64_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)flash_attn_d64_w64_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  if (var == 5) {
    if (!(k_bound > 0.f)) return VSYS_ERR_ARG;
    hipLaunchKernelGGL(flash_attn_d64_w64_kernel<5>, dim3((unsigned)nblk), dim3(256), lds, stream, p);
  } else if (var == 4) hipLaunchKernelGGL(flash_attn_d64_w64_kernel<4>, dim3((unsigned)nblk), dim3(256), lds, stream, p);
  else hipLaunchKernelGGL(flash_attn_d64_w64_kernel<1>, dim3((unsigned)nblk), dim3(256), lds, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

}  // namespace vsys
