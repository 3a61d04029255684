// Attention kernels for head_dim 72 (STDiT3-XL/2: 1152 / 16) on gfx950.
//
// Reference call sites replaced (/root/reference/videosys):
//   flash_attn_d72 (spatial)  OpenSoraAttention.forward: q/k LlamaRMSNorm + SDPA          modules/attentions.py:75,100; normalization.py:28-33
//   flash_attn_d72 (cross)    OpenSoraMultiHeadCrossAttention.torch_impl (mask + SDPA)      modules/attentions.py:259-270
//   attn_prep_kv              the k half of the qk-norm + the layout the PV contraction needs
//   attn_temporal_d72         OpenSoraAttention.forward temporal branch: RMS qk-norm, RoPE (rotary_embedding_torch,
//                             third-party), native_attention with fp32 softmax, incl. the two "B (T S) C <-> (B S) T C"
//                             rearranges (open_sora_transformer_3d.py:203-206) which are never materialised
//
// flash_attn_d72: one workgroup = 4 waves = 128 query rows of one (batch, head); KV tiles of 64 keys staged
// HBM/L2 -> VGPR -> LDS, double-buffered, one barrier per tile.  Both contractions run on
// v_mfma_f32_32x32x16_bf16 in the "swapped" form so the softmax row of a query is lane-local:
//   S^T[kv][q] = K[kv][d] . Q[q][d]^T      (A = K rows, B = Q fragment, d padded 72 -> 80 with zero chunks)
//   O^T[d][q]  = Vt[d][kv] . P^T[kv][q]    (A = Vt rows (V pre-transposed by attn_prep_kv), B = P fragment)
// The K rows fed to MFMA row i are permuted (kv = 16*((i>>2)&1) + 4*(i>>3) + (i&3)) so that a lane's 16 accumulator
// registers of a 32-key tile are 16 CONSECUTIVE keys: P converts to the PV B-fragment with no cross-lane traffic and Vt
// is read with plain ds_read_b128.  K/V tiles are staged by LDS-DMA (buffer_load ... lds, no staging registers and no
// ds_write pass): the K image is the contiguous 9216-byte tile (144-byte rows, conflict-free for ds_read_b128 lane
// groups), the Vt image has 128-byte rows whose 16-byte slots are XOR-swizzled by (row>>1)&7 on the SOURCE side of the
// DMA and on the reads.  Online softmax in fp32; the scale 72^-1/2*log2e rides on K, the running max enters as the
// MFMA C operand, the row sum comes out of the PV contraction (ones rows in Vt).
#include "common.h"
#include "vsys_internal.h"

#include <cstdlib>
#include <type_traits>

namespace vsys {
namespace {

constexpr int HD = 72;          // head dim
constexpr int HD_ROWS = 96;     // Vt rows per head (3 MFMA tiles of 32; rows 72..95 are zero)
constexpr int KROW = 144;       // bytes per K row in LDS (72 bf16)
constexpr int VROW = 128;       // bytes per Vt row in LDS (64 keys * 2; 16-byte slots XOR-swizzled by (row>>1)&7)
constexpr int K_TILE_BYTES = 64 * KROW;        // 9216
constexpr int V_TILE_BYTES = HD_ROWS * VROW;   // 13824
constexpr int KV_STAGE = K_TILE_BYTES + V_TILE_BYTES;  // 23040
constexpr float NEG_BIG = -1.0e30f;

// ---------------------------------------------------------------------------------------------------------
// attn_prep_kv: k,v rows (strided, heads interleaved) -> Kp[batch][H][kv_pad][72] (RMS-normed, rows >= kv_len zero)
//                                                     -> Vt[batch][H][96][kv_pad] (transposed; cols >= kv_len zero)
// grid: (kv_pad/64, batch*H); block 256.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_prep_kv_kernel(const bf16_t* __restrict__ k, int64_t k_stride,
                                                           const bf16_t* __restrict__ v, int64_t v_stride,
                                                           const bf16_t* __restrict__ k_norm_w, bf16_t* __restrict__ kp,
                                                           bf16_t* __restrict__ vt, int heads, int kv_len, int kv_pad,
                                                           float eps, float kscale) {
  __shared__ __attribute__((aligned(16))) bf16_t vs[64][HD + 8];  // [token][d], 160-byte rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s0 = blockIdx.x * 64;
  const int bh = blockIdx.y;
  const int b = bh / heads, h = bh - b * heads;

  // ---- V tile -> LDS (64 tokens x 9 chunks)
  for (int q = tid; q < 64 * 9; q += 256) {
    const int r = q / 9, c = q - r * 9;
    const int s = s0 + r;
    uint4 val = make_uint4(0, 0, 0, 0);
    if (s < kv_len) val = *reinterpret_cast<const uint4*>(v + ((int64_t)b * kv_len + s) * v_stride + h * HD + c * 8);
    *reinterpret_cast<uint4*>(&vs[r][c * 8]) = val;
  }

  // ---- K: 3 lanes per token row (24 dims each), 21 rows per wave
  {
    const int g = lane / 3, part = lane - g * 3;
    const int r = wave * 21 + g;
    const bool active = (lane < 63) && (r < 64);
    const int s = s0 + r;
    float x[24];
#pragma unroll
    for (int e = 0; e < 24; ++e) x[e] = 0.f;
    if (active && s < kv_len) {
      const bf16_t* src = k + ((int64_t)b * kv_len + s) * k_stride + h * HD + part * 24;
#pragma unroll
      for (int c = 0; c < 3; ++c) unpack8(*reinterpret_cast<const uint4*>(src + c * 8), x + c * 8);
    }
    if (k_norm_w != nullptr) {
      float ss = 0.f;
#pragma unroll
      for (int e = 0; e < 24; ++e) ss += x[e] * x[e];
      const int base = g * 3;
      const float tot = __shfl(ss, base, 64) + __shfl(ss, base + 1, 64) + __shfl(ss, base + 2, 64);
      const float rstd = rsqrtf(tot / (float)HD + eps);
#pragma unroll
      for (int e = 0; e < 24; ++e) {
        const float nrm = bf2f(f2bf(x[e] * rstd));  // hidden_states.to(input_dtype) before the weight multiply
        x[e] = nrm * bf2f(k_norm_w[part * 24 + e]);
      }
    }
    // softmax scale 72^-1/2 and log2(e) ride on K (one rounding, the one the reference's k also gets), so that the
    // QK^T accumulators are exp2-ready and the flash kernel spends no VALU on scaling
#pragma unroll
    for (int e = 0; e < 24; ++e) x[e] *= kscale;
    if (active) {
      bf16_t* dst = kp + (((int64_t)bh * kv_pad) + s) * HD + part * 24;
#pragma unroll
      for (int c = 0; c < 3; ++c) *reinterpret_cast<uint4*>(dst + c * 8) = pack8(x + c * 8);
    }
  }
  __syncthreads();

  // ---- Vt rows: thread -> (d, 8-token chunk)
  for (int q = tid; q < HD * 8; q += 256) {
    const int d = q >> 3, c = q & 7;
    uint4 o;
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = (uint32_t)vs[c * 8 + 2 * e][d] | ((uint32_t)vs[c * 8 + 2 * e + 1][d] << 16);
    o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
    *reinterpret_cast<uint4*>(vt + ((int64_t)bh * HD_ROWS + d) * kv_pad + s0 + c * 8) = o;
  }
  // rows 72 and 76 of Vt are all-ones over the valid keys: O^T rows 72/76 then accumulate sum_k P[k][q] on the matrix
  // pipe (the rows are MFMA padding anyway), i.e. the softmax denominator costs no VALU adds.  Rows 73-75, 77-95 stay 0.
  if (tid < 16) {
    const int d = HD + 4 * (tid >> 3), c = tid & 7;
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k0 = s0 + c * 8 + 2 * e;
      w[e] = (k0 < kv_len ? 0x3f80u : 0u) | (k0 + 1 < kv_len ? 0x3f800000u : 0u);
    }
    uint4 o;
    o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
    *reinterpret_cast<uint4*>(vt + ((int64_t)bh * HD_ROWS + d) * kv_pad + s0 + c * 8) = o;
  }
}

// ---------------------------------------------------------------------------------------------------------
// flash_attn_d72.  grid: (ceil(q_len/128), batch*heads); block 256 (4 waves x 32 query rows).
// ---------------------------------------------------------------------------------------------------------
struct FlashParams {
  const bf16_t* q; int64_t q_stride;      // q(b, s, h) at q + (b*q_len + s)*q_stride + h*72
  const bf16_t* q_norm_w;                  // [72] or null
  const bf16_t* kp;                        // [batch][H][kv_pad][72]
  const bf16_t* vt;                        // [batch][H][96][kv_pad]
  bf16_t* out; int64_t out_stride;         // out(b, s, h) at out + (b*q_len + s)*out_stride + h*72
  int heads, q_len, kv_len, kv_pad, nqb;
  int chunks;                              // RES kernel: workgroups per (batch, head), each walks nqb / chunks query blocks
  float eps;
  unsigned long long* dbg;                 // lab variant 2 only: 5 phase-cycle accumulators
};

// WPS = waves per SIMD the register allocation targets: 2 (two workgroups per CU, no spills) or 3 (168 registers: the peeled
// MASKED last tile spills — cross shape 0.156 vs 0.099 ms — but unmasked long key sequences gain: spatial 0.241 vs 0.251 ms;
// the launcher picks 3 for kv_len >= 512 that is a multiple of 64).
// ABL (lab builds): 1 = K/V tiles after the first are not fetched (compute, LDS and barriers only); 2 = s_memtime stamps
// at the phase boundaries of every tile, summed per wave into p.dbg[0..4] (QK issue | max chain | exp + PV | vmcnt | barrier).
// (An 8-wave form of this kernel — 256 rows share a K/V tile and its 19 LDS-DMA pieces, one workgroup per CU — measured 0.256 vs
// 0.247 ms at the config-2 spatial shape: the per-tile barrier then couples both waves of every SIMD.  Not kept.)
// RES ("resident K/V", few keys — the cross attention against <= 320 text tokens): one workgroup = 8 waves = 256 query rows per
// step stages ALL KV tiles of its (batch, head) into LDS once (<= 5 stages = 112.5 KiB) and then walks nqb / chunks query blocks
// with no LDS-DMA, no vmcnt wait and no barrier in the loop: a wave's 5 pieces per tile cost it ~180 cycles each at issue
// (DESIGN.md §3.2), which for 5-tile problems is most of the per-tile overhead.  Same tile() code, same arithmetic, same bits.
constexpr int RES_MAX_TILES = 5;
constexpr int RES_Q_BYTES = 5 * 1024;   // per-wave Q image (32 rows x 144 B = 4.5 KiB, staged in 5 LDS-DMA pieces)
// EXACT (RES only; vsys_flash_attn_d72_exact): the caller promises that the Kp rows and Vt columns behind kv_len are the zeros
// attn_prep_kv wrote for exactly this kv_len.  A padding key then has logit 0 and weight 0 in numerator AND denominator (its Vt
// column is zero in the ones rows too), so the ragged last tile needs no mask: the mask code (64 compare / select pairs per query
// block of five tiles) is compiled out.  The only thing a padding key still touches is the running max (it sees the logit 0); that is
// harmless unless EVERY real logit of a row lies ~100 (exp2 domain) below zero — the block's denominator then underflows, which is
// detected per block and answered by recomputing that block with the masked tile sequence (cold path, same bits as the masked kernel).
template <int ABL, int WPS, bool RES = false, bool EXACT = false>
__global__ __launch_bounds__(RES ? 512 : 256, RES ? 2 : WPS) void flash_attn_d72_kernel(FlashParams p) {
#if __HIP_DEVICE_COMPILE__  // buffer-resource types exist in the device pass only
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  // 1-D grid with the XCD remap: the q-blocks of one (batch, head) share K/V, so they must be CONSECUTIVE on one XCD to
  // hit its L2 (a 2-D grid with gridDim.x == 8 puts each of them on a different XCD: 5.7x over-fetch measured).
  const int tile_id = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = RES ? tile_id / p.chunks : tile_id / p.nqb;
  const int qb = RES ? tile_id - bh * p.chunks : tile_id - bh * p.nqb;   // RES: the chunk of query blocks this workgroup walks
  const int b = bh / p.heads, h = bh - b * p.heads;
  constexpr int NW = RES ? 8 : 4;
  int q0 = qb * 128 + wave * 32;   // (RES: set per query block below)

  // ---- K/V staging by LDS-DMA: 9 K pieces (1 KiB each, the tile is contiguous) + 10 Vt pieces (8 rows x 128 B each,
  // rows 0..79; rows 80..95 of the LDS image are zeroed once and never overwritten); wave w issues pieces w, w+4, ...
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const bf16_t* kbase = p.kp + (int64_t)bh * p.kv_pad * HD;         // tile t: + t*64*72 (contiguous 9216 bytes)
  const bf16_t* vbase = p.vt + (int64_t)bh * HD_ROWS * p.kv_pad;    // row d: + d*kv_pad, tile t: + t*64
  const auto rsrc_k = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, p.kv_pad * HD * 2, 0x00020000);
  const auto rsrc_v = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, HD_ROWS * p.kv_pad * 2, 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int k_voff = lane * 16;
  // Vt piece j: lane -> row 8j + (lane>>3), physical slot lane&7 holds logical slot (lane&7) ^ ((row>>1)&7);
  // (row>>1)&7 = ((j&1)<<2) | (lane>>4), so odd pieces differ from even ones by XOR 64 in the byte offset
  const int v_voff = (lane >> 3) * p.kv_pad * 2 + (((lane & 7) ^ (lane >> 4)) << 4);
  // Streaming kernel (4 waves, 5 slots each): slots 0, 1 = K pieces w, w + 4; slots 2, 3 = Vt pieces w, w + 4; slot 4 = K piece 8
  // (wave 0), Vt piece 8 (wave 1), Vt piece 9 (waves 2 and 3: the same bytes twice).  The kind of a slot is then a COMPILE-TIME
  // property except for slot 4, whose descriptor / offsets are selected once up here: with "pieces w, w + 4, ..." the kind of
  // every slot depends on the wave and the tile loop carries 15 scalar branches per tile around its five loads.
  const bool s4k = wave_u == 0;
  const int s4j = wave_u == 1 ? 8 : 9;
  const auto rsrc_4 = __builtin_amdgcn_make_buffer_rsrc((void*)(s4k ? kbase : vbase), 0, s4k ? p.kv_pad * HD * 2 : HD_ROWS * p.kv_pad * 2,
                                                        0x00020000);
  const int voff_4 = s4k ? k_voff : (v_voff ^ ((s4j & 1) << 6));
  const int soff_4 = s4k ? 8 * 1024 : s4j * 8 * p.kv_pad * 2, step_4 = s4k ? K_TILE_BYTES : 128;
  const int lds_4 = s4k ? 8 * 1024 : K_TILE_BYTES + s4j * 1024;
  auto stage = [&](int t, int buf) {
    char* base = smem + buf * KV_STAGE;
    if constexpr (RES) {
#pragma unroll
      for (int idx = 0; idx < 3; ++idx) {
        const int piece = wave_u + NW * idx;
        if (piece < 9) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_k, (lds_ptr_t)(base + piece * 1024), 16, k_voff, t * K_TILE_BYTES + piece * 1024, 0, 0);
        } else if (piece < 19) {
          const int j = piece - 9;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_v, (lds_ptr_t)(base + K_TILE_BYTES + j * 1024), 16, v_voff ^ ((j & 1) << 6),
                                                   j * 8 * p.kv_pad * 2 + t * 128, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int piece = wave_u + 4 * i;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_k, (lds_ptr_t)(base + piece * 1024), 16, k_voff, t * K_TILE_BYTES + piece * 1024, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int j = wave_u + 4 * i;   // (j & 1) == (wave & 1)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_v, (lds_ptr_t)(base + K_TILE_BYTES + j * 1024), 16, v_voff ^ ((wave_u & 1) << 6),
                                                 j * 8 * p.kv_pad * 2 + t * 128, 0, 0);
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_4, (lds_ptr_t)(base + lds_4), 16, voff_4, soff_4 + t * step_4, 0, 0);
    }
  };
  // the first K/V tile is requested BEFORE the Q rows: both round trips are in flight together (with 300 text keys a workgroup
  // lives for five tiles only, so a serialized prologue is a tenth of it)
  const int ntiles = (p.kv_len + 63) / 64;  // tiles made only of keys >= kv_len are never touched (kv_pad is the stride)
  if constexpr (RES) {
    for (int t = 0; t < ntiles; ++t) stage(t, t);   // every tile of this (batch, head), once
    for (int q = tid; q < ntiles * 128; q += 512)
      *reinterpret_cast<uint4*>(smem + (q >> 7) * KV_STAGE + K_TILE_BYTES + 80 * VROW + (q & 127) * 16) = make_uint4(0, 0, 0, 0);
  } else {
    stage(0, 0);
    // rows 80..95 of both Vt images (MFMA padding the DMA never writes): 2 x 2 KiB of zeros
    char* z = smem + (tid >> 7) * KV_STAGE + K_TILE_BYTES + 80 * VROW + (tid & 127) * 16;
    *reinterpret_cast<uint4*>(z) = make_uint4(0, 0, 0, 0);
  }

  // ---- Q fragment (B operand): lane holds Q[q0 + l31][16c + 8hi .. +8], c = 0..4 (d >= 72 -> 0)
  bf16x8 qf[5];
  // split in two so that the resident-K/V loop can request the rows of query block i + 1 before the tiles of block i
  uint4 qraw[5];
  auto fetch_q = [&](int q0_) {
    int qs = q0_ + l31;
    qs = qs < p.q_len ? qs : p.q_len - 1;
    const bf16_t* qrow = p.q + ((int64_t)b * p.q_len + qs) * p.q_stride + h * HD;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const int d0 = 16 * c + 8 * hi;
      qraw[c] = d0 < HD ? *reinterpret_cast<const uint4*>(qrow + d0) : make_uint4(0, 0, 0, 0);
    }
  };
  auto finish_q = [&]() {
    if (p.q_norm_w == nullptr) {   // (cross attention, Latte: no q norm) the raw bf16 pieces ARE the fragment: no unpack / repack pass
#pragma unroll
      for (int c = 0; c < 5; ++c) qf[c] = __builtin_bit_cast(bf16x8, qraw[c]);
      return;
    }
    float x[5][8];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      unpack8(qraw[c], x[c]);
#pragma unroll
      for (int e = 0; e < 8; ++e) ss += x[c][e] * x[c][e];
    }
    {
      ss += __shfl_xor(ss, 32, 64);
      const float rstd = rsqrtf(ss / (float)HD + p.eps);
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        const int d0 = 16 * c + 8 * hi;
        if (d0 < HD) {
          float w[8];
          unpack8(*reinterpret_cast<const uint4*>(p.q_norm_w + d0), w);
#pragma unroll
          for (int e = 0; e < 8; ++e) x[c][e] = bf2f(f2bf(x[c][e] * rstd)) * w[e];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[c][e] = (__bf16)x[c][e];
  };
  if constexpr (!RES) { fetch_q(q0); finish_q(); }

  // Vt fragment read offsets: row dt*32 + l31, logical slot kt*4 + 2hi + cc  ->  + dt*4096, ^ ((kt*4 + cc) << 4)
  const int v_roff = K_TILE_BYTES + l31 * VROW + (((2 * hi) ^ ((l31 >> 1) & 7)) << 4);

  // permuted K row for MFMA row i = l31: lane's 16 acc regs <-> 16 consecutive keys (16*hi + reg)
  const int krow = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);

  f32x16 o[3];
  f32x16 minit;
  auto reset_acc = [&]() {
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) minit[r] = 0.f;
  };
  reset_acc();
  // Running max m (exp2 domain: K carries scale*log2e) is kept NEGATED and splatted over 16 registers: it is the C input
  // of the first QK^T MFMA of every 32-key tile, so the accumulators come out as s - m, ready for v_exp, with no
  // per-tile zero-init and no per-element subtract.  The row sum l lives in o[2][4] (Vt rows 72/76 are ones).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // Deferred running max: the accumulators are rescaled only when some row's tile max exceeds its running max by more
  // than 8 (exp2 domain), so P <= 2^8 (bf16 keeps fp32's exponent range, sums are fp32) and the common tile skips the
  // 48-register O rescale.  Mathematically the same softmax.
  const float defer_thr = 8.0f;

  // Where the next tile's five LDS-DMA pieces of a wave are issued: in front of the tile (WPS 2), or one behind each of the tile's first five
  // QK MFMAs (WPS 3, the spatial launch: 0.2328 -> 0.2280 / 0.2330 -> 0.2298 ms at config 2, same bits; ABL 4 = in front, the A/B partner)
  constexpr bool DMA_BESIDE_QK = !RES && WPS == 3 && ABL != 4 && ABL != 1 && ABL != 5;
  unsigned long long tacc[5] = {0, 0, 0, 0, 0}, tprev = 0;
#define FLASH_STAMP(i_)                                              \
  do {                                                               \
    if (ABL == 2) {                                                  \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
      tacc[i_] += now_ - tprev;                                      \
      tprev = now_;                                                  \
    }                                                                \
  } while (0)
  auto tile = [&](int t, int cur, const bool masked, const bool stage_next = false) {
    const char* sk = smem + cur * KV_STAGE;

    // ---- S^T - m = K Q^T - m : two 32-key tiles.  All 20 K fragment reads are issued before the first MFMA (hipcc
    // otherwise alternates read / wait / MFMA and exposes one LDS round trip per MFMA: with two waves per SIMD nothing
    // covers it).  The fifth chunk (d 64..79) is read unconditionally: for hi = 1 it is the first 16 bytes of the next
    // row, multiplied by Q's zero chunk.
    bf16x8 kf0[5], kf1[5];
    {
      const char* krp = sk + krow * KROW + 16 * hi;
#pragma unroll
      for (int cc = 0; cc < 5; ++cc) kf0[cc] = *reinterpret_cast<const bf16x8*>(krp + 32 * cc);
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x16 s[2];
    {
      const char* krp = sk + (32 + krow) * KROW + 16 * hi;
#pragma unroll
      for (int cc = 0; cc < 5; ++cc) kf1[cc] = *reinterpret_cast<const bf16x8*>(krp + 32 * cc);
      // D != C on purpose (the builtin ties them and hipcc would first copy the 16 minit registers into s)
      asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(s[0]) : "v"(kf0[0]), "v"(qf[0]), "v"(minit));
      if (DMA_BESIDE_QK && stage_next) {   // one LDS-DMA piece behind each of the first MFMAs: the piece's issue time runs beside the matrix pipe
        stage(t + 1, cur ^ 1);
#pragma unroll
        for (int cc = 1; cc < 5; ++cc) s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0[cc], qf[cc], s[0], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x010, 1, 0); }
      } else {
#pragma unroll
      for (int cc = 1; cc < 5; ++cc) s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0[cc], qf[cc], s[0], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(s[1]) : "v"(kf1[0]), "v"(qf[0]), "v"(minit));
#pragma unroll
    for (int cc = 1; cc < 5; ++cc) s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1[cc], qf[cc], s[1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    FLASH_STAMP(0);
    // V fragments do not depend on the softmax: the reads for keys 0..31 are issued now so their LDS latency hides under
    // the max / exp work below; those for keys 32..63 go out before the first PV MFMAs (register budget: 256).
    bf16x8 vf0[2][3], vf1[2][3];
#define FLASH_VREAD(dst_, kt_)                                                                                   \
  _Pragma("unroll") for (int cc = 0; cc < 2; ++cc) _Pragma("unroll") for (int dt = 0; dt < 3; ++dt)              \
    dst_[cc][dt] = *reinterpret_cast<const bf16x8*>(sk + dt * 32 * VROW + (v_roff ^ (((kt_) * 4 + cc) << 4)))
    FLASH_VREAD(vf0, 0);
    __builtin_amdgcn_sched_barrier(0);

    // ---- online softmax (lane: query l31; keys 64t + 32kt + 16hi + r).  Only the last tile of a ragged kv_len masks.
    if (masked) {
      const int lim = p.kv_len - (t * 64 + 16 * hi);  // keys with 32kt + r >= lim are padding
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt * 32 + r >= lim) s[kt][r] = NEG_BIG;
    }
    float mx = s[0][0];  // tile max relative to the running max
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
    {  // the other 16 keys of each 32-key tile live in lane l31 + 32: exchange on the VALU (no LDS round trip)
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    if (t == 0 || __builtin_amdgcn_ballot_w64(mx > defer_thr) != 0) {  // wave-uniform
      asm volatile("; rescale path (rare): kept out of line" ::: "memory");  // not if-convertible
      const float delta = t == 0 ? mx : fmaxf(mx, 0.f);  // first tile: adopt its max (signed); later: only raise
#pragma unroll
      for (int r = 0; r < 16; ++r) minit[r] -= delta;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] -= delta;
      if (t != 0) {   // (the first tile adopts its max into the accumulator INIT: O is still zero, nothing to scale — and exp2(-max) of a
        const float alpha = __builtin_amdgcn_exp2f(-delta);   // very negative first max would be inf x 0)
#pragma unroll
        for (int dt = 0; dt < 3; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      }
    }
    FLASH_STAMP(1);
    // ---- P = exp2(s - m), O^T += Vt P^T: the exps of keys 32..63 run under the MFMAs of keys 0..31
    bf16x8 pf0[2], pf1[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) pf0[r >> 3][r & 7] = (__bf16)__builtin_amdgcn_exp2f(s[0][r]);
    FLASH_VREAD(vf1, 1);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf0[cc][dt], pf0[cc], o[dt], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) pf1[r >> 3][r & 7] = (__bf16)__builtin_amdgcn_exp2f(s[1][r]);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf1[cc][dt], pf1[cc], o[dt], 0, 0, 0);
#undef FLASH_VREAD
    FLASH_STAMP(2);
  };

  // ---- epilogue: O[q][d] = O^T[d][q] / l ; lane holds d = 32dt + (r&3) + 8(r>>2) + 4hi
  // Stores widened to 16 bytes (cdna_hip_programming.md T21): group pair (g, g+1) of a 32-dim block is exchanged between the
  // half-waves with v_permlane32_swap, after which lanes 0-31 hold dims 32dt + 16k .. +7 and lanes 32-63 dims +8 .. +15 of their
  // row: 5 dwordx4 stores per lane instead of 9 dwordx2 (the tail is store-issue bound, not bandwidth bound).
  auto store_o = [&]() {
    const float inv = 1.0f / o[2][4];  // d = 72 (hi = 0) / 76 (hi = 1): the ones rows of Vt, i.e. sum_k P[k][q]
    const int qs = q0 + l31;
    bf16_t* orow = p.out + ((int64_t)b * p.q_len + (qs < p.q_len ? qs : p.q_len - 1)) * p.out_stride + h * HD + 8 * hi;
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
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
        const int d0 = dt * 32 + 16 * k;   // + 8 hi (in orow)
        if (qs < p.q_len && d0 + 8 * hi + 8 <= HD) *reinterpret_cast<uint4*>(orow + d0) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
      }
    }
  };

  if constexpr (RES) {
    // every KV tile is resident: walk this workgroup's query blocks (256 rows each) with nothing but tile() in the loop
    const int nqb = (p.q_len + 255) >> 8;
    const int qb0 = (int)((int64_t)qb * nqb / p.chunks), qb1 = (int)((int64_t)(qb + 1) * nqb / p.chunks);
    // (The mask of a ragged last tile stays: kv_len may be SHORTER than what the K / Vt buffers were prepared for — Latte's per-sample
    // text lengths inside one buffer, test_flash_attn_short_key_length_inside_a_longer_buffer — so the keys behind it are not zero.)
    const bool ragged = !EXACT && (p.kv_len & 63) != 0;
    // The Q rows of query block i + 1 travel HBM -> LDS (wave-private 5 KiB image: 32 rows x 144 B, contiguous 16-byte units) by
    // LDS-DMA under the tiles of block i: a register prefetch (20 VGPRs) makes hipcc spill inside tile(), and without a prefetch
    // the global round trip (~2 us) is exposed once per five tiles.
    char* qlds = smem + RES_MAX_TILES * KV_STAGE + wave_u * RES_Q_BYTES;
    const auto rsrc_q = __builtin_amdgcn_make_buffer_rsrc((void*)(p.q + (int64_t)b * p.q_len * p.q_stride + h * HD), 0, 0x7fffffff, 0x00020000);
    auto dma_q = [&](int q0_) {
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));   // opaque: the ten row / chunk terms below are recomputed per block, not kept in VGPRs across tile()
#pragma unroll
      for (int pc = 0; pc < 5; ++pc) {
        const int u = pc * 64 + lane_o;          // 16-byte unit: row u / 9, chunk u % 9 (units >= 288 re-read row 31: never used)
        int row = (u * 7282) >> 16;              // u / 9 for u < 320
        const int ch = u - 9 * row;
        row = row < 32 ? row : 31;
        int r = q0_ + row;
        r = r < p.q_len ? r : p.q_len - 1;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_q, (lds_ptr_t)(qlds + pc * 1024), 16, (int)(r * (int)p.q_stride * 2 + ch * 16), 0, 0, 0);
      }
    };
    dma_q(qb0 * 256 + wave * 32);
    for (int blk = qb0; blk < qb1; ++blk) {
      q0 = blk * 256 + wave * 32;
      if (blk != qb0) reset_acc();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's Q image has landed (nobody else touches it)
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        const int d0 = 16 * c + 8 * hi;
        qraw[c] = d0 < HD ? *reinterpret_cast<const uint4*>(qlds + l31 * KROW + (2 * c + hi) * 16) : make_uint4(0, 0, 0, 0);
      }
      finish_q();
      __builtin_amdgcn_sched_barrier(0);
      // the next block's Q rows: requested behind the first tile's QK MFMAs of the pipelined sequence (a piece costs its wave ~100+ cycles
      // at issue; there they run beside the matrix pipe), in front of the tiles otherwise
      auto next_q = [&]() {
        if (blk + 1 < qb1) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the image reads above are done before the image is overwritten
          dma_q(q0 + 256);
        }
      };
      if (ABL == 3) next_q();
      __builtin_amdgcn_sched_barrier(0);
      if (ABL == 3) {   // lab: the unpipelined tile() sequence (A/B of the software pipeline below; same bits)
        for (int t = 0; t < ntiles - 1; ++t) tile(t, t, false);
        if (ragged) tile(ntiles - 1, ntiles - 1, true);
        else tile(ntiles - 1, ntiles - 1, false);
      } else {
        // Software pipeline over the KV tiles (nothing in this loop waits for memory or for another wave, so the only thing
        // between the matrix pipe and its work is the order of this wave's own instruction stream):
        //   (a)  S(t+1) = K(t+1) Q^T - m   [10 MFMAs]   beside   P(t) = exp2(S(t)), keys 0..31     [16 exp]
        //   (b)  O += Vt(t) P(t)^T         [12 MFMAs]   beside   exp2 of keys 32..63, then mask / max of S(t+1)
        //   rare: rescale O, m, S(t+1) when the max of tile t+1 overshoots (same rule and arithmetic as tile())
        // Same operations on the same values in the same per-accumulator order as tile(): bit-identical results.
        f32x16 sa[2], sb[2];
        bf16x8 pa0[2], pa1[2];
        auto qk = [&](int t, f32x16 (&d)[2]) {
          const char* sk = smem + t * KV_STAGE;
          bf16x8 kf0[5], kf1[5];
#pragma unroll
          for (int cc = 0; cc < 5; ++cc) {
            kf0[cc] = *reinterpret_cast<const bf16x8*>(sk + krow * KROW + 16 * hi + 32 * cc);
            kf1[cc] = *reinterpret_cast<const bf16x8*>(sk + (32 + krow) * KROW + 16 * hi + 32 * cc);
          }
          asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d[0]) : "v"(kf0[0]), "v"(qf[0]), "v"(minit));
          asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d[1]) : "v"(kf1[0]), "v"(qf[0]), "v"(minit));
#pragma unroll
          for (int cc = 1; cc < 5; ++cc) {
            d[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0[cc], qf[cc], d[0], 0, 0, 0);
            d[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1[cc], qf[cc], d[1], 0, 0, 0);
          }
        };
        // mask (last ragged tile) and tile max: VALU only, scheduled beside the PV MFMAs of the previous tile
        auto tile_max = [&](int t, f32x16 (&d)[2], const bool masked) -> float {
          if (masked) {   // (wave-uniform, last tile of a ragged kv_len only)
            const int lim = p.kv_len - (t * 64 + 16 * hi);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
              for (int r = 0; r < 16; ++r)
                if (kt * 32 + r >= lim) d[kt][r] = NEG_BIG;
          }
          float mx = d[0][0];
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, d[kt][r]);
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
          return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        };
        // deferred rescale of everything that is relative to the running max (rare after the first tile)
        auto rescale = [&](int t, f32x16 (&d)[2], float mx) {
          if (t == 0 || __builtin_amdgcn_ballot_w64(mx > defer_thr) != 0) {
            asm volatile("; rescale path (rare): kept out of line" ::: "memory");
            const float delta = t == 0 ? mx : fmaxf(mx, 0.f);
#pragma unroll
            for (int r = 0; r < 16; ++r) minit[r] -= delta;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
              for (int r = 0; r < 16; ++r) d[kt][r] -= delta;
            if (t != 0) {   // (first tile: O is still zero — the adoption is folded into the accumulator init)
              const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
              for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            }
          }
        };
#define RES_SGB(mask_, n_) __builtin_amdgcn_sched_group_barrier(mask_, n_, 0)
        // one pipeline step on S(t) in sc (already max-processed); S(t+1) is produced into sn
        // (has_next is a compile-time tag: as a run-time condition it splits (a) and (b) into several basic blocks and nothing can
        // be scheduled across them)
        auto step = [&](auto has_next_tag, int t, f32x16 (&sc)[2], f32x16 (&sn)[2], const bool next_masked) {
          const char* sk = smem + t * KV_STAGE;
          constexpr bool has_next = decltype(has_next_tag)::value;
          // (a)
          if constexpr (has_next) qk(t + 1, sn);
#pragma unroll
          for (int r = 0; r < 16; ++r) pa0[r >> 3][r & 7] = (__bf16)__builtin_amdgcn_exp2f(sc[0][r]);
          bf16x8 vf0[2][3], vf1[2][3];
#pragma unroll
          for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
              vf0[cc][dt] = *reinterpret_cast<const bf16x8*>(sk + dt * 32 * VROW + (v_roff ^ ((0 * 4 + cc) << 4)));
              vf1[cc][dt] = *reinterpret_cast<const bf16x8*>(sk + dt * 32 * VROW + (v_roff ^ ((1 * 4 + cc) << 4)));
            }
          // emitted order of (a): the 20 K fragment reads, then every QK^T MFMA followed by two exps, a convert and a Vt read —
          // left alone hipcc issues the ten MFMAs back to back and the exps behind them, i.e. nothing overlaps (in-order issue)
          if constexpr (has_next) {
            RES_SGB(0x100, 20);
#pragma unroll
            for (int i = 0; i < 10; ++i) { RES_SGB(0x008, 1); RES_SGB(0x400, 2); RES_SGB(0x002, 1); RES_SGB(0x100, 1); }
          }
          __builtin_amdgcn_sched_barrier(0);
          // (b)
#pragma unroll
          for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf0[cc][dt], pa0[cc], o[dt], 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 16; ++r) pa1[r >> 3][r & 7] = (__bf16)__builtin_amdgcn_exp2f(sc[1][r]);
#pragma unroll
          for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf1[cc][dt], pa1[cc], o[dt], 0, 0, 0);
          float mxn = 0.f;
          if constexpr (has_next) mxn = tile_max(t + 1, sn, next_masked);
          // first six PV MFMAs: the 16 exps + 8 converts of keys 32..63 beside them; last six: the max chain of S(t+1)
#pragma unroll
          for (int i = 0; i < 6; ++i) { RES_SGB(0x008, 1); RES_SGB(0x400, 3); RES_SGB(0x002, 2); }
#pragma unroll
          for (int i = 0; i < 6; ++i) { RES_SGB(0x008, 1); RES_SGB(0x002, 5); }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (has_next) rescale(t + 1, sn, mxn);
        };
        qk(0, sa);
        __builtin_amdgcn_sched_barrier(0);
        next_q();
        __builtin_amdgcn_sched_barrier(0);
        rescale(0, sa, (ragged && ntiles == 1) ? tile_max(0, sa, true) : tile_max(0, sa, false));
        int t = 0;
        for (; t + 2 < ntiles; t += 2) {   // two steps per trip: the S buffers swap roles without register copies
          step(std::true_type{}, t, sa, sb, false);
          step(std::true_type{}, t + 1, sb, sa, ragged && t + 2 == ntiles - 1);
        }
        if (t + 1 < ntiles) {   // two tiles left
          step(std::true_type{}, t, sa, sb, ragged);
          step(std::false_type{}, t + 1, sb, sa, false);
        } else {
          step(std::false_type{}, t, sa, sb, false);
        }
#undef RES_SGB
      }
      if constexpr (EXACT) {
        // (see the template comment) a row whose real logits all sit ~100 below the padding keys' 0: recompute the block masked
        if (__builtin_amdgcn_ballot_w64(!(o[2][4] >= 0x1p-100f)) != 0) {
          asm volatile("; exact-keys guard (cold): masked recompute of this query block" ::: "memory");
          reset_acc();
          for (int t = 0; t < ntiles; ++t) tile(t, t, t == ntiles - 1 && (p.kv_len & 63) != 0);
        }
      }
      store_o();
    }
  } else {
    if (ABL == 2) tprev = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < ntiles - 1; ++t) {  // last tile peeled: no conditional staging inside the loop
      const int cur = t & 1;
      if (ABL != 1 && ABL != 5 && !DMA_BESIDE_QK) stage(t + 1, cur ^ 1);  // every wave finished reading buffer cur^1 before the barrier of tile t-1
      __builtin_amdgcn_sched_barrier(0);
      tile(t, cur, false, true);
      __builtin_amdgcn_sched_barrier(0);
      if (ABL == 5) { __builtin_amdgcn_sched_barrier(0); continue; }   // lab: no fetch, no wait, no barrier (compute + fragment reads only)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces have landed ...
      FLASH_STAMP(3);
      __syncthreads();                                   // ... and so have everybody else's
      FLASH_STAMP(4);
    }
    if (p.kv_len & 63) tile(ntiles - 1, (ntiles - 1) & 1, true);
    else tile(ntiles - 1, (ntiles - 1) & 1, false);

    if (ABL == 2 && lane == 0 && p.dbg != nullptr) {
#pragma unroll
      for (int i = 0; i < 5; ++i) atomicAdd(p.dbg + i, tacc[i]);
    }
    store_o();
  }
#undef FLASH_STAMP
#endif
}

// ---------------------------------------------------------------------------------------------------------
// attn_temporal_d72 (the long-sequence kernel: dispatched for T > 40, where the scores no longer fit the register-resident
// kernels below; also flash variant 9 as a cross-check): sequence = the T frames of one pixel token; batch = B*S; heads H.  qkv rows are the
// (b, t, s)-ordered tokens: q(b,t,s,h) at qkv + ((b*T + t)*S + s)*row_stride + h*72 (k at +C, v at +2C), so the
// "(B S) T C" view is just a stride of S rows.  One wave per (b, s, h); lane = (frame, third of the head dim).
// K,V rows (RMS-norm + RoPE applied to K) are parked in wave-private LDS as fp32; queries stay in registers.
// T is small (19 / 38): this kernel is HBM/latency bound, not MFMA work.
// grid: B*S*ceil(H/WPB) blocks of WPB waves; dynamic LDS = WPB * 2*T*72*4 bytes.
// ---------------------------------------------------------------------------------------------------------
__global__ void attn_temporal_d72_kernel(const bf16_t* __restrict__ qkv, int64_t row_stride, int C,
                                         const bf16_t* __restrict__ q_norm_w, const bf16_t* __restrict__ k_norm_w,
                                         const float* __restrict__ rope_cos, const float* __restrict__ rope_sin,
                                         bf16_t* __restrict__ out, int64_t out_stride, int B, int T, int S, int heads,
                                         int wpb, float eps, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hgroups = (heads + wpb - 1) / wpb;
  const int hg = blockIdx.x % hgroups;
  const int64_t bs = blockIdx.x / hgroups;
  const int s = (int)(bs % S), b = (int)(bs / S);
  const int h = hg * wpb + wave;
  if (h >= heads) return;  // no block-level barrier is used below
  float* ks = reinterpret_cast<float*>(smem) + (size_t)wave * 2 * T * HD;
  float* vs = ks + (size_t)T * HD;
  const int g = lane / 3, part = lane - g * 3;
  const bool lane_ok = lane < 63;
  const int npass = (T + 20) / 21;

  auto norm_rope = [&](float* x, const bf16_t* w, int t) {
    // LlamaRMSNorm over the 72 dims of the row (3 lanes), bf16 rounding, weight; then interleaved-pair RoPE at pos t
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 24; ++e) ss += x[e] * x[e];
    const int base = g * 3;
    const float tot = __shfl(ss, base, 64) + __shfl(ss, base + 1, 64) + __shfl(ss, base + 2, 64);
    const float rstd = rsqrtf(tot / (float)HD + eps);
    if (w != nullptr) {  // no qk-norm (Latte): q, k are used as projected
#pragma unroll
      for (int e = 0; e < 24; ++e) x[e] = bf2f(f2bf(bf2f(f2bf(x[e] * rstd)) * bf2f(w[part * 24 + e])));
    }
    if (rope_cos != nullptr) {
      const float* cs = rope_cos + (int64_t)t * HD + part * 24;
      const float* sn = rope_sin + (int64_t)t * HD + part * 24;
#pragma unroll
      for (int e = 0; e < 24; e += 2) {
        const float a = x[e], bb = x[e + 1];
        x[e] = bf2f(f2bf(a * cs[e] - bb * sn[e]));
        x[e + 1] = bf2f(f2bf(bb * cs[e + 1] + a * sn[e + 1]));
      }
    }
  };

  // ---- pass A: K (norm + rope) and V of every frame -> LDS
  for (int pss = 0; pss < npass; ++pss) {
    const int t = pss * 21 + g;
    const bool act = lane_ok && t < T;
    const int tt = t < T ? t : T - 1;
    const bf16_t* row = qkv + (((int64_t)b * T + tt) * S + s) * row_stride + h * HD + part * 24;
    float kx[24], vx[24];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      unpack8(*reinterpret_cast<const uint4*>(row + C + cc * 8), kx + cc * 8);
      unpack8(*reinterpret_cast<const uint4*>(row + 2 * C + cc * 8), vx + cc * 8);
    }
    norm_rope(kx, k_norm_w, tt);
    if (act) {
#pragma unroll
      for (int e = 0; e < 24; e += 4) {
        *reinterpret_cast<float4*>(ks + t * HD + part * 24 + e) = make_float4(kx[e], kx[e + 1], kx[e + 2], kx[e + 3]);
        *reinterpret_cast<float4*>(vs + t * HD + part * 24 + e) = make_float4(vx[e], vx[e + 1], vx[e + 2], vx[e + 3]);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

  // ---- pass B: queries, 21 frames per pass
  for (int pss = 0; pss < npass; ++pss) {
    const int t = pss * 21 + g;
    const bool act = lane_ok && t < T;
    const int tt = t < T ? t : T - 1;
    const bf16_t* row = qkv + (((int64_t)b * T + tt) * S + s) * row_stride + h * HD + part * 24;
    float qx[24];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) unpack8(*reinterpret_cast<const uint4*>(row + cc * 8), qx + cc * 8);
    norm_rope(qx, q_norm_w, tt);
    float m = NEG_BIG, l = 0.f;
    float acc[24];
#pragma unroll
    for (int e = 0; e < 24; ++e) acc[e] = 0.f;
    const int base = g * 3;
    for (int j = 0; j < T; ++j) {
      const float* kr = ks + j * HD + part * 24;
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < 24; e += 4) {
        const float4 kk = *reinterpret_cast<const float4*>(kr + e);
        d += qx[e] * kk.x + qx[e + 1] * kk.y + qx[e + 2] * kk.z + qx[e + 3] * kk.w;
      }
      const float sc = (__shfl(d, base, 64) + __shfl(d, base + 1, 64) + __shfl(d, base + 2, 64)) * scale;
      const float m_new = fmaxf(m, sc);
      const float alpha = __expf(m - m_new);
      const float pj = __expf(sc - m_new);
      l = l * alpha + pj;
      m = m_new;
      const float* vr = vs + j * HD + part * 24;
#pragma unroll
      for (int e = 0; e < 24; e += 4) {
        const float4 vv = *reinterpret_cast<const float4*>(vr + e);
        acc[e] = acc[e] * alpha + pj * vv.x;
        acc[e + 1] = acc[e + 1] * alpha + pj * vv.y;
        acc[e + 2] = acc[e + 2] * alpha + pj * vv.z;
        acc[e + 3] = acc[e + 3] * alpha + pj * vv.w;
      }
    }
    if (act) {
      const float inv = 1.0f / l;
#pragma unroll
      for (int e = 0; e < 24; ++e) acc[e] *= inv;
      bf16_t* orow = out + (((int64_t)b * T + t) * S + s) * out_stride + h * HD + part * 24;
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) *reinterpret_cast<uint4*>(orow + cc * 8) = pack8(acc + cc * 8);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------
// attn_temporal_d72_v2: same contract and lane mapping as attn_temporal_d72_kernel, restructured after its profile
// (1.86 TB/s, VALU-heavy): (i) q, k and v rows are fetched in ONE round trip; (ii) T <= TK keys: all scores are kept in
// registers and the softmax is two-pass (exact max, no per-key rescale of the 24 accumulators: 24 FMAs per key instead of
// 24 mul + 24 FMA + 2 exp); (iii) dot products and accumulation on float2 vectors (v_pk_fma_f32).
// ---------------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int TK>
__global__ __launch_bounds__(256) void attn_temporal_d72_v2_kernel(const bf16_t* __restrict__ qkv, int64_t row_stride, int C,
                                                                   const bf16_t* __restrict__ q_norm_w, const bf16_t* __restrict__ k_norm_w,
                                                                   const float* __restrict__ rope_cos, const float* __restrict__ rope_sin,
                                                                   bf16_t* __restrict__ out, int64_t out_stride, int B, int T, int S,
                                                                   int heads, int wpb, float eps, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hgroups = (heads + wpb - 1) / wpb;
  const int hg = blockIdx.x % hgroups;
  const int64_t bs = blockIdx.x / hgroups;
  const int s = (int)(bs % S), b = (int)(bs / S);
  const int h = hg * wpb + wave;
  if (h >= heads) return;  // no block-level barrier is used below
  float* ks = reinterpret_cast<float*>(smem) + (size_t)wave * 2 * T * HD;
  float* vs = ks + (size_t)T * HD;
  const int g = lane / 3, part = lane - g * 3;
  const bool lane_ok = lane < 63;
  const int npass = (T + 20) / 21;
  const int base = g * 3;

  auto norm_rope = [&](float* x, const bf16_t* w, int t) {
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 24; ++e) ss += x[e] * x[e];
    const float tot = __shfl(ss, base, 64) + __shfl(ss, base + 1, 64) + __shfl(ss, base + 2, 64);
    const float rstd = rsqrtf(tot / (float)HD + eps);
    if (w != nullptr) {
#pragma unroll
      for (int e = 0; e < 24; ++e) x[e] = bf2f(f2bf(bf2f(f2bf(x[e] * rstd)) * bf2f(w[part * 24 + e])));
    }
    if (rope_cos != nullptr) {
      const float* cs = rope_cos + (int64_t)t * HD + part * 24;
      const float* sn = rope_sin + (int64_t)t * HD + part * 24;
#pragma unroll
      for (int e = 0; e < 24; e += 2) {
        const float a = x[e], bb = x[e + 1];
        x[e] = bf2f(f2bf(a * cs[e] - bb * sn[e]));
        x[e + 1] = bf2f(f2bf(bb * cs[e + 1] + a * sn[e + 1]));
      }
    }
  };

  // ---- one round trip: q, k, v of this lane's frame (first pass) are all requested before anything is consumed
  uint4 rq[3], rk[3], rv[3];
  {
    const int tt = g < T ? g : T - 1;
    const bf16_t* row = qkv + (((int64_t)b * T + tt) * S + s) * row_stride + h * HD + part * 24;
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      rk[cc] = *reinterpret_cast<const uint4*>(row + C + cc * 8);
      rv[cc] = *reinterpret_cast<const uint4*>(row + 2 * C + cc * 8);
      rq[cc] = *reinterpret_cast<const uint4*>(row + cc * 8);
    }
  }
  for (int pss = 0; pss < npass; ++pss) {
    const int t = pss * 21 + g;
    const bool act = lane_ok && t < T;
    const int tt = t < T ? t : T - 1;
    if (pss > 0) {
      const bf16_t* row = qkv + (((int64_t)b * T + tt) * S + s) * row_stride + h * HD + part * 24;
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        rk[cc] = *reinterpret_cast<const uint4*>(row + C + cc * 8);
        rv[cc] = *reinterpret_cast<const uint4*>(row + 2 * C + cc * 8);
      }
    }
    float kx[24], vx[24];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      unpack8(rk[cc], kx + cc * 8);
      unpack8(rv[cc], vx + cc * 8);
    }
    norm_rope(kx, k_norm_w, tt);
    if (act) {
#pragma unroll
      for (int e = 0; e < 24; e += 4) {
        *reinterpret_cast<float4*>(ks + t * HD + part * 24 + e) = make_float4(kx[e], kx[e + 1], kx[e + 2], kx[e + 3]);
        *reinterpret_cast<float4*>(vs + t * HD + part * 24 + e) = make_float4(vx[e], vx[e + 1], vx[e + 2], vx[e + 3]);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

  for (int pss = 0; pss < npass; ++pss) {
    const int t = pss * 21 + g;
    const bool act = lane_ok && t < T;
    const int tt = t < T ? t : T - 1;
    if (pss > 0) {
      const bf16_t* row = qkv + (((int64_t)b * T + tt) * S + s) * row_stride + h * HD + part * 24;
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) rq[cc] = *reinterpret_cast<const uint4*>(row + cc * 8);
    }
    float qx[24];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) unpack8(rq[cc], qx + cc * 8);
    norm_rope(qx, q_norm_w, tt);
    f32x2 q2[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) q2[e] = f32x2{qx[2 * e] * scale, qx[2 * e + 1] * scale};  // fold the softmax scale into q
    float sc[TK];
    float m = NEG_BIG;
#pragma unroll
    for (int j = 0; j < TK; ++j) {
      sc[j] = NEG_BIG;
      if (j < T) {
        const float* kr = ks + j * HD + part * 24;
        f32x2 d2 = f32x2{0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 24; e += 4) {
          const float4 kk = *reinterpret_cast<const float4*>(kr + e);
          d2 = __builtin_elementwise_fma(q2[e / 2], f32x2{kk.x, kk.y}, d2);
          d2 = __builtin_elementwise_fma(q2[e / 2 + 1], f32x2{kk.z, kk.w}, d2);
        }
        const float d = d2.x + d2.y;
        sc[j] = __shfl(d, base, 64) + __shfl(d, base + 1, 64) + __shfl(d, base + 2, 64);
        m = fmaxf(m, sc[j]);
      }
    }
    f32x2 acc[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) acc[e] = f32x2{0.f, 0.f};
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < TK; ++j) {
      if (j < T) {
        const float pj = __expf(sc[j] - m);
        l += pj;
        const f32x2 p2 = f32x2{pj, pj};
        const float* vr = vs + j * HD + part * 24;
#pragma unroll
        for (int e = 0; e < 24; e += 4) {
          const float4 vv = *reinterpret_cast<const float4*>(vr + e);
          acc[e / 2] = __builtin_elementwise_fma(p2, f32x2{vv.x, vv.y}, acc[e / 2]);
          acc[e / 2 + 1] = __builtin_elementwise_fma(p2, f32x2{vv.z, vv.w}, acc[e / 2 + 1]);
        }
      }
    }
    if (act) {
      const float inv = 1.0f / l;
      float o[24];
#pragma unroll
      for (int e = 0; e < 12; ++e) { o[2 * e] = acc[e].x * inv; o[2 * e + 1] = acc[e].y * inv; }
      bf16_t* orow = out + (((int64_t)b * T + t) * S + s) * out_stride + h * HD + part * 24;
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) *reinterpret_cast<uint4*>(orow + cc * 8) = pack8(o + cc * 8);
    }
  }
}

}  // namespace

// A/B selector of the measurement tools and the kernel-equivalence tests (vsys_tune_flash_variant): process-wide, read once per
// launch.  Every id the SHIPPED build accepts selects a valid kernel for the same contract.
static std::atomic<int> g_flash_variant_a{0};
static unsigned long long* g_flash_dbg = nullptr;
// 14 / 15 = the 64-rows-per-wave kernel (attention_w64.hip) wherever it applies / never;
// 0 = shipped default (resident-K/V kernel for <= 320 keys and many query rows), 3 = three workgroups per CU, 4 / 9 = force the VALU (v2) /
// online-softmax temporal kernels, 8 = resident-K/V kernel whenever the keys fit, 10 = never, 12 = the d64 kernel (CogVideoX) on its
// two-stage K/V ring instead of the shipped three-stage one: all valid.  1 (K/V tiles not
// fetched: output NOT valid) and 2 (phase timers) exist in -DVSYS_LAB builds only.
int set_flash_variant(int v) {
  switch (v) {
    case 0: case 3: case 4: case 8: case 9: case 10: case 12: case 14: case 15: case 16: case 17: case 18: case 19: case 21: case 22: case 23: case 140: case 141: case 143: case 144: break;
#ifdef VSYS_LAB
    case 1: case 2: case 6: case 146: case 147: case 148: case 149: case 150: break;
#endif
    default: return VSYS_ERR_ARG;
  }
  g_flash_variant_a.store(v, std::memory_order_relaxed);
  return 0;
}
int get_flash_variant() { return g_flash_variant_a.load(std::memory_order_relaxed); }
void set_flash_debug_buffer(void* p) { g_flash_dbg = reinterpret_cast<unsigned long long*>(p); }
void* get_lab_debug_buffer() { return g_flash_dbg; }

int launch_attn_prep_kv(const bf16_t* k, int64_t k_stride, const bf16_t* v, int64_t v_stride, const bf16_t* k_norm_w,
                        bf16_t* kp, bf16_t* vt, int batch, int heads, int kv_len, int kv_pad, float eps,
                        hipStream_t stream) {
  if (batch <= 0 || heads <= 0 || kv_len <= 0) return 0;
  if (kv_pad % 64 != 0 || kv_pad < kv_len || (k_stride % 8) || (v_stride % 8)) return VSYS_ERR_SHAPE;
  dim3 grid(kv_pad / 64, batch * heads);
  hipLaunchKernelGGL(attn_prep_kv_kernel, grid, dim3(256), 0, stream, k, k_stride, v, v_stride, k_norm_w, kp, vt, heads,
                     kv_len, kv_pad, eps, 0.11785113019775793f * 1.4426950408889634f /* 72^-0.5 * log2(e) */);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_flash_attn_d72(const bf16_t* q, int64_t q_stride, const bf16_t* q_norm_w, const bf16_t* kp, const bf16_t* vt,
                          bf16_t* out, int64_t out_stride, int batch, int heads, int q_len, int kv_len, int kv_pad,
                          float eps, float k_bound, hipStream_t stream, bool keys_exact) {
  if (batch <= 0 || heads <= 0 || q_len <= 0) return 0;
  if (kv_len <= 0 || kv_pad % 64 != 0 || kv_pad < kv_len || (q_stride % 8) || (out_stride % 4)) return VSYS_ERR_SHAPE;
  FlashParams p;
  p.q = q; p.q_stride = q_stride; p.q_norm_w = q_norm_w; p.kp = kp; p.vt = vt; p.out = out; p.out_stride = out_stride;
  p.heads = heads; p.q_len = q_len; p.kv_len = kv_len; p.kv_pad = kv_pad; p.eps = eps;
  p.nqb = (q_len + 127) / 128;
  p.chunks = 1;
  const int64_t nblk = (int64_t)p.nqb * batch * heads;
  if (nblk > 0x7fffffff) return VSYS_ERR_SHAPE;
  dim3 grid((unsigned)nblk);
  const size_t lds = 2 * KV_STAGE;
  p.dbg = g_flash_dbg;
  const int g_flash_variant = g_flash_variant_a.load(std::memory_order_relaxed);
  // few keys (cross attention: <= 320 text tokens) and enough query rows to give every CU one workgroup: resident-K/V kernel
  if ((g_flash_variant == 0 || g_flash_variant == 8) && (kv_len + 63) / 64 <= RES_MAX_TILES) {
    const int ncu = cu_count_this_device();
    const int nqb = (q_len + 255) / 256;
    int chunks = ncu / (batch * heads);
    chunks = chunks < 1 ? 1 : (chunks > nqb ? nqb : chunks);
    // worth it when a workgroup walks several query blocks (the K/V load is paid once per workgroup); variant 8 forces it
    const bool fits = (int64_t)q_len * q_stride * 2 < 0x7fffffff;   // 32-bit row offsets in the Q staging
    if (fits && (g_flash_variant == 8 || nqb >= 2 * chunks)) {
      p.chunks = chunks;
      p.nqb = nqb;
      static std::atomic<unsigned long long> attr_seen{0};
      for (DeviceOnce once(attr_seen); once.todo(); once.done()) {
        (void)hipFuncSetAttribute((const void*)flash_attn_d72_kernel<0, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, RES_MAX_TILES * KV_STAGE + 8 * RES_Q_BYTES);
        (void)hipFuncSetAttribute((const void*)flash_attn_d72_kernel<0, 2, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, RES_MAX_TILES * KV_STAGE + 8 * RES_Q_BYTES);
      }
      // the caller's promise about the keys behind kv_len (vsys_flash_attn_d72_exact) only matters for a ragged count; VSYS_FLASH_EXACT=0 ignores it
      static const bool exact_ok = [] { const char* e = getenv("VSYS_FLASH_EXACT"); return !(e && e[0] == '0'); }();
      if (keys_exact && exact_ok && (kv_len & 63) != 0)
        hipLaunchKernelGGL((flash_attn_d72_kernel<0, 2, true, true>), dim3((unsigned)(chunks * batch * heads)), dim3(512), RES_MAX_TILES * KV_STAGE + 8 * RES_Q_BYTES, stream, p);
      else
        hipLaunchKernelGGL((flash_attn_d72_kernel<0, 2, true>), dim3((unsigned)(chunks * batch * heads)), dim3(512), RES_MAX_TILES * KV_STAGE + 8 * RES_Q_BYTES, stream, p);
      return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
    }
  }
  // long key sequences: 64 query rows per wave, one wave per SIMD, hand-allocated instruction stream (attention_w64.hip)
  // (opt-in, VSYS_FLASH_W64=1 or variant 16: in situ it ties the 32-row kernel — 104.9 vs 104.6 ms per step, profiles/r04_flash_w64p_insitu.txt)
  static const bool w64_default = [] { const char* e = getenv("VSYS_FLASH_W64"); return e && e[0] == '1'; }();
  // 16 = the persistent form of the w64 kernel (one workgroup per CU walks the query blocks; any tile count >= 4)
  // It is the default where every workgroup walks at least four items (the seams of an item are what the walk amortises); the
  // one-item-per-workgroup form below is selectable only (it loses to the 32-row kernel: profiles/r04_flash_w64_placement_kbench.txt).
  // With the caller's bound on the Kp row norms (k_bound > 0, vsys_flash_attn_d72_kb) the statements run WITHOUT the running max
  // (FlashW64Params::k_bound): 17 / 18 force the one-item / persistent form of that, 19 ignores the bound.
  static const bool static_ok = [] { const char* e = getenv("VSYS_FLASH_STATIC"); return !(e && e[0] == '0'); }();
  const bool bounded = k_bound > 0.f && q_norm_w != nullptr && static_ok && g_flash_variant != 19 && g_flash_variant != 15;
  const bool many_items = (int64_t)batch * heads * ((q_len + 255) / 256) >= 4ll * cu_count_this_device();
  // (the persistent form stays opt-in — VSYS_FLASH_W64=1 — also with the promise: 0.228 vs 0.254 ms isolated at the config-2 shape,
  // but 101.1 vs 100.8 ms per step in situ, profiles/r04_flash_static_max.json)
  const bool w64p_default = g_flash_variant == 0 && kv_len >= 512 && many_items && w64_default;
  if ((g_flash_variant == 16 || g_flash_variant == 146 || g_flash_variant == 18 || w64p_default) &&
      flash_w64p_supports(q_len, kv_len, kv_pad, q_stride))
    return launch_flash_attn_d72_w64p(q, q_stride, q_norm_w, kp, vt, out, out_stride, batch, heads, q_len, kv_len, kv_pad, eps,
                                      g_flash_variant == 146, bounded && g_flash_variant != 16 && g_flash_variant != 146 ? k_bound : 0.f,
                                      stream);
  constexpr int W64_DEFAULT_VAR = 1;   // 140 / 141 / 143 select placement variant 0 / 1 / 3 (lab builds: 148 / 149 = ablations 8 / 9)
  // Long key sequences (720p frames: 3600 keys = 57 tiles per item) amortise the one-item form's seams: 1003 vs 915 TFLOP/s against
  // the 32-row kernel at 76 x 16 x 3600^2, same bits (profiles/r04_flash_w64_720p.json); at 1024 keys it loses 5 %.
  static const bool w64_off = [] { const char* e = getenv("VSYS_FLASH_W64"); return e && e[0] == '0'; }();
  const bool w64_long = g_flash_variant == 0 && kv_len >= 2048 && !w64_off;
  if ((g_flash_variant == 14 || g_flash_variant == 17 || g_flash_variant >= 140 || w64_long) &&
      flash_w64_supports(q_len, kv_len, kv_pad)) {
    int var = g_flash_variant >= 140 && g_flash_variant != 144 ? g_flash_variant - 140 : W64_DEFAULT_VAR;
    if (bounded && (g_flash_variant == 17 || g_flash_variant == 0)) var = 5;
    if (g_flash_variant == 17 && var != 5) return VSYS_ERR_ARG;
    return launch_flash_attn_d72_w64(q, q_stride, q_norm_w, kp, vt, out, out_stride, batch, heads, q_len, kv_len, kv_pad, eps, var,
                                     k_bound, stream);
  }
  // three workgroups per CU pay for long, unmasked key sequences (spatial attention: 0.241 vs 0.251 ms); with a masked last tile
  // the 168-register variant spills in the peeled tile (cross shape 0.156 vs 0.099 ms)
  static const bool wps3_ok = [] { const char* e = getenv("VSYS_FLASH_WPS3"); return !(e && e[0] == '0'); }();
#ifdef VSYS_LAB
  if (g_flash_variant == 1 || g_flash_variant == 2) {
    if (g_flash_variant == 1) hipLaunchKernelGGL((flash_attn_d72_kernel<1, 3>), grid, dim3(256), lds, stream, p);
    else hipLaunchKernelGGL((flash_attn_d72_kernel<2, 3>), grid, dim3(256), lds, stream, p);   // (three workgroups per CU, as the shipped spatial launch)
    return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
  }
#endif
#ifdef VSYS_LAB
  if (g_flash_variant == 6) {   // no fetch, no wait, no barrier: output NOT valid
    hipLaunchKernelGGL((flash_attn_d72_kernel<5, 3>), grid, dim3(256), lds, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
  }
#endif
  if (g_flash_variant == 23 && kv_len >= 512 && (kv_len & 63) == 0) {   // A/B (same shapes as the shipped rule below): the next tile's LDS-DMA pieces in front of the tile instead of between its first QK MFMAs (three workgroups per CU)
    hipLaunchKernelGGL((flash_attn_d72_kernel<4, 3>), grid, dim3(256), lds, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
  }
  if (g_flash_variant == 3 || (g_flash_variant == 0 && wps3_ok && kv_len >= 512 && (kv_len & 63) == 0))
    hipLaunchKernelGGL((flash_attn_d72_kernel<0, 3>), grid, dim3(256), lds, stream, p);
  else hipLaunchKernelGGL((flash_attn_d72_kernel<0, 2>), grid, dim3(256), lds, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

int launch_attn_temporal_d72(const bf16_t* qkv, int64_t row_stride, int C, const bf16_t* q_norm_w, const bf16_t* k_norm_w,
                             const float* rope_cos, const float* rope_sin, bf16_t* out, int64_t out_stride, int B, int T,
                             int S, int heads, float eps, hipStream_t stream) {
  if (B <= 0 || T <= 0 || S <= 0 || heads <= 0) return 0;
  if ((row_stride % 8) || (out_stride % 8) || (C % 8) || ((q_norm_w == nullptr) != (k_norm_w == nullptr))) return VSYS_ERR_SHAPE;
  const size_t per_wave = (size_t)2 * T * HD * sizeof(float);
  int wpb = 4;
  while (wpb > 1 && per_wave * wpb > 64 * 1024) wpb >>= 1;
  if (per_wave * wpb > 160 * 1024) return VSYS_ERR_SHAPE;
  const int hgroups = (heads + wpb - 1) / wpb;
  const int64_t grid = (int64_t)B * S * hgroups;
  if (grid > 0x7fffffff) return VSYS_ERR_SHAPE;
  const size_t lds = per_wave * wpb;
  const float scale = 0.11785113019775793f;
  const int g_flash_variant = g_flash_variant_a.load(std::memory_order_relaxed);
  if (T <= 64 && g_flash_variant != 9 && g_flash_variant != 4)   // the MFMA formulation (attention_t3.hip: one 32-frame tile, or two); 4 = force the v2 kernel
    return launch_attn_temporal_d72_v3(qkv, row_stride, C, q_norm_w, k_norm_w, rope_cos, rope_sin, out, out_stride, B, T, S, heads, eps,
                                       scale, stream, /* the reference's rounding points, stage by stage (A/B id 21) */ g_flash_variant == 21, /* 22: the per-lane-load kernel instead of the LDS-DMA form */ g_flash_variant == 22);
  if (T <= 20 && g_flash_variant != 9) {
    hipLaunchKernelGGL(attn_temporal_d72_v2_kernel<20>, dim3((unsigned)grid), dim3(64 * wpb), lds, stream, qkv, row_stride, C, q_norm_w,
                       k_norm_w, rope_cos, rope_sin, out, out_stride, B, T, S, heads, wpb, eps, scale);
  } else if (T <= 40 && g_flash_variant != 9) {
    hipLaunchKernelGGL(attn_temporal_d72_v2_kernel<40>, dim3((unsigned)grid), dim3(64 * wpb), lds, stream, qkv, row_stride, C, q_norm_w,
                       k_norm_w, rope_cos, rope_sin, out, out_stride, B, T, S, heads, wpb, eps, scale);
  } else {  // long sequences (or lab variant 9): the online-softmax kernel
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)attn_temporal_d72_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attn_temporal_d72_kernel, dim3((unsigned)grid), dim3(64 * wpb), lds, stream, qkv, row_stride, C, q_norm_w,
                       k_norm_w, rope_cos, rope_sin, out, out_stride, B, T, S, heads, wpb, eps, scale);
  }
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

}  // namespace vsys
