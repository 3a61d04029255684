// Internal launch declarations shared by the .hip translation units and the C-ABI layer (capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <vector>

#include "../../include/videosys_amd.h"

namespace vsys {

typedef uint16_t bf16_t;

// One-time per-DEVICE work of a launcher (hipFuncSetAttribute for > 64 KiB of dynamic LDS is a per-device property of the
// function).  Use:  for (DeviceOnce once(seen); once.todo(); once.done()) { ...attribute calls... }
// The device's bit in ``seen`` is set AFTER the body ran: a second host thread that arrives while the first is still inside the body
// runs the (idempotent) body itself instead of launching before the limit is raised — the bit used to be set before the body.
// Not a stream operation, so it is legal while a stream is being captured into a hipGraph.
class DeviceOnce {
 public:
  explicit DeviceOnce(std::atomic<unsigned long long>& seen) : seen_(seen) {
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev <= 63) bit_ = 1ull << dev;
    todo_ = bit_ == 0 || (seen_.load(std::memory_order_acquire) & bit_) == 0;
  }
  bool todo() const { return todo_; }
  void done() {
    if (bit_ != 0) seen_.fetch_or(bit_, std::memory_order_release);
    todo_ = false;
  }

 private:
  std::atomic<unsigned long long>& seen_;
  unsigned long long bit_ = 0;
  bool todo_ = false;
};
// multiprocessor count of the current device (cached per device)
inline int cu_count_this_device() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return 256;
  int n = cache[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    hipDeviceProp_t prop;
    n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    cache[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

enum { EPI_BIAS = VSYS_EPI_BIAS, EPI_BIAS_GELU = VSYS_EPI_BIAS_GELU, EPI_GATE_RES = VSYS_EPI_GATE_RES,
       // AdaLN folded into the GEMM (vsys_gemm_bf16_ln): W is the pre-scaled weight, the epilogue applies the row statistics
       EPI_LN_BIAS = 3, EPI_LN_GELU = 4,
       // EPI_GATE_RES that also emits the LayerNorm partials of the rows it stores (vsys_gemm_bf16_stats); no aux
       EPI_GATE_RES_STATS = 5,
       EPI_F32_SLICES = 100 /* internal (gemm2_bf16.hip): fp32 K-slice partials, see launch_gemm2_slices */ };
constexpr int LN_BLOCK = 96;   // columns per LayerNorm partial (= the column width of a GEMM wave tile)
enum { ACT_NONE = VSYS_ACT_NONE, ACT_SILU = VSYS_ACT_SILU, ACT_GELU_TANH = VSYS_ACT_GELU_TANH };

struct GemmParams {
  const bf16_t* A; int64_t lda;   // activations [M, K]
  const bf16_t* W; int64_t ldw;   // nn.Linear weight [N, K]
  const bf16_t* bias;             // [N] or null
  bf16_t* out; int64_t ldo;       // [M, N]
  int M, N, K;
  const bf16_t* gate; int64_t gate_stride;  // per-sample gate row [N] at gate + sample*gate_stride, or null (gate = 1)
  const bf16_t* res; int64_t ldr;           // residual [M, N] or null
  bf16_t* aux; int64_t ldaux;               // optional copy of gate*(acc+bias) (PAB cache slab) or null
  int rows_per_sample;
  // CogVideoX joint [text | video] rows: rows whose position inside the sample is < seg_split take their gate vector
  // gate_alt elements further on (enc_gate vs gate of CogVideoXLayerNormZero); 0 = one gate per sample
  int seg_split; int64_t gate_alt;
  // EPI_F32_SLICES only (launch_gemm2_slices): K slice length, fp32 output [slices][N][M] (element (m, n) of slice s at
  // out32[s * slab + n * ldo32 + m]: "A" is the weight here, so this is activation-row-major)
  int ks; float* out32; int64_t slab, ldo32;
  // AdaLN fold.  Row statistics travel as per-96-column partials (mean_b, M2_b) at stats[b * ld + row], b = column / 96:
  //   EPI_GATE_RES_STATS writes them for x_new = res + gate (acc + bias) (what it stores to out);
  //   EPI_LN_BIAS / EPI_LN_GELU combine the ln_nb partials of a row into (mu, rstd) and compute
  //   out = rstd (acc - mu cs[n]) + cv[n]  with W = bf16(W0 (1 + scale)), cs[n] = sum_k W[n][k], cv[n] = shift . W0[n] + bias[n].
  const float* cs = nullptr; const float* cv = nullptr;
  const float2* ln_stats = nullptr; int64_t ln_ld = 0; int ln_nb = 0; float ln_eps = 0.f;
  float2* stats_out = nullptr; int64_t stats_ld = 0;
  // EPI_GATE_RES only: up to two more rows-of-x operands (leading dimension ldr) added AFTER the residual, each with its own bf16
  // rounding — out = bf16(bf16(bf16(res + u) + add1) + add2): the `x += slab` passes of PAB broadcasts that follow this GEMM in
  // program order, folded into its store phase.  With stats_out set the same epilogue also emits the LayerNorm partials of what it
  // stored (same bits as EPI_GATE_RES_STATS / ln_row_stats give for those rows).
  const bf16_t* add1 = nullptr; const bf16_t* add2 = nullptr;
  // tile raster (common.h gemm_raster), filled in by launch_gemm: column-group width / panel-chunk height; 6 / 0 = the default
  int raster_gw = 6, raster_ph = 0;
  // two-way split K of the 128-row geometry (gemm_bf16.hip KS): fp32 partial sums [tile][wave][24][64 lanes] float4 and one flag per
  // (tile, wave), both owned by the library (one workspace per stream); filled in by launch_gemm
  float* sk_ws = nullptr; int* sk_flags = nullptr;
};

// implicit-GEMM convolution / 128-column GEMM (conv_bf16.hip).  A points at the row that tap (0,0,0) reads for output row 0.
struct ConvParams {
  const bf16_t* A; int64_t lda;   // activation rows [*, cin]
  const bf16_t* W; int64_t ldw;   // [N, taps*cin], k = tap*cin + channel
  const bf16_t* bias;             // [N] or null
  const bf16_t* res; int64_t ldr; // residual rows [M, N] added after rounding, or null
  bf16_t* out; float* out32; int64_t ldo;  // exactly one of out / out32
  int M, N, K;
  int cin, taps, cshift, taps_hw, kw, row_pitch, plane_pitch, max_tap_rows;
  int batch; int64_t batch_a, batch_w, batch_o;  // gridDim.y operand strides in elements
  float out_scale;                // fp32 output only
};
int launch_conv(const ConvParams& p, hipStream_t stream);

// activation grid of the VAE kernels (vae_ops.hip): row(n,t,h,w) = n*sample_rows + ((t+tf)*(H+2pad) + h+pad)*(W+2pad) + w+pad
struct VaeGrid {
  int T, H, W, pad, tf;
  int64_t sample_rows;
};
int launch_gn_stats(const bf16_t* x, const VaeGrid& g, int N, int C, int groups, float eps, float* partial, int nblk, float* stats,
                    hipStream_t stream);
int launch_gn_apply(const bf16_t* x, const VaeGrid& gs, bf16_t* y, const VaeGrid& gd, int N, int C, int groups, const float* stats,
                    const bf16_t* gamma, const bf16_t* beta, int act, hipStream_t stream);
int launch_regrid(const bf16_t* x, const VaeGrid& gs, bf16_t* y, const VaeGrid& gd, int N, int C, int up, int tmode, hipStream_t stream);
int launch_subsample(const bf16_t* x, const VaeGrid& gs, bf16_t* y, const VaeGrid& gd, int N, int C, int st, int ss, int t0, int s0,
                     hipStream_t stream);
int launch_spatial_norm_apply(const bf16_t* x, const VaeGrid& gs, bf16_t* y, const VaeGrid& gd, int N, int C, int groups,
                              const float* stats, const bf16_t* gamma, const bf16_t* beta, const bf16_t* yb, int zT, int zH, int zW,
                              hipStream_t stream);
int launch_blend_edge(const bf16_t* a, bf16_t* b, int64_t outer, int Ha, int Wa, int Hb, int Wb, int ext, int axis, hipStream_t stream);
int launch_d2s_time(const bf16_t* x, const VaeGrid& gs, bf16_t* y, const VaeGrid& gd, int N, int Cout, hipStream_t stream);
int launch_vae_first_im2col(const bf16_t* z, int F, int H, int W, int kt, int kcols, const float* scale, const float* shift,
                            const float* pq_w, const float* pq_b, bf16_t* out, hipStream_t stream);
int launch_extract_planar(const bf16_t* x, const VaeGrid& g, int N, int ldx, int nc, int tskip, bf16_t* out, int64_t Ftot, int f0,
                          hipStream_t stream);
int launch_softmax_rows(const float* s, bf16_t* p, int64_t rows, int n, int ld, hipStream_t stream);

// T5 encoder pieces (t5_ops.hip)
int launch_gather_rows(const bf16_t* table, const int64_t* ids, bf16_t* out, int64_t n, int C, int64_t vocab, hipStream_t stream);
int launch_rms_norm_rows(const bf16_t* x, const bf16_t* w, bf16_t* y, int64_t rows, int C, float eps, hipStream_t stream);
int launch_geglu(const bf16_t* h, bf16_t* out, int64_t rows, int F, hipStream_t stream);
int launch_t5_attention(const bf16_t* qkv, int64_t row_stride, int inner, const float* relbias, const int* klen, bf16_t* out,
                        int64_t out_stride, int B, int L, int heads, hipStream_t stream);
int launch_splitk_reduce_t(const float* part, int S, int64_t slab, int ldp, const bf16_t* res, int64_t ldr, bf16_t* out, int64_t ldo,
                           int M, int N, hipStream_t stream);
int launch_splitk_reduce(const float* part, int S, int64_t slab, int64_t ldp, const bf16_t* res, int64_t ldr, bf16_t* out, int64_t ldo,
                         int M, int N, hipStream_t stream);
int launch_t5_attention_mfma(const bf16_t* qkv, int64_t row_stride, int inner, const float* bias, int bias_ld, int bias_center,
                             int kv_len, bf16_t* kp, bf16_t* vt, bf16_t* out, int64_t out_stride, int L, int heads,
                             hipStream_t stream);

int launch_gemm(const GemmParams& p, int epi, hipStream_t stream);
int launch_gemm2(const GemmParams& p, int epi, int wide, hipStream_t stream);
int launch_gemm2_slices(const GemmParams& p, int slices, hipStream_t stream);
int launch_gemm3(const GemmParams& p, int epi, hipStream_t stream);
// ping-pong wave groups (gemm4_bf16.hip); persistent = 1: one workgroup per CU walks the tiles, 2: + stream-K tail
int launch_gemm4(const GemmParams& p, int epi, int persistent, hipStream_t stream);
bool gemm4_supports(const GemmParams& p, int epi);
bool sk_plan(int ntiles, int nt, int grid, std::vector<int4>& segs, int& nseg_max);  // stream-K segment lists (host)
int launch_gemm4_lab(const GemmParams& p, int abl, int persistent, hipStream_t stream);
int launch_gemm2_stamp(const GemmParams& p, hipStream_t stream);  // lab: per-stage cycle stamps into p.aux (int64)
int set_gemm_variant(int v);   // VSYS_ERR_ARG for ids this build does not contain
int set_flash_variant(int v);
int get_flash_variant();
void set_flash_debug_buffer(void* p);
void* get_lab_debug_buffer();
int launch_linear_small(const bf16_t* x, int64_t ldx, const bf16_t* w, int64_t ldw, const bf16_t* bias, bf16_t* out,
                        int64_t ldo, int M, int N, int K, int act_in, int act_out, hipStream_t stream);
int launch_adaln_modulate(const bf16_t* x, const bf16_t* shift, const bf16_t* scale, bf16_t* y, int64_t rows, int C,
                          int64_t rows_per_sample, int64_t mod_stride, float eps, hipStream_t stream);
// AdaLN fold (adaln_fold.hip): pre-scaled weights + column sums of every site of a step; row statistics of a tensor
int launch_adaln_prescale(const int64_t* sites, int nsites, int64_t nblocks, const bf16_t* mod, hipStream_t stream);
int launch_ln_row_stats(const bf16_t* x, int64_t rows, int C, float2* stats, int64_t ld, hipStream_t stream);
int launch_mod_table(const bf16_t* table, const bf16_t* t_mlp, bf16_t* out, int nblk, int B, int C6, hipStream_t stream);
int launch_timestep_embedding(const float* t, bf16_t* out, int B, int dim, hipStream_t stream);
int launch_patch_embed(const float* x, int Bz, const bf16_t* w, const bf16_t* bias, const bf16_t* pos, bf16_t* out, int B,
                       int Cin, int T, int H, int W, int ph, int pw, int C, int s0, int Sl, hipStream_t stream);  // Sl < 0: whole frame
int launch_unpatchify_tokens(const float* tok, float* out, int P, int B, int T, int Sl, int Hp, int Wp, int H, int W, int ph,
                             int pw, int Cout, hipStream_t stream);
int launch_final_layer(const bf16_t* x, const bf16_t* table, const bf16_t* tvec, const bf16_t* w, const bf16_t* bias,
                       float* out, int B, int T, int Hp, int Wp, int H, int W, int ph, int pw, int Cout, int C, float eps,
                       int Sl, float* tokens, hipStream_t stream);   // Sl < 0, tokens = null: whole frames -> pixels
int launch_cfg_euler(float* z, const float* model_out, int Bz, int Cin, int Cout, int64_t thw, float guidance, float dt,
                     hipStream_t stream);
int launch_cfg_axpby(float* z, const float* model_out, int Bz, int Cin, int Cout, int64_t thw, float guidance, float c_z,
                     float c_eps, int cond_first, hipStream_t stream);
int launch_add_bcast_rows(bf16_t* x, const bf16_t* e, int64_t rows, int C, int64_t group, int64_t period, hipStream_t stream);
int launch_add_rows(bf16_t* x, const bf16_t* y, int64_t n, hipStream_t stream);
int launch_copy_4d(const bf16_t* src, bf16_t* dst, int n0, int n1, int n2, int C, int64_t ss0, int64_t ss1, int64_t ss2,
                   int64_t ds0, int64_t ds1, int64_t ds2, int n1_valid, int n2_valid, hipStream_t stream);
constexpr int VSYS_COPY_BATCH_MAX = 16;
struct CopyDesc {  // one copy_4d problem relative to a common (src, dst) pair; offsets and strides in elements
  int64_t src_off, dst_off;
  int n0, n1, n2, C;
  int64_t ss0, ss1, ss2, ds0, ds1, ds2;
  int n1_valid, n2_valid;
};
int launch_copy_4d_batch(const bf16_t* src, bf16_t* dst, const CopyDesc* ops, int nops, hipStream_t stream);
// one-kernel peer-to-peer DSP exchange (p2p.hip): problem i copies into dsts[i] (+ ops[i].dst_off) and signals peer_flags[i]
int launch_p2p_exchange(const bf16_t* src, const CopyDesc* ops, bf16_t* const* dsts, unsigned* const* peer_flags, const int* remote, int nops,
                        const unsigned* my_flags, int n_flags, int self_index, unsigned* state, long long timeout_ticks,
                        hipStream_t stream);
int launch_attn_prep_kv(const bf16_t* k, int64_t k_stride, const bf16_t* v, int64_t v_stride, const bf16_t* k_norm_w,
                        bf16_t* kp, bf16_t* vt, int batch, int heads, int kv_len, int kv_pad, float eps,
                        hipStream_t stream);
int launch_flash_attn_d72(const bf16_t* q, int64_t q_stride, const bf16_t* q_norm_w, const bf16_t* kp, const bf16_t* vt,
                          bf16_t* out, int64_t out_stride, int batch, int heads, int q_len, int kv_len, int kv_pad,
                          float eps, float k_bound, hipStream_t stream,    // k_bound: see FlashW64Params::k_bound (0 = none)
                          bool keys_exact = false);   // the caller's promise of vsys_flash_attn_d72_exact (attention.hip, EXACT)
// 64 query rows per wave, one wave per SIMD, hand-allocated tile loop (attention_w64.hip); same contract as launch_flash_attn_d72
bool flash_w64_supports(int q_len, int kv_len, int kv_pad);
int launch_flash_attn_d72_w64(const bf16_t* q, int64_t q_stride, const bf16_t* q_norm_w, const bf16_t* kp, const bf16_t* vt, bf16_t* out,
                              int64_t out_stride, int batch, int heads, int q_len, int kv_len, int kv_pad, float eps, int var,
                              float k_bound, hipStream_t stream);   // var: LDS-DMA placement variant 0 / 1 / 3, 5 = no running max (k_bound)
bool flash64_w64_supports(int q_len, int kv_len);
int launch_flash_attn_d64_w64(const bf16_t* q, int64_t q_stride, const bf16_t* ln_w, const bf16_t* ln_b, const float* rope_cos,
                              const float* rope_sin, int rope_start, int rope_len, const bf16_t* kp, const bf16_t* vt, bf16_t* out,
                              int64_t out_stride, int batch, int heads, int q_len, int kv_len, int kv_pad, float eps, int var,
                              float k_bound, hipStream_t stream);
bool flash_w64p_supports(int q_len, int kv_len, int kv_pad, int64_t q_stride);   // persistent form of the w64 kernel
int launch_flash_attn_d72_w64p(const bf16_t* q, int64_t q_stride, const bf16_t* q_norm_w, const bf16_t* kp, const bf16_t* vt, bf16_t* out,
                               int64_t out_stride, int batch, int heads, int q_len, int kv_len, int kv_pad, float eps, bool stamp,
                               float k_bound, hipStream_t stream);   // stamp: lab builds only (cycle accounting into the lab debug buffer)
int launch_attn_prep_kv64(const bf16_t* k, int64_t k_stride, const bf16_t* v, int64_t v_stride, const bf16_t* ln_w,
                          const bf16_t* ln_b, const float* rope_cos, const float* rope_sin, int rope_start, int rope_len,
                          bf16_t* kp, bf16_t* vt, int batch, int heads, int kv_len, int kv_pad, float eps, hipStream_t stream);
int launch_flash_attn_d64(const bf16_t* q, int64_t q_stride, const bf16_t* ln_w, const bf16_t* ln_b, const float* rope_cos,
                          const float* rope_sin, int rope_start, int rope_len, const bf16_t* kp, const bf16_t* vt, bf16_t* out,
                          int64_t out_stride, int batch, int heads, int q_len, int kv_len, int kv_pad, float eps, float k_bound,
                          hipStream_t stream);
int launch_ln_modulate(const bf16_t* x, const bf16_t* ln_w, const bf16_t* ln_b, const bf16_t* shift, const bf16_t* scale,
                       bf16_t* y, int64_t rows, int C, int64_t rows_per_sample, int64_t mod_stride, int64_t seg_split,
                       int64_t mod_alt, float eps, hipStream_t stream);
int launch_gate_add_rows(bf16_t* x, const bf16_t* y, const bf16_t* gate, int64_t rows, int C, int64_t rows_per_sample,
                         int64_t gate_stride, int64_t seg_split, int64_t gate_alt, hipStream_t stream);
int launch_unpatchify_cvx(const bf16_t* x, int64_t ldx, float* out, int B, int F, int Hp, int Wp, int Cout, int p,
                          hipStream_t stream);
int launch_im2col_patch(const float* z, int Bz, bf16_t* out, int B, int F, int Cin, int H, int W, int p, hipStream_t stream);
// T <= 32 on the matrix pipe, operands loaded in fragment layout (attention_t3.hip)
int launch_attn_temporal_d72_v3(const bf16_t* qkv, int64_t row_stride, int C, const bf16_t* q_norm_w, const bf16_t* k_norm_w,
                                const float* rope_cos, const float* rope_sin, bf16_t* out, int64_t out_stride, int B, int T, int S,
                                int heads, float eps, float scale, hipStream_t stream, bool ref_rounding = false,
                                bool no_v5 = false);   // no_v5: keep the per-lane-load kernel (A/B id 22)
int launch_attn_temporal_d72(const bf16_t* qkv, int64_t row_stride, int C, const bf16_t* q_norm_w, const bf16_t* k_norm_w,
                             const float* rope_cos, const float* rope_sin, bf16_t* out, int64_t out_stride, int B, int T,
                             int S, int heads, float eps, hipStream_t stream);

}  // namespace vsys
