/* videosys_amd — C ABI of the MI355X (gfx950) STDiT3 denoise-step kernels.
 *
 * The reference (NUS-HPC-AI-Lab/VideoSys @ 2024-12-20) has NO FFI: its hot path is torch ops inside
 * videosys/models/transformers/open_sora_transformer_3d.py and videosys/models/modules/.  This header is the
 * drop-in boundary for that path: every entry point below replaces the torch op sequence at the cited reference
 * lines and is what a reference-side ctypes binding calls (INTEGRATION.md shows the stub).  Plain C: device pointers
 * and sizes only, no torch types.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HBM) unless named host_*; bf16 tensors are raw uint16 bit patterns;
 *  - all tensors row-major, inner (channel) dimension contiguous; "stride"/"ld*" arguments are in ELEMENTS;
 *  - `stream` is a hipStream_t (NULL = the null stream); every call only enqueues work on it — no host sync,
 *    no allocation; the caller owns all buffers;
 *  - return 0 on success, a negative VSYS_ERR_* code otherwise (nothing is enqueued on error);
 *    vsys_strerror() names the code.  There is NO CPU fallback anywhere in the library.
 */
#ifndef VIDEOSYS_AMD_H
#define VIDEOSYS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSYS_ABI_VERSION 1

#define VSYS_ERR_SHAPE  (-1)  /* unsupported size / divisibility */
#define VSYS_ERR_ALIGN  (-2)  /* stride or pointer not 16-byte friendly */
#define VSYS_ERR_ARG    (-3)  /* bad enum / null pointer */
#define VSYS_ERR_LAUNCH (-4)  /* HIP launch error */

/* GEMM epilogues */
#define VSYS_EPI_BIAS      0  /* out = x W^T + b */
#define VSYS_EPI_BIAS_GELU 1  /* out = gelu_tanh(x W^T + b) */
#define VSYS_EPI_GATE_RES  2  /* u = gate[sample] * (x W^T + b); aux = u (optional); out = res + u */

#define VSYS_ACT_NONE      0
#define VSYS_ACT_SILU      1
#define VSYS_ACT_GELU_TANH 2

/* Library identity and error text — no reference counterpart (the reference is pure Python and has no FFI). */
int vsys_abi_version(void);
const char* vsys_strerror(int code);
/* number of HIP devices visible to the library (fails loudly instead of falling back when 0); no reference counterpart */
int vsys_device_count(void);

/* Kernel selection for A/B measurement and for the schedule-equivalence tests — NOT part of the per-call data path: the
 * selection is one process-wide atomic word read once per launch (safe to set from any thread, but it applies to every thread's
 * launches; the product never calls it).  EVERY id this library accepts selects a kernel with VALID output — the GEMM ids differ
 * in schedule / geometry only and are bit-identical to each other; an id the build does not contain returns VSYS_ERR_ARG and
 * leaves the selection unchanged.  (Ablation / cycle-stamp variants whose output is NOT valid, and the measured-but-not-shipped
 * ping-pong / persistent / stream-K GEMMs 60 / 70 / 80, exist only in -DVSYS_LAB builds together with
 * include/videosys_amd_lab.h; the shipped library contains no such code, and no launch path of it allocates or synchronises.)
 * gemm:  2GGPP = tile raster only (see vsys_gemm_raster_probe); 0 = shape dispatch (default); 8 = schedule 8 for every shape (three A slots + two W slots, counted waits); 3 / 6 =
 *        two-stage LDS-DMA schedules; 9 = schedule 8, plain row-major tile order; 20 = 4-wave workgroups, two per CU; 28 =
 *        schedule 8 + producer waves; 30 = 256 x 384 tile; 103 = 128-row tiles.
 * flash: 14 / 15 = the 64-query-rows-per-wave kernel with the hand-allocated tile loop wherever it applies (>= 256 keys and rows) /
 *        never (default: for >= 512 keys);  0 = default (two workgroups per CU; resident K/V for few keys; temporal attention on the matrix pipe for T <= 32); 3 = three workgroups per CU;
 *        4 = VALU temporal kernel (v2) for T <= 40; 8 / 10 = the resident-K/V kernel (all KV tiles of a
 *        (batch, head) staged once per workgroup; default for <= 320 keys and many query rows) whenever the keys fit / never;
 *        9 = online-softmax temporal kernel for every T; 12 = the head-dim-64 kernel on a two-stage K/V ring (shipped: three).
 * No reference counterpart (measurement tooling). */
int vsys_tune_gemm_variant(int variant);
int vsys_tune_flash_variant(int variant);
/* Host-only: where the GEMM kernels' tile raster (column groups of gw tiles inside panel chunks of ph panels inside the 8 per-XCD
 * panel groups; vsys_tune_gemm_variant(20000 + 100 gw + ph) selects it for measurements, 20600 = the default) sends logical tile
 * ``tile`` of an nbm x nbn grid: *bm, *bn.  Lets a CPU test check that every raster is a bijection.  No reference counterpart
 * (measurement tooling). */
int vsys_gemm_raster_probe(int64_t tile, int64_t nbm, int64_t nbn, int64_t gw, int64_t ph, int64_t* bm, int64_t* bn);

/* nn.Linear on token rows with fused epilogue (bf16 in/out, fp32 MFMA accumulate).
 * Replaces: attentions.py:59 (qkv), :107 (proj) + open_sora_transformer_3d.py:219,228 (gate, residual);
 * attentions.py:156,183 + open_sora_transformer_3d.py:237-240 (cross q / proj + residual);
 * timm Mlp fc1+GELU(tanh)/fc2 (open_sora_transformer_3d.py:130-132,267,270,284).
 * Requires N % 192 == 0, K % 64 == 0, strides % 8 == 0; any M.  gate may be NULL (= 1), res/aux may be NULL. */
int vsys_gemm_bf16(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, void* out, int64_t ldo,
                   int64_t M, int64_t N, int64_t K, int epilogue, const void* gate, int64_t gate_sample_stride,
                   int64_t rows_per_sample, const void* res, int64_t ldr, void* aux, int64_t ldaux, void* stream);

/* AdaLN folded into the qkv / fc1 GEMMs: x_m = t2i_modulate(norm(x), shift, scale) followed by the Linear
 * (open_sora_transformer_3d.py:196-197 + attentions.py:59; :260-261 + timm Mlp fc1 :130-132,267) without ever writing x_m:
 *   out[m][n] = rstd_m (sum_k x[m][k] W'[n][k] - mu_m cs[n]) + cv[n]   (+ GELU-tanh for VSYS_EPI_BIAS_GELU)
 * with W' = bf16(W (1 + scale)), cs[n] = sum_k W'[n][k], cv[n] = shift . W[n] + bias[n] (fp32 [N] each, from
 * vsys_adaln_prescale) and the LayerNorm statistics of row m (eps as nn.LayerNorm, :116-117) combined from K / 96 partial
 * (mean, M2) pairs over 96-column blocks: float2 stats[b * stats_ld + m] (written by vsys_gemm_bf16_stats or vsys_ln_row_stats).
 * x is the RAW residual stream [M, K]; K % 96 == 0, K <= 1152; epilogue = VSYS_EPI_BIAS or VSYS_EPI_BIAS_GELU; otherwise as
 * vsys_gemm_bf16. */
int vsys_gemm_bf16_ln(const void* x, int64_t ldx, const void* wp, int64_t ldw, const void* cs, const void* cv, void* out, int64_t ldo,
                      int64_t M, int64_t N, int64_t K, int epilogue, const void* stats, int64_t stats_ld, float eps, void* stream);

/* vsys_gemm_bf16 with VSYS_EPI_GATE_RES (out = res + gate (x W^T + b): attentions.py:107 / :183 / timm Mlp fc2 + the residual adds
 * of open_sora_transformer_3d.py:228,240,284) that ALSO emits the LayerNorm partials of the rows it stores — what the next block's
 * norm1 / norm2 (open_sora_transformer_3d.py:116-117,196,260) would compute from them: float2 stats[b * stats_ld + m] = (mean, M2)
 * of out[m][96 b .. 96 b + 95].  No aux output; N % 192 == 0. */
int vsys_gemm_bf16_stats(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, void* out, int64_t ldo, int64_t M,
                         int64_t N, int64_t K, const void* gate, int64_t gate_sample_stride, int64_t rows_per_sample, const void* res,
                         int64_t ldr, void* stats, int64_t stats_ld, void* stream);

/* vsys_gemm_bf16 with VSYS_EPI_GATE_RES whose store phase also takes the `x = x + cached_output` passes that FOLLOW it in program
 * order when the next sub-blocks of the residual stream are PAB broadcasts (open_sora_transformer_3d.py:186-190,219-225,
 * /root/reference/videosys/core/pab/pab_mgr.py:54-91):  out = bf16(bf16(bf16(res + u) + add1) + add2),  u = bf16(gate (x W^T + b)) —
 * the roundings of the separate torch adds.  res is required; add1 / add2 ([M, N], leading dimension ldr) and aux (the PAB slab
 * copy of u, leading dimension ldr) may be NULL; stats (format of vsys_gemm_bf16_stats, may be NULL) receives the LayerNorm
 * partials of the rows stored. */
int vsys_gemm_bf16_gate_res_add(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, void* out, int64_t ldo,
                                int64_t M, int64_t N, int64_t K, const void* gate, int64_t gate_sample_stride, int64_t rows_per_sample,
                                const void* res, int64_t ldr, void* aux, const void* add1, const void* add2, void* stats,
                                int64_t stats_ld, void* stream);

/* The step-level half of the AdaLN fold: for every site of a step (one qkv or fc1 Linear of one block) W' = bf16(W (1 + scale)),
 * cs, cv as defined at vsys_gemm_bf16_ln, all sites in ONE launch.  shift / scale are the sample-0 rows of the step's modulation
 * table (t2i_modulate operands, open_sora_transformer_3d.py:177-179; valid when every sample of the batch shares the timestep, as
 * the CFG pair of scheduling_rflow_open_sora.py:239-244 does).  ``sites``: DEVICE array of nsites x 10 int64: W, bias, W', cs, cv
 * (addresses), shift_off, scale_off (element offsets into ``mod``), N, K, first block of the site; 4 weight rows per block,
 * nblocks = sum ceil(N / 4). */
int vsys_adaln_prescale(const void* sites, int64_t nsites, int64_t nblocks, const void* mod, void* stream);

/* LayerNorm partials (format of vsys_gemm_bf16_ln) of a [rows, C] bf16 tensor no GEMM epilogue produced: the patch embedding
 * in front of block 0 and the x += cached-output steps of PAB (open_sora_transformer_3d.py:116-117,192-193).  Same accumulation
 * order as the epilogue of vsys_gemm_bf16_stats: the same rows give the same partial bits from either.  C % 96 == 0, C <= 1536. */
int vsys_ln_row_stats(const void* x, int64_t rows, int64_t C, void* stats, int64_t stats_ld, void* stream);

/* Small / odd-shaped nn.Linear (any M, N; K % 8 == 0): act_in is applied to x (SiLU of t_block,
 * open_sora_transformer_3d.py:396-399), act_out to the result (TimestepEmbedder/SizeEmbedder mlp, embeddings.py:114-118;
 * OpenSoraCaptionEmbedder y_proj, embeddings.py:197-203). */
int vsys_linear_small(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, void* out, int64_t ldo,
                      int64_t M, int64_t N, int64_t K, int act_in, int act_out, void* stream);

/* y = LayerNorm(x, eps, no affine) * (1 + scale[b]) + shift[b], b = row / rows_per_sample.
 * Replaces nn.LayerNorm + t2i_modulate (open_sora_transformer_3d.py:47-48,117,129,196-197,260-261). C % 8 == 0, C <= 2048. */
int vsys_adaln_modulate(const void* x, const void* shift, const void* scale, void* y, int64_t rows, int64_t C,
                        int64_t rows_per_sample, int64_t mod_sample_stride, float eps, void* stream);

/* out[blk][b][6*C] = bf16(table[blk][6*C] + t_mlp[b][6*C])  (open_sora_transformer_3d.py:177-179 for all blocks). */
int vsys_mod_table(const void* table, const void* t_mlp, void* out, int64_t nblk, int64_t B, int64_t C6, void* stream);

/* TimestepEmbedder.timestep_embedding (embeddings.py:123-141): t fp32 [B] -> out bf16 [B, dim] = [cos | sin]. */
int vsys_timestep_embedding(const void* t_f32, void* out, int64_t B, int64_t dim, void* stream);

/* OpenSoraPatchEmbed3D + "x + pos_emb" (embeddings.py:85-104; open_sora_transformer_3d.py:593-595).
 * z fp32 [Bz, Cin, T, H, W]; sample b reads z[b % Bz] (CFG duplicate, scheduling_rflow_open_sora.py:239);
 * w bf16 [C, Cin*ph*pw]; pos bf16 [Hp*Wp, C]; out bf16 [B, T, Hp*Wp, C]. */
int vsys_patch_embed(const void* z_f32, int64_t Bz, const void* w, const void* bias, const void* pos, void* out, int64_t B,
                     int64_t Cin, int64_t T, int64_t H, int64_t W, int64_t ph, int64_t pw, int64_t C, void* stream);

/* T2IFinalLayer + unpatchify + .to(float32) (open_sora_transformer_3d.py:75-87,622-630,634-658).
 * x bf16 [B, T*Hp*Wp, C]; table bf16 [2, C]; tvec bf16 [B, C]; w bf16 [ph*pw*Cout, C]; out fp32 [B, Cout, T, H, W]. */
int vsys_final_layer(const void* x, const void* table, const void* tvec, const void* w, const void* bias, void* out_f32,
                     int64_t B, int64_t T, int64_t Hp, int64_t Wp, int64_t H, int64_t W, int64_t ph, int64_t pw,
                     int64_t Cout, int64_t C, float eps, void* stream);

/* The same two layers on the S-shard of a sequence-parallel rank (DSP at rest: rank r holds tokens r*Sl .. r*Sl+Sl-1 of every
 * (b, t), open_sora_transformer_3d.py:598-603,615-619): the patch embedding is computed for the local tokens only (tokens past
 * Hp*Wp are the zero rows split_sequence pads with), the final layer runs on the local rows and leaves its ph*pw*Cout values per
 * token in tokens_f32 [B, T, Sl, n_out]; after an all-gather of those (0.6 MB per rank at config 2 instead of the 11 MB of hidden
 * state) vsys_unpatchify_tokens scatters [P, B, T, Sl, n_out] into out_f32 [B, Cout, T, H, W].  Same arithmetic per token as the
 * whole-frame entry points. */
int vsys_patch_embed_shard(const void* z_f32, int64_t Bz, const void* w, const void* bias, const void* pos, void* out, int64_t B,
                           int64_t Cin, int64_t T, int64_t H, int64_t W, int64_t ph, int64_t pw, int64_t C, int64_t s0, int64_t Sl,
                           void* stream);
int vsys_final_layer_tokens(const void* x, const void* table, const void* tvec, const void* w, const void* bias, void* tokens_f32,
                            int64_t B, int64_t T, int64_t Sl, int64_t n_out, int64_t C, float eps, void* stream);
int vsys_unpatchify_tokens(const void* tokens_f32, void* out_f32, int64_t P, int64_t B, int64_t T, int64_t Sl, int64_t Hp,
                           int64_t Wp, int64_t H, int64_t W, int64_t ph, int64_t pw, int64_t Cout, void* stream);

/* RFLOW CFG combine + Euler update (scheduling_rflow_open_sora.py:245-252): z fp32 [Bz, Cin, thw] +=
 * (uncond + g (cond - uncond)) * dt with cond = model_out[b, :Cin], uncond = model_out[b + Bz, :Cin]. */
int vsys_cfg_euler_step(void* z_f32, const void* model_out_f32, int64_t Bz, int64_t Cin, int64_t Cout, int64_t thw,
                        float guidance, float dt, void* stream);

/* CFG combine + one linear scheduler update, z fp32 [Bz, Cin, thw] = c_z z + c_eps (uncond + g (cond - uncond)) on the
 * first Cin of Cout predicted channels (learned-sigma half dropped).  DDIM eta = 0
 * (diffusers DDIMScheduler.step as called from pipelines/latte/pipeline_latte.py:864-876 and
 * schedulers/scheduling_ddim_cogvideox.py:299-393 with v-prediction folded on the host):
 * c_z = sqrt(a_prev / a_t), c_eps = sqrt(1 - a_prev) - sqrt(a_prev (1 - a_t) / a_t).
 * cond_first != 0: model_out[b] is the conditional half (RFLOW order); 0: model_out[b + Bz] is (negative prompt first). */
int vsys_cfg_linear_step(void* z_f32, const void* model_out_f32, int64_t Bz, int64_t Cin, int64_t Cout, int64_t thw,
                         float guidance, float c_z, float c_eps, int cond_first, void* stream);

/* x[r, :] += e[(r / group) % period, :] over rows of C bf16 — Latte's temporal position embedding added to the
 * (b, f, s)-ordered hidden states before the first temporal block (latte_transformer_3d.py:1410-1411; group = S,
 * period = F). */
int vsys_add_bcast_rows(void* x, const void* e, int64_t rows, int64_t C, int64_t group, int64_t period, void* stream);

/* x += y over n bf16 elements (PAB broadcast step: x + last_attn / last_cross, open_sora_transformer_3d.py:192-193,228,234-235). */
int vsys_add_rows(void* x, const void* y, int64_t n, void* stream);

/* 4-D strided row copy with zero fill outside (n1_valid, n2_valid): DSP all-to-all pack/unpack
 * (core/distributed/comm.py:104-108,282-304; open_sora_transformer_3d.py:288-315). */
int vsys_copy_4d(const void* src, void* dst, int64_t n0, int64_t n1, int64_t n2, int64_t C, int64_t ss0, int64_t ss1,
                 int64_t ss2, int64_t ds0, int64_t ds1, int64_t ds2, int64_t n1_valid, int64_t n2_valid, void* stream);

/* K RMS-norm (normalization.py:28-33; k_norm_w NULL = none) + head-major K and transposed V for vsys_flash_attn_d72.
 * k(b,s,h) at k + (b*kv_len + s)*k_stride + h*72, same for v.  kp [batch, H, kv_pad, 72]; vt [batch, H, 96, kv_pad]
 * (rows 72..95 must be zero-initialised by the caller once).  kv_pad % 64 == 0.
 * The pair (kp, vt) is an opaque operand of vsys_flash_attn_d72: kp carries the softmax scale 72^-1/2 * log2(e)
 * (folded in before its single bf16 rounding) and vt rows 72 and 76 are written as ones over the valid keys, so the
 * flash kernel gets exp2-ready logits and the softmax denominator from the matrix pipe. */
int vsys_attn_prep_kv(const void* k, int64_t k_stride, const void* v, int64_t v_stride, const void* k_norm_w, void* kp,
                      void* vt, int64_t batch, int64_t heads, int64_t kv_len, int64_t kv_pad, float eps, void* stream);

/* softmax(q k^T / sqrt(72)) v for head_dim 72, non-causal, keys >= kv_len masked; optional q RMS-norm.
 * Spatial self-attention (attentions.py:75,100) and cross-attention (attentions.py:259-270).
 * q(b,s,h) at q + (b*q_len + s)*q_stride + h*72; out likewise with out_stride. */
int vsys_flash_attn_d72(const void* q, int64_t q_stride, const void* q_norm_w, const void* kp, const void* vt, void* out,
                        int64_t out_stride, int64_t batch, int64_t heads, int64_t q_len, int64_t kv_len, int64_t kv_pad,
                        float eps, void* stream);

/* vsys_flash_attn_d72 for the cross attention against hoisted text K / V (attentions.py:259-270; kv_linear of the packed prompt
 * :157, :240-258) with a promise about the PADDING: (kp, vt) were written by vsys_attn_prep_kv for exactly this kv_len into
 * zero-initialised buffers, so every Kp row and Vt column behind kv_len is zero — including the ones rows of Vt.  A padding key then
 * has logit 0 and weight 0 in numerator and denominator, and the kernels that take the promise (the resident-K/V cross-attention
 * kernel) drop the mask of the ragged last tile.  Same results as vsys_flash_attn_d72 to the last bit of the tolerance (P of the last
 * tile may be rounded at another scale when a padding key raises the running max); a row whose real logits all lie ~100 (exp2
 * domain) below zero is detected and recomputed with the mask.  NOT valid when kv_len is SHORTER than what the buffers were
 * prepared for (Latte's per-sample text lengths inside one buffer: call vsys_flash_attn_d72).  kv_len % 64 == 0: identical to
 * vsys_flash_attn_d72. */
int vsys_flash_attn_d72_exact(const void* q, int64_t q_stride, const void* q_norm_w, const void* kp, const void* vt, void* out,
                              int64_t out_stride, int64_t batch, int64_t heads, int64_t q_len, int64_t kv_len, int64_t kv_pad,
                              float eps, void* stream);

/* vsys_flash_attn_d72 (attentions.py:75,100) with a promise about the keys: k_norm_bound >= the Euclidean norm of every Kp row (as stored: normed, scaled
 * by log2(e) / sqrt(72)).  For an RMS-normed key that is sqrt(72) max|k_norm.weight| log2(e) / sqrt(72) (1 + rounding), a property
 * of the WEIGHTS (normalization.py:28-33): the caller computes it once per block.  By Cauchy-Schwarz m_i = |q_i| k_norm_bound bounds
 * every logit of query row i, and softmax is invariant under the choice of the subtracted m: the kernels that take the promise
 * (64 rows per wave, attention_w64.hip) use exp2(s - m_i) with no running maximum and no rescale of the accumulator.  The caller
 * must also guarantee |q_i| k_norm_bound <= 60 for every query row (RMS-normed q: sqrt(72) max|q_norm.weight| k_norm_bound):
 * 2 m < 126 keeps every row sum representable.  k_norm_bound = 0: exactly vsys_flash_attn_d72.  q_norm_w == NULL: the bound is
 * ignored.  The BITS of vsys_flash_attn_d72 are not promised (P is rounded to bf16 at another scale): the same tolerance is. */
int vsys_flash_attn_d72_kb(const void* q, int64_t q_stride, const void* q_norm_w, const void* kp, const void* vt, void* out,
                           int64_t out_stride, int64_t batch, int64_t heads, int64_t q_len, int64_t kv_len, int64_t kv_pad,
                           float eps, float k_norm_bound, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * CogVideoX: joint [text | video] blocks, head_dim 64.
 * ------------------------------------------------------------------------------------------------------------------ */

/* vsys_gemm_bf16 with VSYS_EPI_GATE_RES and TWO gate vectors per sample: rows whose position inside the sample is
 * < seg_split (the text tokens) read the gate gate_alt elements after the sample's gate pointer
 * (CogVideoXBlock.forward: hidden + gate_msa * attn / encoder_hidden + enc_gate_msa * attn,
 * cogvideox_transformer_3d.py:288-289,300-301; gates from CogVideoXLayerNormZero, modules/normalization.py:52-59). */
int vsys_gemm_bf16_gate2(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, void* out, int64_t ldo,
                         int64_t M, int64_t N, int64_t K, const void* gate, int64_t gate_sample_stride,
                         int64_t rows_per_sample, int64_t seg_split, int64_t gate_alt, const void* res, int64_t ldr,
                         void* aux, int64_t ldaux, void* stream);

/* y = (LayerNorm(x) * ln_w + ln_b) * (1 + scale) + shift over rows of C bf16 (C <= 3072).  ln_w/ln_b NULL: no affine;
 * shift/scale NULL: no modulation (nn.LayerNorm norm_final, cogvideox_transformer_3d.py:566-572).  shift/scale point at
 * sample 0's vectors (sample stride mod_sample_stride); rows at position < seg_split inside their sample read them
 * mod_alt elements further on (CogVideoXLayerNormZero, modules/normalization.py:36-60; AdaLayerNorm chunk_dim=1 :62-114). */
int vsys_ln_modulate(const void* x, const void* ln_w, const void* ln_b, const void* shift, const void* scale, void* y,
                     int64_t rows, int64_t C, int64_t rows_per_sample, int64_t mod_sample_stride, int64_t seg_split,
                     int64_t mod_alt, float eps, void* stream);

/* x[r] += gate[sample(r)] * y[r] with the two-segment gate addressing above: the PAB broadcast step of CogVideoXBlock
 * re-gates the cached, un-gated attention output (cogvideox_transformer_3d.py:276-289). */
int vsys_gate_add_rows(void* x, const void* y, const void* gate, int64_t rows, int64_t C, int64_t rows_per_sample,
                       int64_t gate_sample_stride, int64_t seg_split, int64_t gate_alt, void* stream);

/* CogVideoXPatchEmbed's Conv2d(k = s = p) operand (modules/embeddings.py:14-51): z fp32 [Bz, F, Cin, H, W] (sample b reads
 * z[b % Bz]) -> out bf16 [B*F*(H/p)*(W/p), Cin*p*p] with columns ordered (c, dy, dx) like the flattened conv weight. */
int vsys_im2col_patch(const void* z_f32, int64_t Bz, void* out, int64_t B, int64_t F, int64_t Cin, int64_t H, int64_t W,
                      int64_t p, void* stream);

/* CogVideoX unpatchify (cogvideox_transformer_3d.py:581-583): x bf16 [(b, f, hp, wp)] rows of ldx >= Cout*p*p elements,
 * columns ordered (c, dy, dx) -> out fp32 [B, F, Cout, Hp*p, Wp*p]. */
int vsys_unpatchify_cvx(const void* x, int64_t ldx, void* out_f32, int64_t B, int64_t F, int64_t Hp, int64_t Wp, int64_t Cout,
                        int64_t p, void* stream);

/* K side of CogVideoXAttnProcessor2_0 (cogvideox_transformer_3d.py:108,127-149): LayerNorm qk-norm (ln_w/ln_b [64] or NULL),
 * rotary embedding on tokens rope_start <= s < rope_start + rope_len (cos/sin fp32 [rope_len, 64], NULL = none), softmax
 * scale folded in; head-major K (16-byte chunks of a row stored at chunk ^ ((row>>1)&7): an operand private to
 * vsys_flash_attn_d64) and transposed V.  kp [batch, H, kv_pad, 64], vt [batch, H, 64, kv_pad], kv_pad % 64 == 0. */
int vsys_attn_prep_kv64(const void* k, int64_t k_stride, const void* v, int64_t v_stride, const void* ln_w, const void* ln_b,
                        const void* rope_cos_f32, const void* rope_sin_f32, int64_t rope_start, int64_t rope_len, void* kp,
                        void* vt, int64_t batch, int64_t heads, int64_t kv_len, int64_t kv_pad, float eps, void* stream);

/* softmax(q k^T / 8) v for head_dim 64, non-causal (F.scaled_dot_product_attention in CogVideoXAttnProcessor2_0,
 * cogvideox_transformer_3d.py:151-158), with the q side of the processor above (LayerNorm + RoPE, :108-149) applied on
 * the fly.  q(b,s,h) at q + (b*q_len + s)*q_stride + h*64; out likewise. */
int vsys_flash_attn_d64(const void* q, int64_t q_stride, const void* ln_w, const void* ln_b, const void* rope_cos_f32,
                        const void* rope_sin_f32, int64_t rope_start, int64_t rope_len, const void* kp, const void* vt, void* out,
                        int64_t out_stride, int64_t batch, int64_t heads, int64_t q_len, int64_t kv_len, int64_t kv_pad, float eps,
                        void* stream);

/* vsys_flash_attn_d64 (cogvideox_transformer_3d.py:108-158) with the promise of vsys_flash_attn_d72_kb about the Kp rows (LayerNorm qk-norm with affine weights, rotary
 * embedding, scale log2(e) / 8: |k| <= (8 max|w| + |b|_2) log2(e) / 8, a property of the block's norm_k weights): no running
 * maximum in the kernels that take it (attention64_w64.hip).  |q_i| k_norm_bound <= 60 for every query row is the caller's to
 * guarantee (|q_i| <= 8 max|norm_q.weight| + |norm_q.bias|_2).  0 = exactly vsys_flash_attn_d64. */
int vsys_flash_attn_d64_kb(const void* q, int64_t q_stride, const void* ln_w, const void* ln_b, const void* rope_cos_f32,
                        const void* rope_sin_f32, int64_t rope_start, int64_t rope_len, const void* kp, const void* vt, void* out,
                        int64_t out_stride, int64_t batch, int64_t heads, int64_t q_len, int64_t kv_len, int64_t kv_pad, float eps, float k_norm_bound,
                        void* stream);

/* Temporal self-attention over the T frames of every (b, s) token: RMS qk-norm, RoPE (cos/sin fp32 [T, 72], NULL =
 * none), fp32 softmax (attentions.py:75-78,95-97,111-120; open_sora_transformer_3d.py:203-206).
 * q_norm_w == k_norm_w == NULL: no qk-norm (Latte temporal blocks, latte_transformer_3d.py:680-760).
 * qkv bf16 rows ordered (b, t, s), row_stride elements per row, q|k|v at column offsets 0|C|2C, head h at h*72. */
int vsys_attn_temporal_d72(const void* qkv, int64_t row_stride, int64_t C, const void* q_norm_w, const void* k_norm_w,
                           const void* rope_cos_f32, const void* rope_sin_f32, void* out, int64_t out_stride, int64_t B,
                           int64_t T, int64_t S, int64_t heads, float eps, void* stream);

/* Up to 16 vsys_copy_4d problems over one (src, dst) pair in a single launch: the per-peer pack (or unpack) pieces of a DSP /
 * Ulysses all-to-all (comm.py:104-108,282-304; cogvideox_transformer_3d.py:45-86).  desc (HOST) holds nops x 14 int64:
 * src_off, dst_off, n0, n1, n2, run, ss0, ss1, ss2, ds0, ds1, ds2, n1_valid, n2_valid (elements; run % 8 == 0). */
int vsys_copy_4d_batch(const void* src, void* dst, int64_t nops, const int64_t* desc, void* stream);

/* The whole layout switch of Dynamic Sequence Parallelism in ONE launch per rank, peer to peer over xGMI (comm.py:104-141 _all_to_all_func
 * and :282-304 all_to_all_with_pad around open_sora_transformer_3d.py:288-315 dynamic_switch; replaces pack + all_to_all_single + unpack).
 * Problem i of the batch copies rows of ``src`` straight into the DESTINATION tensor of peer i in its final layout: desc (HOST) holds
 * nops x 17 int64 — the 14 of vsys_copy_4d_batch (dst_off relative to the problem's own destination), then the destination base
 * address (this process's mapping of the peer's tensor: vsys_p2p_ipc_open, or a plain device pointer inside one process), the
 * address of flag[self_index] in that peer's flag array (0 for the rank's own problem) and 1 when ANOTHER process or device reads
 * that destination (its rows then leave as system-scope write-through stores; 0: a reader ordered by launches on this device).  After the write-through stores of EVERY problem of the
 * launch have been acknowledged (drained per wave, counted per workgroup) the launch's last workgroup stores the exchange's sequence number into each
 * peer's flag (a relaxed system-scope store: the ordering comes from the drained write-through payload, there is no release fence); the launch ends when every flag q != self_index of
 * ``my_flags`` (n_flags x uint32, fine-grained memory: vsys_p2p_alloc) has reached that number, i.e. when this rank's own destination
 * is complete: the consumer is simply the next launch on ``stream``.  ``state``: 19 x 32 uint32 (one 128-byte line per word) of zeroed
 * device memory private to this exchange site — word 0 the sequence number (advanced by the kernel itself, so a recorded launch
 * program replays unchanged), word 18 (uint32 index 576) an error word (0; 1 + q when peer q's flag did not arrive within ``timeout_ticks`` of the 100 MHz wall clock; sticky: a site that timed
 * out once no longer waits; timeout_ticks 0 = wait for ever, < 0 = do not wait at all: the caller orders the peers' launches itself).
 * Every rank of the group must issue the same sequence of exchanges per site. */
int vsys_p2p_exchange(const void* src, int64_t nops, const int64_t* desc, const void* my_flags, int64_t n_flags, int64_t self_index,
                      void* state, int64_t timeout_ticks, void* stream);
/* Set-up of vsys_p2p_exchange (host side, synchronous, never inside a step; no reference counterpart: the reference's exchange is
 * torch.distributed over NCCL, comm.py:104-141).  Zeroed device memory — fine-grained (flags: polled inside a running kernel while a
 * peer stores to them) or ordinary. */
int vsys_p2p_alloc(int64_t bytes, int64_t fine_grained, void** ptr);
/* Release of vsys_p2p_alloc memory (no reference counterpart). */
int vsys_p2p_free(void* ptr);
/* The 64-byte HIP IPC handle of a vsys_p2p_alloc allocation, for the other rank processes (no reference counterpart). */
int vsys_p2p_ipc_export(const void* ptr, void* handle64);
/* Mapping of a peer process's handle into this process: a device pointer valid here (no reference counterpart). */
int vsys_p2p_ipc_open(const void* handle64, void** ptr);
/* Unmapping of vsys_p2p_ipc_open (no reference counterpart). */
int vsys_p2p_ipc_close(void* ptr);

/* ---- T5 text encoder (T5EncoderModel of transformers, third-party; called once per prompt at pipeline_open_sora.py:269-287,
 * pipeline_cogvideox.py:211-247, pipeline_latte.py) — the linears are vsys_conv_bf16 with one tap. */
/* out[i, :] = table[ids[i], :] (nn.Embedding = T5Stack.embed_tokens, transformers, third-party; ids int64 on the device, clamped
 * to [0, vocab)). */
int vsys_gather_rows(const void* table, const void* ids_i64, void* out, int64_t n, int64_t C, int64_t vocab, void* stream);
/* T5LayerNorm (transformers modeling_t5.py, third-party): y = w * bf16(x * rsqrt(mean(x^2) + eps)), fp32 statistics, C <= 8192. */
int vsys_rms_norm_rows(const void* x, const void* w, void* y, int64_t rows, int64_t C, float eps, void* stream);
/* T5DenseGatedActDense (transformers modeling_t5.py, third-party): out[r, f] = bf16(gelu_new(h[r, f])) * h[r, F + f],
 * h = [wi_0 x | wi_1 x] of 2F columns. */
int vsys_geglu(const void* h, void* out, int64_t rows, int64_t F, void* stream);
/* Finish of a split-K, transposed skinny linear (few activation rows against a large weight — T5 at 300 tokens: every weight byte
 * is used for 300 MACs, so the layer is a weight stream): the GEMM ran as vsys_conv_bf16 with the WEIGHT as the row operand, the
 * activations as the 128-column operand and S slices of K in the batch dimension, leaving part_f32[s][n][m] (row pitch ldp >= M,
 * slice pitch slab elements); out[m][n] = bf16(sum_s part[s][n][m]) (+ res[m][n], added after the rounding).
 * Replaces the tail of nn.Linear inside transformers' T5Attention / T5DenseGatedActDense (third-party; call sites
 * pipeline_open_sora.py:269-287). */
int vsys_splitk_reduce_t(const void* part_f32, int64_t nsplit, int64_t slab, int64_t ldp, const void* res, int64_t ldr, void* out, int64_t ldo,
                         int64_t M, int64_t N, void* stream);
/* T5Attention (transformers modeling_t5.py, third-party; encoder self-attention, d_kv = 64, no score scaling):
 * softmax_j(q_i k_j + relbias[h][j - i + L - 1]) v over the
 * first klen[b] keys.  qkv rows (b, l) of row_stride elements, q | k | v at column 0 | inner | 2 inner, head h at h*64;
 * relbias fp32 [heads, 2L-1]; klen int32 [B] on the device; L <= 512. */
int vsys_t5_attention(const void* qkv, int64_t row_stride, int64_t inner, const void* relbias_f32, const void* klen_i32, void* out,
                      int64_t out_stride, int64_t B, int64_t L, int64_t heads, void* stream);
/* The weight-streaming linear on the 256 x 384 tile (one 8-wave workgroup per CU): part_f32[s][m][n] = sum over K slice s of
 * x[m][k] w[n][k] for m < rows (x holds rows_padded rows, a multiple of 384; the rows past the real ones are multiplied but not
 * stored), n < N, nsplit slices of ceil(K / 32 / nsplit) * 32 columns (the last one takes what is left; K % 32 == 0).  Every 256-row weight panel is read by exactly one workgroup.  vsys_splitk_reduce then
 * gives out[m][n] = bf16(sum_s part[s][m][n]) (+ res[m][n] after the rounding), slab = rows_padded * N, ldp = N.
 * Replaces nn.Linear inside transformers' T5 stack at a few hundred rows (third-party; call sites pipeline_open_sora.py:269-287). */
int vsys_gemm_skinny_slices(const void* w, int64_t ldw, const void* x, int64_t ldx, void* part_f32, int64_t rows, int64_t rows_padded,
                            int64_t N, int64_t K, int64_t nsplit, void* stream);
int vsys_splitk_reduce(const void* part_f32, int64_t nsplit, int64_t slab, int64_t ldp, const void* res, int64_t ldr, void* out, int64_t ldo,
                       int64_t M, int64_t N, void* stream);
/* The same attention (T5Attention, transformers modeling_t5.py, third-party; call sites pipeline_open_sora.py:269-287) on the matrix
 * pipe, ONE sample per call (its own key length kv_len <= L): the head_dim-64 flash kernel with
 * an additive (head, key - query) logit bias.  qkv bf16 rows [L, 3*inner] (q | k | v, head h at h*64); bias_f32 [heads, bias_ld]
 * holds log2(e) * T5Attention.compute_bias, entry of (h, key - query) at bias_center + key - query, padded by the caller so that
 * every key < 64*ceil(L/64) and query < 128*ceil(L/128) stays inside (bias_center >= 128*ceil(L/128) - 1, bias_ld >= bias_center +
 * 64*ceil(L/64)); kp / vt: workspaces of heads * 64*ceil(L/64) * 64 bf16 each (the K / V^T layouts of vsys_attn_prep_kv64). */
int vsys_t5_attention_mfma(const void* qkv, int64_t row_stride, int64_t inner, const void* bias_f32, int64_t bias_ld, int64_t bias_center,
                           int64_t kv_len, void* kp, void* vt, void* out, int64_t out_stride, int64_t L, int64_t heads, void* stream);

/* ---- VAE decode (VideoAutoencoderPipeline.decode, autoencoder_kl_open_sora.py:672-695) --------------
 * Activations are channels-last bf16 row matrices over a grid; a grid is described by int64 g[6] = {T, H, W, pad, tf,
 * sample_rows}: sample n, frame t, pixel (h, w) is row n*sample_rows + ((t + tf)*(H + 2 pad) + h + pad)*(W + 2 pad) + w + pad.
 * Borders (pad = 1) and front frames (tf) of conv INPUT buffers are zero and never written by these entry points. */

/* Implicit-GEMM convolution (replaces nn.Conv3d inside CausalConv3d, autoencoder_kl_open_sora.py:89-124, and the Conv2d /
 * Linear layers of the diffusers 0.30.0 AutoencoderKL decoder): out[r, n] = bias[n] + sum_{tap, c} a[r + delta(tap), c] *
 * w[n, tap*cin + c] (+ res[r, n] after rounding to bf16), tap = (a, b, c) in kt x kh x kw, delta = a*plane_pitch +
 * b*row_pitch + c rows.  `a` points at the row tap (0,0,0) reads for output row 0; rows up to M - 1 + delta(last tap) must
 * be readable.  kt*kh*kw == 1 is a plain GEMM with any K = cin (multiple of 32); otherwise cin/32 must be a power of two.
 * N % 128 == 0.  Exactly one of out (bf16) / out_f32 (fp32, scaled by out_scale, no bias/res) is non-NULL.  batch > 1
 * repeats the product with operand strides batch_a / batch_w / batch_o (elements). */
int vsys_conv_bf16(const void* a, int64_t lda, const void* w, int64_t ldw, const void* bias, const void* res, int64_t ldr,
                   void* out, void* out_f32, int64_t ldo, int64_t M, int64_t N, int64_t cin, int64_t kt, int64_t kh, int64_t kw,
                   int64_t row_pitch, int64_t plane_pitch, int64_t batch, int64_t batch_a, int64_t batch_w, int64_t batch_o,
                   float out_scale, void* stream);

/* nn.GroupNorm statistics over the interior of every sample (autoencoder_kl_open_sora.py:145,147,343; diffusers ResnetBlock2D
 * norm1/norm2): stats_f32[n][group] = (mean, 1/sqrt(var + eps)).  partial_f32 is scratch of N*nblk*(C/4)*2 floats. */
int vsys_gn_stats(const void* x, const int64_t* grid, int64_t N, int64_t C, int64_t groups, float eps, void* partial_f32,
                  int64_t nblk, void* stats_f32, void* stream);
/* y = act(bf16((x - mean) * rstd * gamma + beta)), act = VSYS_ACT_NONE | VSYS_ACT_SILU, interior rows of grid_dst only: the
 * GroupNorm (+ SiLU) of ResBlock.forward (autoencoder_kl_open_sora.py:152-164) / diffusers ResnetBlock2D with vsys_gn_stats. */
int vsys_gn_apply(const void* x, const int64_t* grid_src, void* y, const int64_t* grid_dst, int64_t N, int64_t C, int64_t groups,
                  const void* stats_f32, const void* gamma, const void* beta, int act, void* stream);
/* grid-to-grid copy of the interior; up = 1: nearest-neighbour 2x upsampling in H and W (diffusers Upsample2D before its conv).
 * tmode 0: same frame count; 1: destination frame t reads source frame t/2; 2: CogVideoXUpsample3D with an odd frame count
 * (modules/upsampling.py:42-49): 2T-1 frames, frame 0 stays single. */
int vsys_regrid(const void* x, const int64_t* grid_src, void* y, const int64_t* grid_dst, int64_t N, int64_t C, int64_t up,
                int64_t tmode, void* stream);
/* Strided pick, destination (t, h, w) = source (t * t_stride + t_first, h * s_stride + s_first, w * s_stride + s_first): the
 * strided convolutions of the VAE ENCODERS are their stride-1 convolution (vsys_conv_bf16) sampled — diffusers Downsample2D
 * (pad (0, 1, 0, 1) + 3 x 3 stride 2; third-party, called from autoencoder_kl_open_sora.py:503-520) at (2i + 1, 2j + 1) of the
 * pad-1 conv, the stride-(2, 1, 1) CausalConv3d of the temporal encoder (autoencoder_kl_open_sora.py:107-125,229-236) at frame
 * 2t + 1 of the two-front-frames conv. */
int vsys_subsample(const void* x, const int64_t* grid_src, void* y, const int64_t* grid_dst, int64_t N, int64_t C, int64_t t_stride,
                   int64_t s_stride, int64_t t_first, int64_t s_first, void* stream);
/* CogVideoXSpatialNorm3D + SiLU (autoencoder_kl_cogvideox.py:165-178,275-276): y = silu(GN(x) * Y + B); yb = [conv_y(zq) |
 * conv_b(zq)] computed at LATENT resolution, rows (n, zt, zh, zw) of 2C columns; the latent voxel of (t, h, w) follows
 * F.interpolate(zq, size=f.shape) including the first-frame split for odd frame counts.  stats from vsys_gn_stats. */
int vsys_spatial_norm_apply(const void* x, const int64_t* grid_src, void* y, const int64_t* grid_dst, int64_t N, int64_t C,
                            int64_t groups, const void* stats_f32, const void* gamma, const void* beta, const void* yb, int64_t zT,
                            int64_t zH, int64_t zW, void* stream);
/* tiled_decode's blend_v (axis 0) / blend_h (axis 1) (autoencoder_kl_cogvideox.py:1145-1159) on planar bf16 tiles
 * a [outer, Ha, Wa], b [outer, Hb, Wb], in place on b: the first ext rows (columns) of b fade in from the last ext of a. */
int vsys_blend_edge(const void* a, void* b, int64_t outer, int64_t Ha, int64_t Wa, int64_t Hb, int64_t Wb, int64_t ext, int64_t axis,
                    void* stream);
/* temporal depth-to-space "B (C ts) T H W -> B C (T ts) H W", ts = 2 (autoencoder_kl_open_sora.py:362-368): x has 2*Cout channels. */
int vsys_d2s_time(const void* x, const int64_t* grid_src, void* y, const int64_t* grid_dst, int64_t N, int64_t Cout, void* stream);
/* First decoder layer (VideoAutoencoderPipeline.decode z * scale + shift, autoencoder_kl_open_sora.py:676-677; post_quant_conv
 * + Decoder.conv1 :459,355-357; also the encoders' first convolutions with identity parameters): planar bf16 latent
 * z[4][F][H][W] -> z*scale + shift -> 1x1 post_quant_conv -> im2col rows
 * [F*H*W, kcols] (column = tap*4 + channel, zero padding; kt = 3: causal, two virtual zero frames in front).
 * params (HOST floats): scale[4], shift[4], pq_w[4][4], pq_b[4]. */
int vsys_vae_first_im2col(const void* z, int64_t F, int64_t H, int64_t W, int64_t kt, int64_t kcols, const float* params,
                          void* out, void* stream);
/* Last decoder layer: the first nc (<= 4) channels of rows [*, ldx] -> planar bf16 out[c][f0 + frame - tskip][h][w], frames
 * below tskip dropped (VAE_Temporal.decode's x[:, :, time_padding:], autoencoder_kl_open_sora.py:461). */
int vsys_extract_planar(const void* x, const int64_t* grid, int64_t N, int64_t ldx, int64_t nc, int64_t tskip, void* out,
                        int64_t Ftot, int64_t f0, void* stream);
/* row softmax over the first n of ld columns, fp32 [rows, ld] -> bf16 [rows, ld] with zeros in columns n..ld-1 (mid-block
 * attention of the 2-D VAE — diffusers Attention under VideoAutoencoderKL.decode / .encode, autoencoder_kl_open_sora.py:503-538 —
 * keys padded to the 128-column tile; n % 4 == 0, ld % 4 == 0, ld <= 8192). */
int vsys_softmax_rows(const void* s_f32, void* p, int64_t rows, int64_t n, int64_t ld, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------------
 * Launch programs.  A denoise step is ~450 launches of the entry points above with arguments that do not change from step to
 * step (workspaces, weights and PAB slabs are resident; per-step values live in device buffers).  The host mirror records the
 * sequence once per (geometry, PAB decision pattern, parallel layout) and replays it with ONE call: the loop below runs in C,
 * so the per-launch host cost is hipLaunchKernel alone — what matters when P-way sequence parallelism shrinks a rank's device
 * time per step P-fold while the launch count stays (the reference issues every op from Python every step,
 * open_sora_transformer_3d.py:608-613).  A command = an entry point (VSYS_OP_*), its integer / pointer arguments in declaration
 * order in a[], its float arguments in declaration order in f[], and the index of its stream in the table handed to
 * vsys_program_run.  Commands run in order; the first failing command stops the run, its index goes to *failed_at and its error
 * code is returned (nothing after it is enqueued).  No allocation, no synchronisation.  Host-side actions between launches
 * (collectives, event record / wait across streams) stay with the caller, which replays the program in segments. */
/* >>> VSYS_OP codes (generated: csrc/gen/program_gen.py) */
#define VSYS_OP_GEMM_BF16                    1
#define VSYS_OP_LINEAR_SMALL                 2
#define VSYS_OP_ADALN_MODULATE               3
#define VSYS_OP_MOD_TABLE                    4
#define VSYS_OP_TIMESTEP_EMBEDDING           5
#define VSYS_OP_PATCH_EMBED                  6
#define VSYS_OP_FINAL_LAYER                  7
#define VSYS_OP_CFG_EULER_STEP               8
#define VSYS_OP_ADD_ROWS                     9
#define VSYS_OP_COPY_4D_BATCH                10
#define VSYS_OP_ATTN_PREP_KV                 11
#define VSYS_OP_FLASH_ATTN_D72               12
#define VSYS_OP_ATTN_TEMPORAL_D72            13
#define VSYS_OP_ADD_BCAST_ROWS               14
#define VSYS_OP_GEMM_BF16_GATE2              15
#define VSYS_OP_LN_MODULATE                  16
#define VSYS_OP_GATE_ADD_ROWS                17
#define VSYS_OP_ATTN_PREP_KV64               18
#define VSYS_OP_FLASH_ATTN_D64               19
#define VSYS_OP_PATCH_EMBED_SHARD            20
#define VSYS_OP_FINAL_LAYER_TOKENS           21
#define VSYS_OP_UNPATCHIFY_TOKENS            22
#define VSYS_OP_GEMM_BF16_LN                 23
#define VSYS_OP_GEMM_BF16_STATS              24
#define VSYS_OP_ADALN_PRESCALE               25
#define VSYS_OP_LN_ROW_STATS                 26
#define VSYS_OP_GEMM_BF16_GATE_RES_ADD       27
#define VSYS_OP_FLASH_ATTN_D72_KB            28
#define VSYS_OP_FLASH_ATTN_D64_KB            29
#define VSYS_OP_FLASH_ATTN_D72_EXACT         30
#define VSYS_OP_P2P_EXCHANGE                 31
#define VSYS_OP_CFG_LINEAR_STEP              32
#define VSYS_OP_COPY_4D                      33
#define VSYS_OP_IM2COL_PATCH                 34
#define VSYS_OP_UNPATCHIFY_CVX               35
#define VSYS_OP_GATHER_ROWS                  36
#define VSYS_OP_RMS_NORM_ROWS                37
#define VSYS_OP_GEGLU                        38
#define VSYS_OP_SPLITK_REDUCE_T              39
#define VSYS_OP_T5_ATTENTION                 40
#define VSYS_OP_GEMM_SKINNY_SLICES           41
#define VSYS_OP_SPLITK_REDUCE                42
#define VSYS_OP_T5_ATTENTION_MFMA            43
#define VSYS_OP_CONV_BF16                    44
#define VSYS_OP_GN_STATS                     45
#define VSYS_OP_GN_APPLY                     46
#define VSYS_OP_REGRID                       47
#define VSYS_OP_SUBSAMPLE                    48
#define VSYS_OP_SPATIAL_NORM_APPLY           49
#define VSYS_OP_BLEND_EDGE                   50
#define VSYS_OP_D2S_TIME                     51
#define VSYS_OP_VAE_FIRST_IM2COL             52
#define VSYS_OP_EXTRACT_PLANAR               53
#define VSYS_OP_SOFTMAX_ROWS                 54
#define VSYS_OP_COUNT 55
/* <<< VSYS_OP codes */

#define VSYS_CMD_MAX_INT 24
#define VSYS_CMD_MAX_FLOAT 4
typedef struct vsys_cmd {
  int32_t op;      /* VSYS_OP_* */
  int32_t stream;  /* index into the streams[] table of vsys_program_run */
  int64_t a[VSYS_CMD_MAX_INT];   /* integer and pointer arguments, declaration order, the trailing stream excluded */
  float f[VSYS_CMD_MAX_FLOAT];      /* float arguments, declaration order */
} vsys_cmd;

/* arity of an op (what the recorder must fill): 0, or VSYS_ERR_ARG for an unknown op.  (vsys_program_*: no reference counterpart —
 * the reference issues every op from Python every step, open_sora_transformer_3d.py:608-613.) */
int vsys_program_op_info(int op, int* n_int, int* n_float);
int vsys_program_run(const vsys_cmd* cmds, int64_t n, void* const* streams, int64_t n_streams, int64_t* failed_at);

#ifdef __cplusplus
}
#endif
#endif /* VIDEOSYS_AMD_H */
