// extern "C" entry points declared in include/videosys_amd.h: argument validation + launch, nothing else.
#include "common.h"
#include "vsys_internal.h"

using namespace vsys;

namespace {
inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline const bf16_t* B16(const void* p) { return reinterpret_cast<const bf16_t*>(p); }
inline bf16_t* B16(void* p) { return reinterpret_cast<bf16_t*>(p); }
inline bool fits_int(int64_t v) { return v >= 0 && v <= 0x7fffffff; }
}  // namespace

extern "C" {

int vsys_abi_version(void) { return VSYS_ABI_VERSION; }

const char* vsys_strerror(int code) {
  switch (code) {
    case 0: return "ok";
    case VSYS_ERR_SHAPE: return "unsupported shape";
    case VSYS_ERR_ALIGN: return "stride/alignment not a multiple of 8 elements";
    case VSYS_ERR_ARG: return "bad argument";
    case VSYS_ERR_LAUNCH: return "HIP launch failed";
    default: return "unknown error";
  }
}

int vsys_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int vsys_tune_gemm_variant(int variant) { return set_gemm_variant(variant); }

int vsys_tune_flash_variant(int variant) { return set_flash_variant(variant); }

int vsys_gemm_raster_probe(int64_t tile, int64_t nbm, int64_t nbn, int64_t gw, int64_t ph, int64_t* bm, int64_t* bn) {
  if (!bm || !bn || tile < 0 || nbm <= 0 || nbn <= 0 || tile >= nbm * nbn || !fits_int(nbm * nbn)) return VSYS_ERR_ARG;
  int m = 0, n = 0;
  gemm_raster((int)tile, (int)nbm, (int)nbn, (int)gw, (int)ph, m, n);
  *bm = m;
  *bn = n;
  return 0;
}

#ifdef VSYS_LAB   // include/videosys_amd_lab.h
int vsys_gemm_streamk_plan(int ntiles, int nt, int grid, int32_t* segs, int cap_rows, int* nseg_max) {
  if (ntiles <= 0 || nt <= 0 || grid <= 0 || segs == nullptr || nseg_max == nullptr) return VSYS_ERR_ARG;
  std::vector<int4> v;
  int m = 0;
  if (!sk_plan(ntiles, nt, grid, v, m)) return 0;
  if ((int)v.size() > cap_rows) return VSYS_ERR_ARG;
  for (size_t i = 0; i < v.size(); ++i) {
    segs[4 * i] = v[i].x; segs[4 * i + 1] = v[i].y; segs[4 * i + 2] = v[i].z; segs[4 * i + 3] = v[i].w;
  }
  *nseg_max = m;
  return (int)v.size();
}

int vsys_lab_flash_debug_buffer(void* dev_u64x5) {
  set_flash_debug_buffer(dev_u64x5);
  return 0;
}
#endif

int vsys_gemm_bf16(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, void* out, int64_t ldo,
                   int64_t M, int64_t N, int64_t K, int epilogue, const void* gate, int64_t gate_sample_stride,
                   int64_t rows_per_sample, const void* res, int64_t ldr, void* aux, int64_t ldaux, void* stream) {
  if (!x || !w || !out) return VSYS_ERR_ARG;
  if (!fits_int(M) || !fits_int(N) || !fits_int(K) || !fits_int(rows_per_sample)) return VSYS_ERR_SHAPE;
  GemmParams p;
  p.A = B16(x); p.lda = ldx; p.W = B16(w); p.ldw = ldw; p.bias = B16(bias); p.out = B16(out); p.ldo = ldo;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.gate = B16(gate); p.gate_stride = gate_sample_stride; p.res = B16(res); p.ldr = ldr; p.aux = B16(aux); p.ldaux = ldaux;
  p.rows_per_sample = (int)rows_per_sample;
  p.seg_split = 0; p.gate_alt = 0; p.ks = 0; p.out32 = nullptr; p.slab = 0; p.ldo32 = 0;
  if (epilogue != VSYS_EPI_GATE_RES && (gate || res || aux)) return VSYS_ERR_ARG;
  return launch_gemm(p, epilogue, S(stream));
}

int vsys_gemm_bf16_ln(const void* x, int64_t ldx, const void* wp, int64_t ldw, const void* cs, const void* cv, void* out, int64_t ldo,
                      int64_t M, int64_t N, int64_t K, int epilogue, const void* stats, int64_t stats_ld, float eps, void* stream) {
  if (!x || !wp || !cs || !cv || !out || !stats) return VSYS_ERR_ARG;
  if (!fits_int(M) || !fits_int(N) || !fits_int(K)) return VSYS_ERR_SHAPE;
  if (epilogue != VSYS_EPI_BIAS && epilogue != VSYS_EPI_BIAS_GELU) return VSYS_ERR_ARG;
  if (K % LN_BLOCK != 0 || K / LN_BLOCK > 12) return VSYS_ERR_SHAPE;
  GemmParams p;
  p.A = B16(x); p.lda = ldx; p.W = B16(wp); p.ldw = ldw; p.bias = nullptr; p.out = B16(out); p.ldo = ldo;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.gate = nullptr; p.gate_stride = 0; p.res = nullptr; p.ldr = 0; p.aux = nullptr; p.ldaux = 0; p.rows_per_sample = 0;
  p.seg_split = 0; p.gate_alt = 0; p.ks = 0; p.out32 = nullptr; p.slab = 0; p.ldo32 = 0;
  p.cs = reinterpret_cast<const float*>(cs); p.cv = reinterpret_cast<const float*>(cv);
  p.ln_stats = reinterpret_cast<const float2*>(stats); p.ln_ld = stats_ld; p.ln_nb = (int)(K / LN_BLOCK); p.ln_eps = eps;
  return launch_gemm(p, epilogue == VSYS_EPI_BIAS ? EPI_LN_BIAS : EPI_LN_GELU, S(stream));
}

int vsys_gemm_bf16_stats(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, void* out, int64_t ldo, int64_t M,
                         int64_t N, int64_t K, const void* gate, int64_t gate_sample_stride, int64_t rows_per_sample, const void* res,
                         int64_t ldr, void* stats, int64_t stats_ld, void* stream) {
  if (!x || !w || !out || !stats) return VSYS_ERR_ARG;
  if (!fits_int(M) || !fits_int(N) || !fits_int(K) || !fits_int(rows_per_sample)) return VSYS_ERR_SHAPE;
  GemmParams p;
  p.A = B16(x); p.lda = ldx; p.W = B16(w); p.ldw = ldw; p.bias = B16(bias); p.out = B16(out); p.ldo = ldo;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.gate = B16(gate); p.gate_stride = gate_sample_stride; p.res = B16(res); p.ldr = ldr; p.aux = nullptr; p.ldaux = 0;
  p.rows_per_sample = (int)rows_per_sample;
  p.seg_split = 0; p.gate_alt = 0; p.ks = 0; p.out32 = nullptr; p.slab = 0; p.ldo32 = 0;
  p.stats_out = reinterpret_cast<float2*>(stats); p.stats_ld = stats_ld;
  return launch_gemm(p, EPI_GATE_RES_STATS, S(stream));
}

int vsys_gemm_bf16_gate_res_add(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, void* out, int64_t ldo,
                                int64_t M, int64_t N, int64_t K, const void* gate, int64_t gate_sample_stride, int64_t rows_per_sample,
                                const void* res, int64_t ldr, void* aux, const void* add1, const void* add2, void* stats,
                                int64_t stats_ld, void* stream) {
  if (!x || !w || !out || !res) return VSYS_ERR_ARG;
  if (!fits_int(M) || !fits_int(N) || !fits_int(K) || !fits_int(rows_per_sample)) return VSYS_ERR_SHAPE;
  GemmParams p;
  p.A = B16(x); p.lda = ldx; p.W = B16(w); p.ldw = ldw; p.bias = B16(bias); p.out = B16(out); p.ldo = ldo;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.gate = B16(gate); p.gate_stride = gate_sample_stride; p.res = B16(res); p.ldr = ldr; p.aux = B16(aux); p.ldaux = ldr;
  p.rows_per_sample = (int)rows_per_sample;
  p.seg_split = 0; p.gate_alt = 0; p.ks = 0; p.out32 = nullptr; p.slab = 0; p.ldo32 = 0;
  p.add1 = B16(add1 ? add1 : add2); p.add2 = add1 ? B16(add2) : nullptr;
  p.stats_out = reinterpret_cast<float2*>(stats); p.stats_ld = stats_ld;
  return launch_gemm(p, EPI_GATE_RES, S(stream));
}

int vsys_adaln_prescale(const void* sites, int64_t nsites, int64_t nblocks, const void* mod, void* stream) {
  if (!sites || !mod) return VSYS_ERR_ARG;
  if (!fits_int(nsites)) return VSYS_ERR_SHAPE;
  return launch_adaln_prescale(reinterpret_cast<const int64_t*>(sites), (int)nsites, nblocks, B16(mod), S(stream));
}

int vsys_ln_row_stats(const void* x, int64_t rows, int64_t C, void* stats, int64_t stats_ld, void* stream) {
  if (!x || !stats) return VSYS_ERR_ARG;
  if (!fits_int(C)) return VSYS_ERR_SHAPE;
  return launch_ln_row_stats(B16(x), rows, (int)C, reinterpret_cast<float2*>(stats), stats_ld, S(stream));
}

int vsys_gemm_bf16_gate2(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, void* out, int64_t ldo,
                         int64_t M, int64_t N, int64_t K, const void* gate, int64_t gate_sample_stride,
                         int64_t rows_per_sample, int64_t seg_split, int64_t gate_alt, const void* res, int64_t ldr,
                         void* aux, int64_t ldaux, void* stream) {
  if (!x || !w || !out || !gate) return VSYS_ERR_ARG;
  if (!fits_int(M) || !fits_int(N) || !fits_int(K) || !fits_int(rows_per_sample) || !fits_int(seg_split)) return VSYS_ERR_SHAPE;
  if (seg_split < 0 || seg_split > rows_per_sample) return VSYS_ERR_SHAPE;
  GemmParams p;
  p.A = B16(x); p.lda = ldx; p.W = B16(w); p.ldw = ldw; p.bias = B16(bias); p.out = B16(out); p.ldo = ldo;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.gate = B16(gate); p.gate_stride = gate_sample_stride; p.res = B16(res); p.ldr = ldr; p.aux = B16(aux); p.ldaux = ldaux;
  p.rows_per_sample = (int)rows_per_sample;
  p.seg_split = (int)seg_split; p.gate_alt = gate_alt; p.ks = 0; p.out32 = nullptr; p.slab = 0; p.ldo32 = 0;
  return launch_gemm(p, VSYS_EPI_GATE_RES, S(stream));
}

int vsys_ln_modulate(const void* x, const void* ln_w, const void* ln_b, const void* shift, const void* scale, void* y,
                     int64_t rows, int64_t C, int64_t rows_per_sample, int64_t mod_sample_stride, int64_t seg_split,
                     int64_t mod_alt, float eps, void* stream) {
  if (!x || !y) return VSYS_ERR_ARG;
  if (!fits_int(C)) return VSYS_ERR_SHAPE;
  return launch_ln_modulate(B16(x), B16(ln_w), B16(ln_b), B16(shift), B16(scale), B16(y), rows, (int)C, rows_per_sample,
                            mod_sample_stride, seg_split, mod_alt, eps, S(stream));
}

int vsys_gate_add_rows(void* x, const void* y, const void* gate, int64_t rows, int64_t C, int64_t rows_per_sample,
                       int64_t gate_sample_stride, int64_t seg_split, int64_t gate_alt, void* stream) {
  if (!x || !y || !gate) return VSYS_ERR_ARG;
  if (!fits_int(C)) return VSYS_ERR_SHAPE;
  return launch_gate_add_rows(B16(x), B16(y), B16(gate), rows, (int)C, rows_per_sample, gate_sample_stride, seg_split, gate_alt,
                              S(stream));
}

int vsys_im2col_patch(const void* z_f32, int64_t Bz, void* out, int64_t B, int64_t F, int64_t Cin, int64_t H, int64_t W,
                      int64_t p, void* stream) {
  if (!z_f32 || !out) return VSYS_ERR_ARG;
  if (!fits_int(B) || !fits_int(F) || !fits_int(Cin) || !fits_int(H) || !fits_int(W) || !fits_int(p) || !fits_int(Bz)) return VSYS_ERR_SHAPE;
  if ((Cin * p * p) % 8) return VSYS_ERR_SHAPE;
  return launch_im2col_patch(reinterpret_cast<const float*>(z_f32), (int)Bz, B16(out), (int)B, (int)F, (int)Cin, (int)H, (int)W,
                             (int)p, S(stream));
}

int vsys_unpatchify_cvx(const void* x, int64_t ldx, void* out_f32, int64_t B, int64_t F, int64_t Hp, int64_t Wp, int64_t Cout,
                        int64_t p, void* stream) {
  if (!x || !out_f32) return VSYS_ERR_ARG;
  if (!fits_int(B) || !fits_int(F) || !fits_int(Hp) || !fits_int(Wp) || !fits_int(Cout) || !fits_int(p)) return VSYS_ERR_SHAPE;
  return launch_unpatchify_cvx(B16(x), ldx, reinterpret_cast<float*>(out_f32), (int)B, (int)F, (int)Hp, (int)Wp, (int)Cout,
                               (int)p, S(stream));
}

int vsys_attn_prep_kv64(const void* k, int64_t k_stride, const void* v, int64_t v_stride, const void* ln_w, const void* ln_b,
                        const void* rope_cos_f32, const void* rope_sin_f32, int64_t rope_start, int64_t rope_len, void* kp,
                        void* vt, int64_t batch, int64_t heads, int64_t kv_len, int64_t kv_pad, float eps, void* stream) {
  if (!k || !v || !kp || !vt) return VSYS_ERR_ARG;
  if ((ln_w == nullptr) != (ln_b == nullptr)) return VSYS_ERR_ARG;
  if (!fits_int(batch) || !fits_int(heads) || !fits_int(kv_len) || !fits_int(kv_pad) || !fits_int(rope_start) || !fits_int(rope_len))
    return VSYS_ERR_SHAPE;
  return launch_attn_prep_kv64(B16(k), k_stride, B16(v), v_stride, B16(ln_w), B16(ln_b),
                               reinterpret_cast<const float*>(rope_cos_f32), reinterpret_cast<const float*>(rope_sin_f32),
                               (int)rope_start, (int)rope_len, B16(kp), B16(vt), (int)batch, (int)heads, (int)kv_len, (int)kv_pad,
                               eps, S(stream));
}

int vsys_flash_attn_d64(const void* q, int64_t q_stride, const void* ln_w, const void* ln_b, const void* rope_cos_f32,
                        const void* rope_sin_f32, int64_t rope_start, int64_t rope_len, const void* kp, const void* vt, void* out,
                        int64_t out_stride, int64_t batch, int64_t heads, int64_t q_len, int64_t kv_len, int64_t kv_pad, float eps,
                        void* stream) {
  if (!q || !kp || !vt || !out) return VSYS_ERR_ARG;
  if ((ln_w == nullptr) != (ln_b == nullptr)) return VSYS_ERR_ARG;
  if (!fits_int(batch) || !fits_int(heads) || !fits_int(q_len) || !fits_int(kv_len) || !fits_int(kv_pad) || !fits_int(rope_start) ||
      !fits_int(rope_len))
    return VSYS_ERR_SHAPE;
  return launch_flash_attn_d64(B16(q), q_stride, B16(ln_w), B16(ln_b), reinterpret_cast<const float*>(rope_cos_f32),
                               reinterpret_cast<const float*>(rope_sin_f32), (int)rope_start, (int)rope_len, B16(kp), B16(vt),
                               B16(out), out_stride, (int)batch, (int)heads, (int)q_len, (int)kv_len, (int)kv_pad, eps, 0.f, S(stream));
}

int vsys_flash_attn_d64_kb(const void* q, int64_t q_stride, const void* ln_w, const void* ln_b, const void* rope_cos_f32,
                           const void* rope_sin_f32, int64_t rope_start, int64_t rope_len, const void* kp, const void* vt, void* out,
                           int64_t out_stride, int64_t batch, int64_t heads, int64_t q_len, int64_t kv_len, int64_t kv_pad, float eps,
                           float k_norm_bound, void* stream) {
  if (!q || !kp || !vt || !out || !(k_norm_bound >= 0.f)) return VSYS_ERR_ARG;
  if ((ln_w == nullptr) != (ln_b == nullptr)) return VSYS_ERR_ARG;
  if (!fits_int(batch) || !fits_int(heads) || !fits_int(q_len) || !fits_int(kv_len) || !fits_int(kv_pad) || !fits_int(rope_start) ||
      !fits_int(rope_len))
    return VSYS_ERR_SHAPE;
  return launch_flash_attn_d64(B16(q), q_stride, B16(ln_w), B16(ln_b), reinterpret_cast<const float*>(rope_cos_f32),
                               reinterpret_cast<const float*>(rope_sin_f32), (int)rope_start, (int)rope_len, B16(kp), B16(vt),
                               B16(out), out_stride, (int)batch, (int)heads, (int)q_len, (int)kv_len, (int)kv_pad, eps, k_norm_bound,
                               S(stream));
}

int vsys_linear_small(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, void* out, int64_t ldo,
                      int64_t M, int64_t N, int64_t K, int act_in, int act_out, void* stream) {
  if (!x || !w || !out) return VSYS_ERR_ARG;
  if (!fits_int(M) || !fits_int(N) || !fits_int(K)) return VSYS_ERR_SHAPE;
  return launch_linear_small(B16(x), ldx, B16(w), ldw, B16(bias), B16(out), ldo, (int)M, (int)N, (int)K, act_in, act_out,
                             S(stream));
}

int vsys_adaln_modulate(const void* x, const void* shift, const void* scale, void* y, int64_t rows, int64_t C,
                        int64_t rows_per_sample, int64_t mod_sample_stride, float eps, void* stream) {
  if (!x || !shift || !scale || !y) return VSYS_ERR_ARG;
  if (!fits_int(C)) return VSYS_ERR_SHAPE;
  return launch_adaln_modulate(B16(x), B16(shift), B16(scale), B16(y), rows, (int)C, rows_per_sample, mod_sample_stride, eps,
                               S(stream));
}

int vsys_mod_table(const void* table, const void* t_mlp, void* out, int64_t nblk, int64_t B, int64_t C6, void* stream) {
  if (!table || !t_mlp || !out) return VSYS_ERR_ARG;
  if (!fits_int(nblk) || !fits_int(B) || !fits_int(C6)) return VSYS_ERR_SHAPE;
  return launch_mod_table(B16(table), B16(t_mlp), B16(out), (int)nblk, (int)B, (int)C6, S(stream));
}

int vsys_timestep_embedding(const void* t_f32, void* out, int64_t B, int64_t dim, void* stream) {
  if (!t_f32 || !out) return VSYS_ERR_ARG;
  if (!fits_int(B) || !fits_int(dim)) return VSYS_ERR_SHAPE;
  return launch_timestep_embedding(reinterpret_cast<const float*>(t_f32), B16(out), (int)B, (int)dim, S(stream));
}

int vsys_patch_embed(const void* z_f32, int64_t Bz, const void* w, const void* bias, const void* pos, void* out, int64_t B,
                     int64_t Cin, int64_t T, int64_t H, int64_t W, int64_t ph, int64_t pw, int64_t C, void* stream) {
  if (!z_f32 || !w || !bias || !pos || !out) return VSYS_ERR_ARG;
  if (!fits_int(B) || !fits_int(T) || !fits_int(H) || !fits_int(W) || !fits_int(C) || ph <= 0 || pw <= 0) return VSYS_ERR_SHAPE;
  return launch_patch_embed(reinterpret_cast<const float*>(z_f32), (int)Bz, B16(w), B16(bias), B16(pos), B16(out), (int)B,
                            (int)Cin, (int)T, (int)H, (int)W, (int)ph, (int)pw, (int)C, 0, -1, S(stream));
}

int vsys_patch_embed_shard(const void* z_f32, int64_t Bz, const void* w, const void* bias, const void* pos, void* out, int64_t B,
                           int64_t Cin, int64_t T, int64_t H, int64_t W, int64_t ph, int64_t pw, int64_t C, int64_t s0, int64_t Sl,
                           void* stream) {
  if (!z_f32 || !w || !bias || !pos || !out) return VSYS_ERR_ARG;
  if (!fits_int(B) || !fits_int(T) || !fits_int(H) || !fits_int(W) || !fits_int(C) || ph <= 0 || pw <= 0 || !fits_int(s0) ||
      !fits_int(Sl))
    return VSYS_ERR_SHAPE;
  return launch_patch_embed(reinterpret_cast<const float*>(z_f32), (int)Bz, B16(w), B16(bias), B16(pos), B16(out), (int)B,
                            (int)Cin, (int)T, (int)H, (int)W, (int)ph, (int)pw, (int)C, (int)s0, (int)Sl, S(stream));
}

int vsys_final_layer(const void* x, const void* table, const void* tvec, const void* w, const void* bias, void* out_f32,
                     int64_t B, int64_t T, int64_t Hp, int64_t Wp, int64_t H, int64_t W, int64_t ph, int64_t pw,
                     int64_t Cout, int64_t C, float eps, void* stream) {
  if (!x || !table || !tvec || !w || !bias || !out_f32) return VSYS_ERR_ARG;
  if (!fits_int(B) || !fits_int(T) || !fits_int(Hp) || !fits_int(Wp) || !fits_int(C)) return VSYS_ERR_SHAPE;
  return launch_final_layer(B16(x), B16(table), B16(tvec), B16(w), B16(bias), reinterpret_cast<float*>(out_f32), (int)B,
                            (int)T, (int)Hp, (int)Wp, (int)H, (int)W, (int)ph, (int)pw, (int)Cout, (int)C, eps, -1, nullptr, S(stream));
}

int vsys_final_layer_tokens(const void* x, const void* table, const void* tvec, const void* w, const void* bias, void* tokens_f32,
                            int64_t B, int64_t T, int64_t Sl, int64_t n_out, int64_t C, float eps, void* stream) {
  if (!x || !table || !tvec || !w || !bias || !tokens_f32) return VSYS_ERR_ARG;
  if (!fits_int(B) || !fits_int(T) || !fits_int(Sl) || !fits_int(C) || n_out <= 0 || n_out > 64) return VSYS_ERR_SHAPE;
  // (ph, pw, Cout) = (1, 1, n_out): the token-major form needs only their product
  return launch_final_layer(B16(x), B16(table), B16(tvec), B16(w), B16(bias), nullptr, (int)B, (int)T, 1, (int)Sl, 1, 1, 1, 1,
                            (int)n_out, (int)C, eps, (int)Sl, reinterpret_cast<float*>(tokens_f32), S(stream));
}

int vsys_unpatchify_tokens(const void* tokens_f32, void* out_f32, int64_t P, int64_t B, int64_t T, int64_t Sl, int64_t Hp,
                           int64_t Wp, int64_t H, int64_t W, int64_t ph, int64_t pw, int64_t Cout, void* stream) {
  if (!tokens_f32 || !out_f32) return VSYS_ERR_ARG;
  if (!fits_int(P) || !fits_int(B) || !fits_int(T) || !fits_int(Sl) || !fits_int(Hp) || !fits_int(Wp) || !fits_int(H) || !fits_int(W) ||
      ph <= 0 || pw <= 0 || Cout <= 0 || Hp * ph < H || Wp * pw < W)
    return VSYS_ERR_SHAPE;
  return launch_unpatchify_tokens(reinterpret_cast<const float*>(tokens_f32), reinterpret_cast<float*>(out_f32), (int)P, (int)B, (int)T,
                                  (int)Sl, (int)Hp, (int)Wp, (int)H, (int)W, (int)ph, (int)pw, (int)Cout, S(stream));
}

int vsys_cfg_euler_step(void* z_f32, const void* model_out_f32, int64_t Bz, int64_t Cin, int64_t Cout, int64_t thw,
                        float guidance, float dt, void* stream) {
  if (!z_f32 || !model_out_f32) return VSYS_ERR_ARG;
  if (!fits_int(Bz) || !fits_int(Cin) || !fits_int(Cout) || Cout < Cin) return VSYS_ERR_SHAPE;
  return launch_cfg_euler(reinterpret_cast<float*>(z_f32), reinterpret_cast<const float*>(model_out_f32), (int)Bz, (int)Cin,
                          (int)Cout, thw, guidance, dt, S(stream));
}

int vsys_cfg_linear_step(void* z_f32, const void* model_out_f32, int64_t Bz, int64_t Cin, int64_t Cout, int64_t thw,
                         float guidance, float c_z, float c_eps, int cond_first, void* stream) {
  if (!z_f32 || !model_out_f32) return VSYS_ERR_ARG;
  if (!fits_int(Bz) || !fits_int(Cin) || !fits_int(Cout) || Cout < Cin) return VSYS_ERR_SHAPE;
  return launch_cfg_axpby(reinterpret_cast<float*>(z_f32), reinterpret_cast<const float*>(model_out_f32), (int)Bz, (int)Cin,
                          (int)Cout, thw, guidance, c_z, c_eps, cond_first, S(stream));
}

int vsys_add_bcast_rows(void* x, const void* e, int64_t rows, int64_t C, int64_t group, int64_t period, void* stream) {
  if (!x || !e) return VSYS_ERR_ARG;
  if (!fits_int(C)) return VSYS_ERR_SHAPE;
  return launch_add_bcast_rows(B16(x), B16(e), rows, (int)C, group, period, S(stream));
}

int vsys_add_rows(void* x, const void* y, int64_t n, void* stream) {
  if (!x || !y) return VSYS_ERR_ARG;
  return launch_add_rows(B16(x), B16(y), n, S(stream));
}

int vsys_copy_4d(const void* src, void* dst, int64_t n0, int64_t n1, int64_t n2, int64_t C, int64_t ss0, int64_t ss1,
                 int64_t ss2, int64_t ds0, int64_t ds1, int64_t ds2, int64_t n1_valid, int64_t n2_valid, void* stream) {
  if (!src || !dst) return VSYS_ERR_ARG;
  if (!fits_int(n0) || !fits_int(n1) || !fits_int(n2) || !fits_int(C)) return VSYS_ERR_SHAPE;
  if ((ss0 % 8) || (ss1 % 8) || (ss2 % 8) || (ds0 % 8) || (ds1 % 8) || (ds2 % 8)) return VSYS_ERR_ALIGN;
  return launch_copy_4d(B16(src), B16(dst), (int)n0, (int)n1, (int)n2, (int)C, ss0, ss1, ss2, ds0, ds1, ds2, (int)n1_valid,
                        (int)n2_valid, S(stream));
}

int vsys_attn_prep_kv(const void* k, int64_t k_stride, const void* v, int64_t v_stride, const void* k_norm_w, void* kp,
                      void* vt, int64_t batch, int64_t heads, int64_t kv_len, int64_t kv_pad, float eps, void* stream) {
  if (!k || !v || !kp || !vt) return VSYS_ERR_ARG;
  if (!fits_int(batch) || !fits_int(heads) || !fits_int(kv_len) || !fits_int(kv_pad) || batch * heads > 65535) return VSYS_ERR_SHAPE;
  return launch_attn_prep_kv(B16(k), k_stride, B16(v), v_stride, B16(k_norm_w), B16(kp), B16(vt), (int)batch, (int)heads,
                             (int)kv_len, (int)kv_pad, eps, S(stream));
}

int vsys_flash_attn_d72(const void* q, int64_t q_stride, const void* q_norm_w, const void* kp, const void* vt, void* out,
                        int64_t out_stride, int64_t batch, int64_t heads, int64_t q_len, int64_t kv_len, int64_t kv_pad,
                        float eps, void* stream) {
  if (!q || !kp || !vt || !out) return VSYS_ERR_ARG;
  if (!fits_int(batch) || !fits_int(heads) || !fits_int(q_len) || !fits_int(kv_len) || !fits_int(kv_pad) ||
      batch * heads > 65535)
    return VSYS_ERR_SHAPE;
  return launch_flash_attn_d72(B16(q), q_stride, B16(q_norm_w), B16(kp), B16(vt), B16(out), out_stride, (int)batch,
                               (int)heads, (int)q_len, (int)kv_len, (int)kv_pad, eps, 0.f, S(stream));
}

int vsys_flash_attn_d72_exact(const void* q, int64_t q_stride, const void* q_norm_w, const void* kp, const void* vt, void* out,
                              int64_t out_stride, int64_t batch, int64_t heads, int64_t q_len, int64_t kv_len, int64_t kv_pad,
                              float eps, void* stream) {
  if (!q || !kp || !vt || !out) return VSYS_ERR_ARG;
  if (!fits_int(batch) || !fits_int(heads) || !fits_int(q_len) || !fits_int(kv_len) || !fits_int(kv_pad) ||
      batch * heads > 65535)
    return VSYS_ERR_SHAPE;
  return launch_flash_attn_d72(B16(q), q_stride, B16(q_norm_w), B16(kp), B16(vt), B16(out), out_stride, (int)batch,
                               (int)heads, (int)q_len, (int)kv_len, (int)kv_pad, eps, 0.f, S(stream), true);
}

int vsys_flash_attn_d72_kb(const void* q, int64_t q_stride, const void* q_norm_w, const void* kp, const void* vt, void* out,
                           int64_t out_stride, int64_t batch, int64_t heads, int64_t q_len, int64_t kv_len, int64_t kv_pad,
                           float eps, float k_norm_bound, void* stream) {
  if (!q || !kp || !vt || !out || !(k_norm_bound >= 0.f)) return VSYS_ERR_ARG;
  if (!fits_int(batch) || !fits_int(heads) || !fits_int(q_len) || !fits_int(kv_len) || !fits_int(kv_pad) ||
      batch * heads > 65535)
    return VSYS_ERR_SHAPE;
  return launch_flash_attn_d72(B16(q), q_stride, B16(q_norm_w), B16(kp), B16(vt), B16(out), out_stride, (int)batch,
                               (int)heads, (int)q_len, (int)kv_len, (int)kv_pad, eps, k_norm_bound, S(stream));
}

int vsys_attn_temporal_d72(const void* qkv, int64_t row_stride, int64_t C, const void* q_norm_w, const void* k_norm_w,
                           const void* rope_cos_f32, const void* rope_sin_f32, void* out, int64_t out_stride, int64_t B,
                           int64_t T, int64_t S_, int64_t heads, float eps, void* stream) {
  if (!qkv || !out || ((q_norm_w == nullptr) != (k_norm_w == nullptr))) return VSYS_ERR_ARG;
  if ((rope_cos_f32 == nullptr) != (rope_sin_f32 == nullptr)) return VSYS_ERR_ARG;
  if (!fits_int(B) || !fits_int(T) || !fits_int(S_) || !fits_int(heads) || !fits_int(C)) return VSYS_ERR_SHAPE;
  return launch_attn_temporal_d72(B16(qkv), row_stride, (int)C, B16(q_norm_w), B16(k_norm_w),
                                  reinterpret_cast<const float*>(rope_cos_f32), reinterpret_cast<const float*>(rope_sin_f32),
                                  B16(out), out_stride, (int)B, (int)T, (int)S_, (int)heads, eps, S(stream));
}

namespace {
inline bool to_grid(const int64_t* g, VaeGrid& o) {
  if (!g) return false;
  for (int i = 0; i < 5; ++i) if (!fits_int(g[i])) return false;
  o.T = (int)g[0]; o.H = (int)g[1]; o.W = (int)g[2]; o.pad = (int)g[3]; o.tf = (int)g[4]; o.sample_rows = g[5];
  return g[5] >= 0;
}
}  // namespace

int vsys_gather_rows(const void* table, const void* ids_i64, void* out, int64_t n, int64_t C, int64_t vocab, void* stream) {
  if (!table || !ids_i64 || !out) return VSYS_ERR_ARG;
  if (!fits_int(C)) return VSYS_ERR_SHAPE;
  return launch_gather_rows(B16(table), reinterpret_cast<const int64_t*>(ids_i64), B16(out), n, (int)C, vocab, S(stream));
}

int vsys_rms_norm_rows(const void* x, const void* w, void* y, int64_t rows, int64_t C, float eps, void* stream) {
  if (!x || !w || !y) return VSYS_ERR_ARG;
  if (!fits_int(C)) return VSYS_ERR_SHAPE;
  return launch_rms_norm_rows(B16(x), B16(w), B16(y), rows, (int)C, eps, S(stream));
}

int vsys_geglu(const void* h, void* out, int64_t rows, int64_t F, void* stream) {
  if (!h || !out) return VSYS_ERR_ARG;
  if (!fits_int(F)) return VSYS_ERR_SHAPE;
  return launch_geglu(B16(h), B16(out), rows, (int)F, S(stream));
}

int vsys_gemm_skinny_slices(const void* w, int64_t ldw, const void* x, int64_t ldx, void* part_f32, int64_t rows, int64_t rows_padded,
                            int64_t N, int64_t K, int64_t nsplit, void* stream) {
  if (!w || !x || !part_f32) return VSYS_ERR_ARG;
  if (!fits_int(rows_padded) || !fits_int(N) || !fits_int(K) || !fits_int(nsplit) || nsplit < 1 || K % 32 != 0 || rows < 1 || rows > rows_padded)
    return VSYS_ERR_SHAPE;
  GemmParams p;
  p.A = B16(w); p.lda = ldw; p.W = B16(x); p.ldw = ldx; p.bias = nullptr; p.out = nullptr; p.ldo = 0;
  p.M = (int)N; p.N = (int)rows_padded; p.K = (int)K;
  p.gate = nullptr; p.gate_stride = 0; p.res = nullptr; p.ldr = 0; p.aux = nullptr; p.ldaux = 0; p.rows_per_sample = (int)rows;
  p.seg_split = 0; p.gate_alt = 0;
  p.ks = (int)(((K / 32 + nsplit - 1) / nsplit) * 32);   // k-tiles dealt out evenly, the last slice takes what is left
  if ((int64_t)p.ks * (nsplit - 1) >= K) return VSYS_ERR_SHAPE;
  p.out32 = reinterpret_cast<float*>(part_f32); p.slab = rows_padded * N; p.ldo32 = N;
  return launch_gemm2_slices(p, (int)nsplit, S(stream));
}

int vsys_splitk_reduce(const void* part_f32, int64_t nsplit, int64_t slab, int64_t ldp, const void* res, int64_t ldr, void* out, int64_t ldo,
                       int64_t M, int64_t N, void* stream) {
  if (!part_f32 || !out) return VSYS_ERR_ARG;
  if (!fits_int(nsplit) || !fits_int(M) || !fits_int(N)) return VSYS_ERR_SHAPE;
  return launch_splitk_reduce(reinterpret_cast<const float*>(part_f32), (int)nsplit, slab, ldp, B16(res), ldr, B16(out), ldo, (int)M, (int)N,
                              S(stream));
}

int vsys_t5_attention_mfma(const void* qkv, int64_t row_stride, int64_t inner, const void* bias_f32, int64_t bias_ld, int64_t bias_center,
                           int64_t kv_len, void* kp, void* vt, void* out, int64_t out_stride, int64_t L, int64_t heads, void* stream) {
  if (!qkv || !bias_f32 || !kp || !vt || !out) return VSYS_ERR_ARG;
  if (!fits_int(inner) || !fits_int(bias_ld) || !fits_int(bias_center) || !fits_int(kv_len) || !fits_int(L) || !fits_int(heads))
    return VSYS_ERR_SHAPE;
  return launch_t5_attention_mfma(B16(qkv), row_stride, (int)inner, reinterpret_cast<const float*>(bias_f32), (int)bias_ld,
                                  (int)bias_center, (int)kv_len, B16(kp), B16(vt), B16(out), out_stride, (int)L, (int)heads, S(stream));
}

int vsys_splitk_reduce_t(const void* part_f32, int64_t nsplit, int64_t slab, int64_t ldp, const void* res, int64_t ldr, void* out, int64_t ldo,
                         int64_t M, int64_t N, void* stream) {
  if (!part_f32 || !out) return VSYS_ERR_ARG;
  if (!fits_int(nsplit) || !fits_int(ldp) || !fits_int(M) || !fits_int(N)) return VSYS_ERR_SHAPE;
  return launch_splitk_reduce_t(reinterpret_cast<const float*>(part_f32), (int)nsplit, slab, (int)ldp, B16(res), ldr, B16(out), ldo, (int)M,
                                (int)N, S(stream));
}

int vsys_t5_attention(const void* qkv, int64_t row_stride, int64_t inner, const void* relbias_f32, const void* klen_i32, void* out,
                      int64_t out_stride, int64_t B, int64_t L, int64_t heads, void* stream) {
  if (!qkv || !relbias_f32 || !klen_i32 || !out) return VSYS_ERR_ARG;
  if (!fits_int(inner) || !fits_int(B) || !fits_int(L) || !fits_int(heads)) return VSYS_ERR_SHAPE;
  return launch_t5_attention(B16(qkv), row_stride, (int)inner, reinterpret_cast<const float*>(relbias_f32),
                             reinterpret_cast<const int*>(klen_i32), B16(out), out_stride, (int)B, (int)L, (int)heads, S(stream));
}

int vsys_copy_4d_batch(const void* src, void* dst, int64_t nops, const int64_t* desc, void* stream) {
  if (!src || !dst || (nops > 0 && !desc)) return VSYS_ERR_ARG;
  if (nops < 0 || nops > VSYS_COPY_BATCH_MAX) return VSYS_ERR_SHAPE;
  CopyDesc ops[VSYS_COPY_BATCH_MAX];
  for (int i = 0; i < (int)nops; ++i) {
    const int64_t* d = desc + 14 * i;
    for (int k = 2; k < 6; ++k) if (!fits_int(d[k])) return VSYS_ERR_SHAPE;
    if (!fits_int(d[12]) || !fits_int(d[13]) || d[0] < 0 || d[1] < 0) return VSYS_ERR_SHAPE;
    ops[i].src_off = d[0]; ops[i].dst_off = d[1];
    ops[i].n0 = (int)d[2]; ops[i].n1 = (int)d[3]; ops[i].n2 = (int)d[4]; ops[i].C = (int)d[5];
    ops[i].ss0 = d[6]; ops[i].ss1 = d[7]; ops[i].ss2 = d[8]; ops[i].ds0 = d[9]; ops[i].ds1 = d[10]; ops[i].ds2 = d[11];
    ops[i].n1_valid = (int)d[12]; ops[i].n2_valid = (int)d[13];
  }
  return launch_copy_4d_batch(B16(src), B16(dst), ops, (int)nops, S(stream));
}

int vsys_p2p_exchange(const void* src, int64_t nops, const int64_t* desc, const void* my_flags, int64_t n_flags, int64_t self_index,
                      void* state, int64_t timeout_ticks, void* stream) {
  if (!src || !my_flags || !state || (nops > 0 && !desc)) return VSYS_ERR_ARG;
  if (nops < 0 || nops > VSYS_COPY_BATCH_MAX || !fits_int(n_flags) || self_index < 0 || self_index >= n_flags) return VSYS_ERR_SHAPE;
  CopyDesc ops[VSYS_COPY_BATCH_MAX];
  bf16_t* dsts[VSYS_COPY_BATCH_MAX];
  unsigned* flags[VSYS_COPY_BATCH_MAX];
  int remote[VSYS_COPY_BATCH_MAX];
  for (int i = 0; i < (int)nops; ++i) {
    const int64_t* d = desc + 17 * i;
    for (int k = 2; k < 6; ++k) if (!fits_int(d[k])) return VSYS_ERR_SHAPE;
    if (!fits_int(d[12]) || !fits_int(d[13]) || d[0] < 0 || d[1] < 0 || d[14] == 0) return VSYS_ERR_SHAPE;
    ops[i].src_off = d[0]; ops[i].dst_off = d[1];
    ops[i].n0 = (int)d[2]; ops[i].n1 = (int)d[3]; ops[i].n2 = (int)d[4]; ops[i].C = (int)d[5];
    ops[i].ss0 = d[6]; ops[i].ss1 = d[7]; ops[i].ss2 = d[8]; ops[i].ds0 = d[9]; ops[i].ds1 = d[10]; ops[i].ds2 = d[11];
    ops[i].n1_valid = (int)d[12]; ops[i].n2_valid = (int)d[13];
    dsts[i] = reinterpret_cast<bf16_t*>(d[14]);
    flags[i] = reinterpret_cast<unsigned*>(d[15]);
    remote[i] = d[16] != 0;
  }
  return launch_p2p_exchange(B16(src), ops, dsts, flags, remote, (int)nops, reinterpret_cast<const unsigned*>(my_flags), (int)n_flags,
                             (int)self_index, reinterpret_cast<unsigned*>(state), (long long)timeout_ticks, S(stream));
}

int vsys_p2p_alloc(int64_t bytes, int64_t fine_grained, void** ptr) {
  if (!ptr || bytes <= 0) return VSYS_ERR_ARG;
  void* p = nullptr;
  hipError_t e = fine_grained ? hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained) : hipMalloc(&p, (size_t)bytes);
  if (e != hipSuccess || !p) return VSYS_ERR_LAUNCH;
  if (hipMemset(p, 0, (size_t)bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return VSYS_ERR_LAUNCH; }
  *ptr = p;
  return 0;
}

int vsys_p2p_free(void* ptr) { return (!ptr || hipFree(ptr) == hipSuccess) ? 0 : VSYS_ERR_LAUNCH; }

int vsys_p2p_ipc_export(const void* ptr, void* handle64) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle travels as 64 opaque bytes");
  if (!ptr || !handle64) return VSYS_ERR_ARG;
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, const_cast<void*>(ptr)) != hipSuccess) return VSYS_ERR_LAUNCH;
  __builtin_memcpy(handle64, &h, 64);
  return 0;
}

int vsys_p2p_ipc_open(const void* handle64, void** ptr) {
  if (!handle64 || !ptr) return VSYS_ERR_ARG;
  hipIpcMemHandle_t h;
  __builtin_memcpy(&h, handle64, 64);
  void* p = nullptr;
  if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess || !p) return VSYS_ERR_LAUNCH;
  *ptr = p;
  return 0;
}

int vsys_p2p_ipc_close(void* ptr) { return (!ptr || hipIpcCloseMemHandle(ptr) == hipSuccess) ? 0 : VSYS_ERR_LAUNCH; }

int vsys_conv_bf16(const void* a, int64_t lda, const void* w, int64_t ldw, const void* bias, const void* res, int64_t ldr,
                   void* out, void* out_f32, int64_t ldo, int64_t M, int64_t N, int64_t cin, int64_t kt, int64_t kh, int64_t kw,
                   int64_t row_pitch, int64_t plane_pitch, int64_t batch, int64_t batch_a, int64_t batch_w, int64_t batch_o,
                   float out_scale, void* stream) {
  if (!a || !w || ((out == nullptr) == (out_f32 == nullptr))) return VSYS_ERR_ARG;
  if (out_f32 && (bias || res)) return VSYS_ERR_ARG;
  if (!fits_int(M) || !fits_int(N) || !fits_int(cin) || !fits_int(row_pitch) || !fits_int(plane_pitch) || !fits_int(batch))
    return VSYS_ERR_SHAPE;
  if (kt < 1 || kt > 3 || kh < 1 || kh > 3 || kw < 1 || kw > 3 || kh != kw || cin <= 0) return VSYS_ERR_SHAPE;
  ConvParams p;
  p.A = B16(a); p.lda = lda; p.W = B16(w); p.ldw = ldw; p.bias = B16(bias); p.res = B16(res); p.ldr = ldr;
  p.out = B16(out); p.out32 = reinterpret_cast<float*>(out_f32); p.ldo = ldo;
  p.M = (int)M; p.N = (int)N; p.cin = (int)cin; p.taps = (int)(kt * kh * kw);
  const int64_t K = cin * p.taps;
  if (!fits_int(K)) return VSYS_ERR_SHAPE;
  p.K = (int)K;
  p.taps_hw = (int)(kh * kw); p.kw = (int)kw; p.row_pitch = (int)row_pitch; p.plane_pitch = (int)plane_pitch;
  const int64_t mtr = (kt - 1) * plane_pitch + (kh - 1) * row_pitch + (kw - 1);
  if (!fits_int(mtr)) return VSYS_ERR_SHAPE;
  p.max_tap_rows = (int)mtr;
  if (p.taps == 1) {
    p.cshift = 30;  // k-tile index never reaches the next tap
  } else {
    int sh = 0;
    while ((32 << sh) < cin) ++sh;
    p.cshift = sh;  // launch_conv checks cin == 32 << sh
  }
  p.batch = (int)batch; p.batch_a = batch_a; p.batch_w = batch_w; p.batch_o = batch_o; p.out_scale = out_scale;
  return launch_conv(p, S(stream));
}

int vsys_gn_stats(const void* x, const int64_t* grid, int64_t N, int64_t C, int64_t groups, float eps, void* partial_f32,
                  int64_t nblk, void* stats_f32, void* stream) {
  VaeGrid g;
  if (!x || !partial_f32 || !stats_f32 || !to_grid(grid, g)) return VSYS_ERR_ARG;
  if (!fits_int(N) || !fits_int(C) || !fits_int(groups) || !fits_int(nblk)) return VSYS_ERR_SHAPE;
  return launch_gn_stats(B16(x), g, (int)N, (int)C, (int)groups, eps, reinterpret_cast<float*>(partial_f32), (int)nblk,
                         reinterpret_cast<float*>(stats_f32), S(stream));
}

int vsys_gn_apply(const void* x, const int64_t* grid_src, void* y, const int64_t* grid_dst, int64_t N, int64_t C, int64_t groups,
                  const void* stats_f32, const void* gamma, const void* beta, int act, void* stream) {
  VaeGrid gs, gd;
  if (!x || !y || !stats_f32 || !gamma || !beta || !to_grid(grid_src, gs) || !to_grid(grid_dst, gd)) return VSYS_ERR_ARG;
  if (!fits_int(N) || !fits_int(C) || !fits_int(groups)) return VSYS_ERR_SHAPE;
  return launch_gn_apply(B16(x), gs, B16(y), gd, (int)N, (int)C, (int)groups, reinterpret_cast<const float*>(stats_f32), B16(gamma),
                         B16(beta), act, S(stream));
}

int vsys_regrid(const void* x, const int64_t* grid_src, void* y, const int64_t* grid_dst, int64_t N, int64_t C, int64_t up,
                int64_t tmode, void* stream) {
  VaeGrid gs, gd;
  if (!x || !y || !to_grid(grid_src, gs) || !to_grid(grid_dst, gd)) return VSYS_ERR_ARG;
  if (!fits_int(N) || !fits_int(C) || !fits_int(up) || !fits_int(tmode)) return VSYS_ERR_SHAPE;
  return launch_regrid(B16(x), gs, B16(y), gd, (int)N, (int)C, (int)up, (int)tmode, S(stream));
}

int vsys_subsample(const void* x, const int64_t* grid_src, void* y, const int64_t* grid_dst, int64_t N, int64_t C, int64_t t_stride,
                   int64_t s_stride, int64_t t_first, int64_t s_first, void* stream) {
  VaeGrid gs, gd;
  if (!x || !y || !to_grid(grid_src, gs) || !to_grid(grid_dst, gd)) return VSYS_ERR_ARG;
  if (!fits_int(N) || !fits_int(C) || !fits_int(t_stride) || !fits_int(s_stride) || !fits_int(t_first) || !fits_int(s_first))
    return VSYS_ERR_SHAPE;
  return launch_subsample(B16(x), gs, B16(y), gd, (int)N, (int)C, (int)t_stride, (int)s_stride, (int)t_first, (int)s_first, S(stream));
}

int vsys_spatial_norm_apply(const void* x, const int64_t* grid_src, void* y, const int64_t* grid_dst, int64_t N, int64_t C,
                            int64_t groups, const void* stats_f32, const void* gamma, const void* beta, const void* yb, int64_t zT,
                            int64_t zH, int64_t zW, void* stream) {
  VaeGrid gs, gd;
  if (!x || !y || !stats_f32 || !gamma || !beta || !yb || !to_grid(grid_src, gs) || !to_grid(grid_dst, gd)) return VSYS_ERR_ARG;
  if (!fits_int(N) || !fits_int(C) || !fits_int(groups) || !fits_int(zT) || !fits_int(zH) || !fits_int(zW)) return VSYS_ERR_SHAPE;
  return launch_spatial_norm_apply(B16(x), gs, B16(y), gd, (int)N, (int)C, (int)groups, reinterpret_cast<const float*>(stats_f32),
                                   B16(gamma), B16(beta), B16(yb), (int)zT, (int)zH, (int)zW, S(stream));
}

int vsys_blend_edge(const void* a, void* b, int64_t outer, int64_t Ha, int64_t Wa, int64_t Hb, int64_t Wb, int64_t ext, int64_t axis,
                    void* stream) {
  if (!a || !b) return VSYS_ERR_ARG;
  if (!fits_int(Ha) || !fits_int(Wa) || !fits_int(Hb) || !fits_int(Wb) || !fits_int(ext) || !fits_int(axis)) return VSYS_ERR_SHAPE;
  return launch_blend_edge(B16(a), B16(b), outer, (int)Ha, (int)Wa, (int)Hb, (int)Wb, (int)ext, (int)axis, S(stream));
}

int vsys_d2s_time(const void* x, const int64_t* grid_src, void* y, const int64_t* grid_dst, int64_t N, int64_t Cout, void* stream) {
  VaeGrid gs, gd;
  if (!x || !y || !to_grid(grid_src, gs) || !to_grid(grid_dst, gd)) return VSYS_ERR_ARG;
  if (!fits_int(N) || !fits_int(Cout)) return VSYS_ERR_SHAPE;
  return launch_d2s_time(B16(x), gs, B16(y), gd, (int)N, (int)Cout, S(stream));
}

int vsys_vae_first_im2col(const void* z, int64_t F, int64_t H, int64_t W, int64_t kt, int64_t kcols, const float* params,
                          void* out, void* stream) {
  if (!z || !out || !params) return VSYS_ERR_ARG;
  if (!fits_int(F) || !fits_int(H) || !fits_int(W) || !fits_int(kt) || !fits_int(kcols)) return VSYS_ERR_SHAPE;
  return launch_vae_first_im2col(B16(z), (int)F, (int)H, (int)W, (int)kt, (int)kcols, params, params + 4, params + 8, params + 24,
                                 B16(out), S(stream));
}

int vsys_extract_planar(const void* x, const int64_t* grid, int64_t N, int64_t ldx, int64_t nc, int64_t tskip, void* out,
                        int64_t Ftot, int64_t f0, void* stream) {
  VaeGrid g;
  if (!x || !out || !to_grid(grid, g)) return VSYS_ERR_ARG;
  if (!fits_int(N) || !fits_int(ldx) || !fits_int(nc) || !fits_int(tskip) || !fits_int(f0) || Ftot <= 0) return VSYS_ERR_SHAPE;
  return launch_extract_planar(B16(x), g, (int)N, (int)ldx, (int)nc, (int)tskip, B16(out), Ftot, (int)f0, S(stream));
}

int vsys_softmax_rows(const void* s_f32, void* p, int64_t rows, int64_t n, int64_t ld, void* stream) {
  if (!s_f32 || !p) return VSYS_ERR_ARG;
  if (!fits_int(n) || !fits_int(ld)) return VSYS_ERR_SHAPE;
  return launch_softmax_rows(reinterpret_cast<const float*>(s_f32), B16(p), rows, (int)n, (int)ld, S(stream));
}

}  // extern "C"
