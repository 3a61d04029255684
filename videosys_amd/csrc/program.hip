// Launch programs (include/videosys_amd.h "Launch programs"): replay of a recorded sequence of C-ABI entry points in one call.
// Host code only; every case forwards to the SAME extern "C" function a direct call would use, so argument validation, kernel
// selection and error codes are identical to issuing the launches one by one.
#include "vsys_internal.h"

namespace {
struct OpInfo { int n_int, n_float; };
// integer / float argument counts of each op = its C signature without the trailing stream (checked against the ctypes table
// of the host mirror in tests/test_host_cpu.py)
constexpr OpInfo kOps[VSYS_OP_COUNT] = {
    {0, 0},
    {18, 0},  // GEMM_BF16
    {12, 0},  // LINEAR_SMALL
    {8, 1},   // ADALN_MODULATE
    {6, 0},   // MOD_TABLE
    {4, 0},   // TIMESTEP_EMBEDDING
    {14, 0},  // PATCH_EMBED
    {16, 1},  // FINAL_LAYER
    {6, 2},   // CFG_EULER_STEP
    {3, 0},   // ADD_ROWS
    {4, 0},   // COPY_4D_BATCH
    {11, 1},  // ATTN_PREP_KV
    {12, 1},  // FLASH_ATTN_D72
    {13, 1},  // ATTN_TEMPORAL_D72
    {6, 0},   // ADD_BCAST_ROWS
    {19, 0},  // GEMM_BF16_GATE2
    {12, 1},  // LN_MODULATE
    {9, 0},   // GATE_ADD_ROWS
    {16, 1},  // ATTN_PREP_KV64
    {17, 1},  // FLASH_ATTN_D64
    {16, 0},  // PATCH_EMBED_SHARD
    {11, 1},  // FINAL_LAYER_TOKENS
    {13, 0},  // UNPATCHIFY_TOKENS
    {14, 1},  // GEMM_BF16_LN
    {17, 0},  // GEMM_BF16_STATS
    {4, 0},   // ADALN_PRESCALE
    {5, 0},   // LN_ROW_STATS
    {20, 0},  // GEMM_BF16_GATE_RES_ADD
    {12, 2},  // FLASH_ATTN_D72_KB
    {17, 2},  // FLASH_ATTN_D64_KB
    {12, 1},  // FLASH_ATTN_D72_EXACT
    {8, 0},   // P2P_EXCHANGE
};
}  // namespace

extern "C" {

int vsys_program_op_info(int op, int* n_int, int* n_float) {
  if (op <= 0 || op >= VSYS_OP_COUNT || !n_int || !n_float) return VSYS_ERR_ARG;
  *n_int = kOps[op].n_int;
  *n_float = kOps[op].n_float;
  return 0;
}

int vsys_program_run(const vsys_cmd* cmds, int64_t n, void* const* streams, int64_t n_streams, int64_t* failed_at) {
  if (n < 0 || (n > 0 && (!cmds || !streams)) || n_streams < 0) return VSYS_ERR_ARG;
  for (int64_t i = 0; i < n; ++i) {
    const vsys_cmd& c = cmds[i];
    int rc = VSYS_ERR_ARG;
    if (c.stream >= 0 && c.stream < n_streams) {
      void* st = streams[c.stream];
#define I(k) c.a[k]
#define N32(k) static_cast<int>(c.a[k])
#define P(k) reinterpret_cast<void*>(c.a[k])
#define CP(k) reinterpret_cast<const void*>(c.a[k])
      switch (c.op) {
        case VSYS_OP_GEMM_BF16:
          rc = vsys_gemm_bf16(CP(0), I(1), CP(2), I(3), CP(4), P(5), I(6), I(7), I(8), I(9), N32(10), CP(11), I(12), I(13), CP(14), I(15),
                              P(16), I(17), st);
          break;
        case VSYS_OP_LINEAR_SMALL:
          rc = vsys_linear_small(CP(0), I(1), CP(2), I(3), CP(4), P(5), I(6), I(7), I(8), I(9), N32(10), N32(11), st);
          break;
        case VSYS_OP_ADALN_MODULATE:
          rc = vsys_adaln_modulate(CP(0), CP(1), CP(2), P(3), I(4), I(5), I(6), I(7), c.f[0], st);
          break;
        case VSYS_OP_MOD_TABLE: rc = vsys_mod_table(CP(0), CP(1), P(2), I(3), I(4), I(5), st); break;
        case VSYS_OP_TIMESTEP_EMBEDDING: rc = vsys_timestep_embedding(CP(0), P(1), I(2), I(3), st); break;
        case VSYS_OP_PATCH_EMBED:
          rc = vsys_patch_embed(CP(0), I(1), CP(2), CP(3), CP(4), P(5), I(6), I(7), I(8), I(9), I(10), I(11), I(12), I(13), st);
          break;
        case VSYS_OP_FINAL_LAYER:
          rc = vsys_final_layer(CP(0), CP(1), CP(2), CP(3), CP(4), P(5), I(6), I(7), I(8), I(9), I(10), I(11), I(12), I(13), I(14), I(15),
                                c.f[0], st);
          break;
        case VSYS_OP_CFG_EULER_STEP: rc = vsys_cfg_euler_step(P(0), CP(1), I(2), I(3), I(4), I(5), c.f[0], c.f[1], st); break;
        case VSYS_OP_ADD_ROWS: rc = vsys_add_rows(P(0), CP(1), I(2), st); break;
        case VSYS_OP_COPY_4D_BATCH: rc = vsys_copy_4d_batch(CP(0), P(1), I(2), reinterpret_cast<const int64_t*>(c.a[3]), st); break;
        case VSYS_OP_ATTN_PREP_KV:
          rc = vsys_attn_prep_kv(CP(0), I(1), CP(2), I(3), CP(4), P(5), P(6), I(7), I(8), I(9), I(10), c.f[0], st);
          break;
        case VSYS_OP_FLASH_ATTN_D72:
          rc = vsys_flash_attn_d72(CP(0), I(1), CP(2), CP(3), CP(4), P(5), I(6), I(7), I(8), I(9), I(10), I(11), c.f[0], st);
          break;
        case VSYS_OP_ATTN_TEMPORAL_D72:
          rc = vsys_attn_temporal_d72(CP(0), I(1), I(2), CP(3), CP(4), CP(5), CP(6), P(7), I(8), I(9), I(10), I(11), I(12), c.f[0], st);
          break;
        case VSYS_OP_ADD_BCAST_ROWS: rc = vsys_add_bcast_rows(P(0), CP(1), I(2), I(3), I(4), I(5), st); break;
        case VSYS_OP_GEMM_BF16_GATE2:
          rc = vsys_gemm_bf16_gate2(CP(0), I(1), CP(2), I(3), CP(4), P(5), I(6), I(7), I(8), I(9), CP(10), I(11), I(12), I(13), I(14),
                                    CP(15), I(16), P(17), I(18), st);
          break;
        case VSYS_OP_LN_MODULATE:
          rc = vsys_ln_modulate(CP(0), CP(1), CP(2), CP(3), CP(4), P(5), I(6), I(7), I(8), I(9), I(10), I(11), c.f[0], st);
          break;
        case VSYS_OP_GATE_ADD_ROWS: rc = vsys_gate_add_rows(P(0), CP(1), CP(2), I(3), I(4), I(5), I(6), I(7), I(8), st); break;
        case VSYS_OP_ATTN_PREP_KV64:
          rc = vsys_attn_prep_kv64(CP(0), I(1), CP(2), I(3), CP(4), CP(5), CP(6), CP(7), I(8), I(9), P(10), P(11), I(12), I(13), I(14),
                                   I(15), c.f[0], st);
          break;
        case VSYS_OP_FLASH_ATTN_D64:
          rc = vsys_flash_attn_d64(CP(0), I(1), CP(2), CP(3), CP(4), CP(5), I(6), I(7), CP(8), CP(9), P(10), I(11), I(12), I(13), I(14),
                                   I(15), I(16), c.f[0], st);
          break;
        case VSYS_OP_PATCH_EMBED_SHARD:
          rc = vsys_patch_embed_shard(CP(0), I(1), CP(2), CP(3), CP(4), P(5), I(6), I(7), I(8), I(9), I(10), I(11), I(12), I(13), I(14),
                                      I(15), st);
          break;
        case VSYS_OP_FINAL_LAYER_TOKENS:
          rc = vsys_final_layer_tokens(CP(0), CP(1), CP(2), CP(3), CP(4), P(5), I(6), I(7), I(8), I(9), I(10), c.f[0], st);
          break;
        case VSYS_OP_UNPATCHIFY_TOKENS:
          rc = vsys_unpatchify_tokens(CP(0), P(1), I(2), I(3), I(4), I(5), I(6), I(7), I(8), I(9), I(10), I(11), I(12), st);
          break;
        case VSYS_OP_GEMM_BF16_LN:
          rc = vsys_gemm_bf16_ln(CP(0), I(1), CP(2), I(3), CP(4), CP(5), P(6), I(7), I(8), I(9), I(10), N32(11), CP(12), I(13), c.f[0], st);
          break;
        case VSYS_OP_GEMM_BF16_STATS:
          rc = vsys_gemm_bf16_stats(CP(0), I(1), CP(2), I(3), CP(4), P(5), I(6), I(7), I(8), I(9), CP(10), I(11), I(12), CP(13), I(14),
                                    P(15), I(16), st);
          break;
        case VSYS_OP_ADALN_PRESCALE: rc = vsys_adaln_prescale(CP(0), I(1), I(2), CP(3), st); break;
        case VSYS_OP_LN_ROW_STATS: rc = vsys_ln_row_stats(CP(0), I(1), I(2), P(3), I(4), st); break;
        case VSYS_OP_FLASH_ATTN_D64_KB:
          rc = vsys_flash_attn_d64_kb(CP(0), I(1), CP(2), CP(3), CP(4), CP(5), I(6), I(7), CP(8), CP(9), P(10), I(11), I(12), I(13), I(14),
                                      I(15), I(16), c.f[0], c.f[1], st);
          break;
        case VSYS_OP_FLASH_ATTN_D72_EXACT:
          rc = vsys_flash_attn_d72_exact(CP(0), I(1), CP(2), CP(3), CP(4), P(5), I(6), I(7), I(8), I(9), I(10), I(11), c.f[0], st);
          break;
        case VSYS_OP_P2P_EXCHANGE:
          rc = vsys_p2p_exchange(CP(0), I(1), reinterpret_cast<const int64_t*>(c.a[2]), CP(3), I(4), I(5), P(6), I(7), st);
          break;
        case VSYS_OP_FLASH_ATTN_D72_KB:
          rc = vsys_flash_attn_d72_kb(CP(0), I(1), CP(2), CP(3), CP(4), P(5), I(6), I(7), I(8), I(9), I(10), I(11), c.f[0], c.f[1], st);
          break;
        case VSYS_OP_GEMM_BF16_GATE_RES_ADD:
          rc = vsys_gemm_bf16_gate_res_add(CP(0), I(1), CP(2), I(3), CP(4), P(5), I(6), I(7), I(8), I(9), CP(10), I(11), I(12), CP(13),
                                           I(14), P(15), CP(16), CP(17), P(18), I(19), st);
          break;
        default: rc = VSYS_ERR_ARG;
      }
#undef I
#undef N32
#undef P
#undef CP
    }
    if (rc != 0) {
      if (failed_at) *failed_at = i;
      return rc;
    }
  }
  return 0;
}

}  // extern "C"
