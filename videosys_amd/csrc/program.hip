// Launch programs (include/videosys_amd.h "Launch programs"): replay of a recorded sequence of C-ABI entry points in one call.
// Host code only; every case forwards to the SAME extern "C" function a direct call would use, so argument validation, kernel
// selection and error codes are identical to issuing the launches one by one.
#include "vsys_internal.h"

namespace {
struct OpInfo { int n_int, n_float; };
// integer / float argument counts of each op = its C signature without the trailing stream: GENERATED from the prototypes of
// include/videosys_amd.h (csrc/gen/program_gen.py -> program_ops.inc; tests/test_host_cpu.py keeps it fresh)
#define VSYS_PROGRAM_TABLE
#include "program_ops.inc"
#undef VSYS_PROGRAM_TABLE
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
      switch (c.op) {
#define VSYS_PROGRAM_CASES
#include "program_ops.inc"
#undef VSYS_PROGRAM_CASES
        default: rc = VSYS_ERR_ARG;
      }
    }
    if (rc != 0) {
      if (failed_at) *failed_at = i;
      return rc;
    }
  }
  return 0;
}

}  // extern "C"
