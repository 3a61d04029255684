// PyTorch custom-op registration of the C ABI: torch.ops.vsys.launch / torch.ops.vsys.program_run (TORCH_LIBRARY fragment "vsys").
//
// north_star: "Python host code calls these [kernels] through PyTorch-ROCm custom ops only"; SURVEY.md §8(b) proposes a TORCH_LIBRARY
// boundary whose ops take at::Tensor, run on the current HIP stream, never allocate, never sync the host and throw c10::Error on a
// device mismatch.  The reference has no native layer at all (every op is a torch call issued from Python every step,
// /root/reference/videosys/models/transformers/open_sora_transformer_3d.py:608-613), so this file replaces nothing there: it is the
// dispatcher-visible doorway to include/videosys_amd.h.
//
//   vsys::launch(int op, Tensor?[] tensors, int[] ints, float[] floats, int stream) -> ()
//       ONE entry point of the C ABI, named by its VSYS_OP_* code (include/videosys_amd.h "Launch programs").  ``ints`` are the entry
//       point's integer / pointer arguments in declaration order; where ``tensors[k]`` is defined, argument k is that tensor's data
//       pointer (checked: a HIP device tensor, on the device the stream belongs to); every other pointer argument travels as the
//       integer it is (a host descriptor array, a fine-grained flag array of vsys_p2p_alloc).  ``stream`` = the raw handle of torch's
//       current HIP stream (the Python side passes torch.cuda.current_stream().cuda_stream: this file needs no HIP headers).
//   vsys::program_run(Tensor cmds, int n, int[] streams) -> ()
//       a recorded launch program (videosys_amd/program.py): ``cmds`` is a CPU uint8 tensor over n vsys_cmd records, ``streams`` the
//       stream table (slot 0 = the stream current at replay time).  One dispatcher call per denoise step.
//
// Both forward to the SAME extern "C" functions a ctypes caller binds (vsys_program_run -> vsys_<entry>), so argument validation,
// kernel selection and error codes are identical; a non-zero code becomes a c10::Error naming the command that failed.
// Built by __graft_entry__.build() with the host compiler against the image's torch headers into videosys_amd/libvideosys_torch.so,
// loaded with torch.ops.load_library by videosys_amd/_lib.py (VSYS_TORCH_OPS=0: the ctypes route).
#include <ATen/ATen.h>
#include <torch/library.h>

#include <vector>

#include "../../include/videosys_amd.h"

namespace {

void check_rc(int rc, int64_t failed_at, int64_t op) {
  TORCH_CHECK(rc == 0, "vsys op ", op, " (command ", failed_at, "): ", vsys_strerror(rc), " (code ", rc, ")");
}

void vsys_launch(int64_t op, c10::List<c10::optional<at::Tensor>> tensors, at::IntArrayRef ints, at::ArrayRef<double> floats,
                 int64_t stream) {
  int n_int = 0, n_float = 0;
  TORCH_CHECK(vsys_program_op_info((int)op, &n_int, &n_float) == 0, "vsys::launch: unknown op code ", op);
  TORCH_CHECK((int64_t)ints.size() == n_int && (int64_t)floats.size() == n_float, "vsys::launch: op ", op, " takes ", n_int, " integer and ",
              n_float, " float arguments, got ", ints.size(), " and ", floats.size());
  TORCH_CHECK(tensors.size() <= ints.size(), "vsys::launch: more tensors than integer arguments");
  vsys_cmd cmd;
  cmd.op = (int32_t)op;
  cmd.stream = 0;
  for (int k = 0; k < VSYS_CMD_MAX_INT; ++k) cmd.a[k] = k < n_int ? ints[k] : 0;
  for (int k = 0; k < VSYS_CMD_MAX_FLOAT; ++k) cmd.f[k] = k < n_float ? (float)floats[k] : 0.f;
  int device = -1;
  for (size_t k = 0; k < tensors.size(); ++k) {
    const c10::optional<at::Tensor>& t = tensors.get(k);
    if (!t.has_value() || !t->defined()) continue;
    TORCH_CHECK(t->is_cuda(), "vsys::launch: argument ", k, " of op ", op, " is a ", t->device(), " tensor; the kernels take HIP device tensors (no CPU fallback)");
    TORCH_CHECK(device < 0 || t->get_device() == device, "vsys::launch: tensors of op ", op, " live on different devices");
    device = (int)t->get_device();
    cmd.a[k] = reinterpret_cast<int64_t>(t->data_ptr());
  }
  void* streams[1] = {reinterpret_cast<void*>(stream)};
  int64_t failed_at = -1;
  const int rc = vsys_program_run(&cmd, 1, streams, 1, &failed_at);
  check_rc(rc, failed_at, op);
}

void vsys_program_run_op(const at::Tensor& cmds, int64_t n, at::IntArrayRef streams) {
  TORCH_CHECK(cmds.device().is_cpu() && cmds.is_contiguous() && cmds.scalar_type() == at::kByte, "vsys::program_run: cmds is a contiguous CPU uint8 tensor");
  TORCH_CHECK(n >= 0 && (int64_t)cmds.numel() >= n * (int64_t)sizeof(vsys_cmd), "vsys::program_run: ", n, " commands do not fit ", cmds.numel(), " bytes");
  std::vector<void*> st(streams.size());
  for (size_t i = 0; i < streams.size(); ++i) st[i] = reinterpret_cast<void*>(streams[i]);
  int64_t failed_at = -1;
  const vsys_cmd* c = reinterpret_cast<const vsys_cmd*>(cmds.data_ptr());
  const int rc = vsys_program_run(c, n, st.data(), (int64_t)st.size(), &failed_at);
  check_rc(rc, failed_at, failed_at >= 0 && failed_at < n ? c[failed_at].op : -1);
}

}  // namespace

TORCH_LIBRARY_FRAGMENT(vsys, m) {
  m.def("launch(int op, Tensor?[] tensors, int[] ints, float[] floats, int stream) -> ()");
  m.def("program_run(Tensor cmds, int n, int[] streams) -> ()");
}
// (CompositeExplicitAutograd: one kernel for every backend key — the ops mutate caller-owned device memory named by the tensors and by
//  raw addresses, there is nothing to differentiate and nothing for a backend-specific dispatch to choose)
TORCH_LIBRARY_IMPL(vsys, CompositeExplicitAutograd, m) {
  m.impl("launch", &vsys_launch);
  m.impl("program_run", &vsys_program_run_op);
}
