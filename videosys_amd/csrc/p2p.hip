// One-kernel peer-to-peer layout switch of Dynamic Sequence Parallelism (xGMI is point-to-point: every GPU has a direct link to each
// of its seven peers, and a store to IPC-mapped peer memory travels over it).
//
// Replaces, at the two exchange sites of every spatial block (/root/reference/videosys):
//   core/distributed/comm.py:104-141        _all_to_all_func: tensor_split + P x contiguous + list all_to_all + cat + contiguous
//   core/distributed/comm.py:282-304        all_to_all_with_pad: F.pad before, narrow after
//   models/transformers/open_sora_transformer_3d.py:288-315   dynamic_switch
// and this repo's own RCCL-shaped path (pack kernel -> all_to_all_single -> unpack kernel: dsp.py), which stays as the fallback.
//
// One launch per rank and exchange does everything: problem r of the batch copies the rows this rank owes peer r straight from the
// source tensor (pack strides, zero fill for the padded frames) into peer r's DESTINATION tensor in its final layout (unpack
// strides, narrowed) — no send buffer, no receive buffer, no second pass.  When the stores of every problem have been acknowledged
// (write-through stores, drained per wave, counted per workgroup) the launch's last workgroup stores this exchange's sequence number
// into flag[me] of every peer and then waits (one lane, s_sleep between polls, wall-clock timeout) until every peer's number has
// arrived in this rank's own flag array.
// The kernel therefore ends only when this rank's destination tensor is complete, and the consumer (the qkv GEMM, the projection
// GEMM) is simply the next launch on the stream.  The sequence number lives in device memory (state[0]) and is advanced by the
// kernel itself, so a recorded launch program replays the same command every step.
//
// Reuse of a destination tensor needs no acknowledgement: a peer reaches exchange site X again only after it has passed the other
// site of the same block, whose kernel waited for THIS rank's flag there — and this rank raised that flag after (in stream order) the
// consumers of X's previous data.  (dsp.py PeerExchange states the argument in full.)
//
// Flags live in fine-grained memory (vsys_p2p_alloc): a poll inside a running kernel must see a remote store, which the
// coarse-grained default does not promise before the next kernel boundary.  Payload tensors are ordinary (coarse-grained)
// allocations: they are only ever read by LATER kernels, whose start invalidates the XCD L2s.
#include "common.h"
#include "vsys_internal.h"

namespace vsys {

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct P2PDesc {
  CopyDesc c;            // c.dst_off is relative to this problem's own destination pointer
  bf16_t* dst;           // peer's (or this rank's own) destination tensor
  unsigned* peer_flag;   // &flags_of_peer[me]; null for the rank's own problem
  int remote;            // 1: the destination is read by ANOTHER process / device right after the flag (write-through stores);
                         // 0: by later launches ordered on this device (the rank itself, ranks that are threads of this process)
};
struct P2PBatch {
  int nops;
  P2PDesc d[VSYS_COPY_BATCH_MAX];
};

// state (uint32, one 128-byte line per word so that the counters of different problems do not share an L2 channel):
//   word 0 = sequence number of the last finished exchange of this site, word 1 = problems finished in the running launch,
//   word 2 + i = workgroups of problem i finished, word 18 = error word (0 ok; 1 + q: peer q's flag did not arrive in time)
constexpr int ST = 32;                       // uint32 per word slot
constexpr int ST_WORDS = 19;
__global__ __launch_bounds__(256) void p2p_exchange_kernel(const bf16_t* __restrict__ src, P2PBatch b, const unsigned* my_flags, int n_flags,
                                                           int self_index, unsigned* state, long long timeout_ticks) {
  const P2PDesc& pd = b.d[blockIdx.y];
  const CopyDesc& o = pd.c;
  const unsigned seq = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  const int cch = o.C >> 3;
  const int64_t total = (int64_t)o.n0 * o.n1 * o.n2 * cch;
  const bf16_t* s = src + o.src_off;
  bf16_t* d = pd.dst + o.dst_off;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cch);
    int64_t r = i / cch;
    const int i2 = (int)(r % o.n2);
    r /= o.n2;
    const int i1 = (int)(r % o.n1);
    const int i0 = (int)(r / o.n1);
    u32x4 v = {0u, 0u, 0u, 0u};
    if (i1 < o.n1_valid && i2 < o.n2_valid) v = *reinterpret_cast<const u32x4*>(s + i0 * o.ss0 + i1 * o.ss1 + i2 * o.ss2 + c * 8);
    bf16_t* q = d + i0 * o.ds0 + i1 * o.ds1 + i2 * o.ds2 + c * 8;
    if (pd.remote) {
      // system-scope WRITE-THROUGH store: the rows go to the peer's memory past this GPU's L2, so publishing them needs no cache
      // write-back (a __threadfence_system() per thread — buffer_wbl2 in every wave — made this kernel 15x slower).  Only for
      // destinations another process / device reads: a local reader is ordered by the launch boundary anyway.
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(q), "v"(v) : "memory");
    } else {
      *reinterpret_cast<u32x4*>(q) = v;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every store of this wave has been acknowledged by its destination ...
  __syncthreads();                                   // ... and so have the other waves' of this workgroup
  if (threadIdx.x != 0) return;
  // The hand-off is the counter form of cdna_hip_programming.md section 6 Guideline 16: write-through payload, drained, THEN the
  // relaxed ticket.  No release fence anywhere: a fence is a buffer_wbl2 — with one per workgroup (2880 of them at config 2, all
  // on one counter line) this launch took 88 us for a 1.4 MB exchange; it is why the grid is at most 32 workgroups per problem
  // and every counter has a cache line to itself.
  if (__hip_atomic_fetch_add(&state[ST * (2 + blockIdx.y)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gridDim.x - 1) return;
  // ---- last workgroup of this problem: every store of the problem has been acknowledged
  __hip_atomic_store(&state[ST * (2 + blockIdx.y)], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (__hip_atomic_fetch_add(&state[ST * 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gridDim.y - 1) return;
  // ---- last problem of the launch: EVERY problem's stores have been acknowledged (several problems may feed one peer — the text
  // and video rows of the Ulysses gather — so a peer is told once, here), then wait for every peer's rows (their flag in MY array)
  __hip_atomic_store(&state[ST * 1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int i = 0; i < b.nops; ++i)
    if (b.d[i].peer_flag != nullptr) __hip_atomic_store(b.d[i].peer_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const long long t0 = wall_clock64();
  // (timeout_ticks < 0: the caller orders the peers' launches itself — ranks that are threads of one process rendezvous on the host;
  //  a site that has already timed out once does not wait again: the error word is sticky and the host raises at its next check)
  const bool wait = timeout_ticks >= 0 && __hip_atomic_load(&state[ST * 18], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
  for (int q = 0; wait && q < n_flags; ++q) {
    if (q == self_index) continue;
    while ((int)(__hip_atomic_load(my_flags + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {   // (polled relaxed: an acquire per poll would drop the L1 every time)
      __builtin_amdgcn_s_sleep(16);
      if (timeout_ticks > 0 && wall_clock64() - t0 > timeout_ticks) {
        __hip_atomic_store(&state[ST * 18], (unsigned)(1 + q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
  __hip_atomic_store(state, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the next launch of this site reads it: launch boundary)
}

}  // namespace

int launch_p2p_exchange(const bf16_t* src, const CopyDesc* ops, bf16_t* const* dsts, unsigned* const* peer_flags, const int* remote, int nops,
                        const unsigned* my_flags, int n_flags, int self_index, unsigned* state, long long timeout_ticks,
                        hipStream_t stream) {
  if (nops <= 0) return 0;
  if (nops > VSYS_COPY_BATCH_MAX) return VSYS_ERR_SHAPE;
  P2PBatch b;
  b.nops = nops;
  int64_t most = 0;
  for (int i = 0; i < nops; ++i) {
    if (ops[i].C % 8 || ops[i].n0 < 0 || ops[i].n1 < 0 || ops[i].n2 < 0 || dsts[i] == nullptr) return VSYS_ERR_SHAPE;
    b.d[i].c = ops[i];
    b.d[i].dst = dsts[i];
    b.d[i].peer_flag = peer_flags[i];
    b.d[i].remote = remote[i];
    const int64_t t = (int64_t)ops[i].n0 * ops[i].n1 * ops[i].n2 * (ops[i].C / 8);
    most = t > most ? t : most;
  }
  // (an exchange whose every problem is empty still signals and waits: the peers count on this rank's flag)
  int64_t grid = (most + 255) / 256;
  grid = grid < 1 ? 1 : (grid > 32 ? 32 : grid);   // <= 32 workgroups per problem: 8 problems fill the chip, 256 tickets per launch
  hipLaunchKernelGGL(p2p_exchange_kernel, dim3((unsigned)grid, (unsigned)nops), dim3(256), 0, stream, src, b, my_flags, n_flags,
                     self_index, state, timeout_ticks);
  return hipGetLastError() == hipSuccess ? 0 : VSYS_ERR_LAUNCH;
}

}  // namespace vsys
