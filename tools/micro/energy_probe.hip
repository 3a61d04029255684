// Where does the power of a bf16 GEMM go?  The GEMM kernels sit at the 1400 W socket cap, so throughput = cap /
// energy-per-flop.  This probe runs the GEMM's ingredients separately and together, each sustained for ~3 s, and prints the rate;
// sample `rocm-smi --showpower --showclocks` beside it (tools/micro/energy_probe.sh).
//   mode 0: MFMA only, operands toggling (four register sets in rotation)            -> the matrix-pipe ceiling
//   mode 1: + LDS fragment reads at the GEMM's ratio (7 ds_read_b128 per 12 MFMAs, the 128 x 96 wave tile)
//   mode 2: + LDS-DMA operand stream from an L2-resident buffer (0.0091 B/flop, the 256 x 192 tile)
//   mode 3: mode 2 with the stream coming from a 4 GB buffer (HBM)
//   mode 4: LDS reads at the 64 x 96 wave-tile ratio (5 per 6 MFMAs), no DMA
//   mode 5 / 6: mode 2 with 5 / 3 pieces per batch instead of 7 (0.0065 / 0.0039 B/flop: what a wider block tile would stream)
// hipcc --offload-arch=gfx950 -O3 energy_probe.hip -o energy_probe ; ./energy_probe <mode>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ inline unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }

// W16 (modes + 10): the same ingredients with v_mfma_f32_16x16x32_bf16 (24 accumulators of four registers, same flops per iteration):
// round 6 — the bare matrix pipe sustains 2.05 PFLOP/s with it against 1.83 for the 32x32x16 form at the same 1400 W cap.
template <int MODE, bool W16>
__global__ __launch_bounds__(256, 2) void probe(float* out, const char* __restrict__ stream_buf, size_t stream_bytes, int iters) {
#if __HIP_DEVICE_COMPILE__
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 64 KiB: fragment source + DMA landing zone
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x16 acc[W16 ? 1 : 12];
  f32x4 acc4[W16 ? 24 : 1];
  for (int i = 0; i < (W16 ? 1 : 12); ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int i = 0; i < (W16 ? 24 : 1); ++i)
    for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
  bf16x8 a[4], b[4];
  unsigned s = tid * 2654435761u + blockIdx.x;
  for (int k = 0; k < 4; ++k)
    for (int e = 0; e < 8; ++e) {
      a[k][e] = (__bf16)(((int)(lcg(s) >> 16) - 32768) * (1.0f / 32768.0f));
      b[k][e] = (__bf16)(((int)(lcg(s) >> 16) - 32768) * (1.0f / 32768.0f));
    }
  // fill LDS with pseudo-random bf16 so the fragment reads toggle like real operands
  for (int i = tid; i < 65536 / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = lcg(s) & 0x3f7f3f7fu;
  __syncthreads();
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)stream_buf, 0, (int)(stream_bytes < 0x7fffffff ? stream_bytes : 0x7fffffff), 0x00020000);
  // per-block stream position: each iteration of 12 MFMAs (393216 flops per wave) needs 0.0091 B/flop * 4 waves ~ 14 KB per block,
  // i.e. 3.5 one-KiB pieces per wave; modes 2 / 3 issue 7 pieces every second iteration
  unsigned pos = (blockIdx.x * 9973u) % 4096u;
  const unsigned span_kib = (unsigned)(stream_bytes >> 10);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 1 || MODE == 2 || MODE == 3 || MODE == 5 || MODE == 6) {
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] = *reinterpret_cast<const bf16x8*>(smem + ((it * 7 + k) & 63) * 1024 + lane * 16);
#pragma unroll
      for (int k = 0; k < 3; ++k) b[k] = *reinterpret_cast<const bf16x8*>(smem + ((it * 5 + k + 17) & 63) * 1024 + lane * 16);
    }
    if (MODE == 4) {  // 64 x 96 wave tile: 5 reads per 6 MFMAs = 10 per 12
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] = *reinterpret_cast<const bf16x8*>(smem + ((it * 7 + k) & 63) * 1024 + lane * 16);
#pragma unroll
      for (int k = 0; k < 4; ++k) b[k] = *reinterpret_cast<const bf16x8*>(smem + ((it * 5 + k + 17) & 63) * 1024 + lane * 16);
      bf16x8 t0 = *reinterpret_cast<const bf16x8*>(smem + ((it * 3 + 40) & 63) * 1024 + lane * 16);
      bf16x8 t1 = *reinterpret_cast<const bf16x8*>(smem + ((it * 3 + 41) & 63) * 1024 + lane * 16);
      a[0][0] += t0[0]; b[0][0] += t1[0];
    }
    constexpr int NP = MODE == 5 ? 5 : (MODE == 6 ? 3 : 7);
    if ((MODE == 2 || MODE == 3 || MODE == 5 || MODE == 6) && (it & 1)) {
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const unsigned piece = (pos + (unsigned)(wave * 7 + k)) % span_kib;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(smem + 32768 + ((wave * 7 + k) & 31) * 1024), 16, lane * 16,
                                                 piece * 1024, 0, 0);
      }
      pos += 28 * 37;
      if constexpr (NP == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else if constexpr (NP == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    }
    if constexpr (W16) {
#pragma unroll
      for (int i = 0; i < 24; ++i) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i + (i >> 2)) & 3], acc4[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i + (i >> 2)) & 3], acc[i], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float t = 0.f;
  for (int i = 0; i < (W16 ? 1 : 12); ++i)
    for (int r = 0; r < 16; ++r) t += acc[i][r];
  for (int i = 0; i < (W16 ? 24 : 1); ++i)
    for (int r = 0; r < 4; ++r) t += acc4[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
#endif
}

template <int MODE, bool W16 = false>
void run(const char* buf, size_t bytes) {
  const int blocks = 512, iters = 10000, launches = 150;
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipFuncSetAttribute((const void*)probe<MODE, W16>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int seg = 0; seg < 3; ++seg) {
    hipEventRecord(e0);
    for (int l = 0; l < launches; ++l) probe<MODE, W16><<<blocks, 256, 65536>>>(out, buf, bytes, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 32 * 32 * 16 * 12.0 * iters * 4 * blocks * launches;
    printf("{\"mfma\": \"%s\", \"mode\": %d, \"segment\": %d, \"seconds\": %.2f, \"tflops\": %.1f}\n", W16 ? "16x16x32" : "32x32x16", MODE, seg, ms / 1e3, flops / ms / 1e9);
    fflush(stdout);
  }
  hipFree(out);
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const size_t bytes = mode % 10 == 3 ? (size_t)1900 << 20 : (size_t)8 << 20;  // < 2 GiB: buffer offsets are 32-bit
  char* buf;
  hipMalloc(&buf, bytes);
  hipMemset(buf, 0x3c, bytes);
  switch (mode) {
    case 0: run<0>(buf, bytes); break;
    case 1: run<1>(buf, bytes); break;
    case 2: run<2>(buf, bytes); break;
    case 3: run<3>(buf, bytes); break;
    case 5: run<5>(buf, bytes); break;
    case 6: run<6>(buf, bytes); break;
    case 10: run<0, true>(buf, bytes); break;
    case 11: run<1, true>(buf, bytes); break;
    case 12: run<2, true>(buf, bytes); break;
    case 13: run<3, true>(buf, bytes); break;
    case 14: run<4, true>(buf, bytes); break;
    default: run<4>(buf, bytes); break;
  }
  return 0;
}
