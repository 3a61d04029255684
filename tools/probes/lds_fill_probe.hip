// Measurement tool (not part of the library): how fast can ONE CU bring bytes from L2 / HBM into LDS, by path?
//   dma     : buffer_load_dwordx4 ... lds (1 KiB per wave-instruction), counted vmcnt waits          — what every GEMM / attention kernel here uses
//   classic : global_load_dwordx4 -> VGPR -> ds_write_b128 (8 loads in flight per wave)
//   regs    : global_load_dwordx4 -> VGPR only (the vector-memory path without the LDS write)
// One workgroup per CU (256 workgroups), W waves each, every wave streams its own contiguous slice of a FOOT-byte window that is
// re-read ITERS times (FOOT small -> L2 hits, large -> HBM).  Prints GB/s per CU and chip-wide.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/lds_fill_probe tools/probes/lds_fill_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// each wave: `chunks` pieces of 1 KiB per pass over its slice
template <int MODE>
__global__ __launch_bounds__(512) void fill_kernel(const char* __restrict__ src, long slice_bytes, int passes, float* sink) {
#if __HIP_DEVICE_COMPILE__
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const long wg_slice = slice_bytes;                       // per workgroup
  const long wave_slice = wg_slice / nw;                    // per wave, multiple of 8 KiB
  const char* base = src + (long)blockIdx.x * wg_slice + (long)wave_u * wave_slice;
  char* my_lds = smem + wave_u * 8192;                      // 8 KiB ring per wave
  const int npieces = (int)(wave_slice / 1024);
  float acc = 0.f;
  if (MODE == 0) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)wave_slice, 0x00020000);
    for (int p = 0; p < passes; ++p) {
      for (int i = 0; i < npieces; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(my_lds + j * 1024), 16, lane * 16, (i + j) * 1024, 0, 0);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // half the ring may stay in flight
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc = *reinterpret_cast<float*>(my_lds + lane * 4);
  } else {
    for (int p = 0; p < passes; ++p) {
      for (int i = 0; i < npieces; i += 8) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const uint4*>(base + (long)(i + j) * 1024 + lane * 16);
        if (MODE == 1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) *reinterpret_cast<uint4*>(my_lds + j * 1024 + lane * 16) = v[j];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) acc += __uint_as_float(v[j].x ^ v[j].y ^ v[j].z ^ v[j].w);
        }
      }
    }
    if (MODE == 1) { __builtin_amdgcn_s_waitcnt(0); acc = *reinterpret_cast<float*>(my_lds + lane * 4); }
  }
  if (acc == 123.456f) sink[0] = acc;
#endif
}

int main(int argc, char** argv) {
  const int ncu = 256;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float* sink; CHECK(hipMalloc(&sink, 4));
  const long footprints[] = {16L << 20, 128L << 20, 1024L << 20};   // whole-chip window: 16 MB (fits the 8 x 4 MB L2s), 128 MB (Infinity Cache), 1 GB (HBM)
  const char* fname[] = {"16MB(L2)", "128MB(MALL)", "1GB(HBM)"};
  const char* mname[] = {"dma", "classic", "regs"};
  char* src; CHECK(hipMalloc(&src, 1024L << 20)); CHECK(hipMemset(src, 1, 1024L << 20));
  for (int f = 0; f < 3; ++f) {
    for (int waves = 4; waves <= 8; waves += 4) {
      for (int mode = 0; mode < 3; ++mode) {
        const long slice = footprints[f] / ncu;            // per workgroup
        const long total_target = 8L << 30;                  // ~8 GB of traffic per measurement
        int passes = (int)(total_target / footprints[f]); if (passes < 1) passes = 1;
        const size_t lds = (size_t)waves * 8192;
        auto launch = [&]() {
          if (mode == 0) hipLaunchKernelGGL(fill_kernel<0>, dim3(ncu), dim3(waves * 64), lds, 0, src, slice, passes, sink);
          else if (mode == 1) hipLaunchKernelGGL(fill_kernel<1>, dim3(ncu), dim3(waves * 64), lds, 0, src, slice, passes, sink);
          else hipLaunchKernelGGL(fill_kernel<2>, dim3(ncu), dim3(waves * 64), lds, 0, src, slice, passes, sink);
        };
        launch(); CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
          CHECK(hipEventRecord(a)); launch(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
          float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        const double bytes = (double)footprints[f] * passes;
        printf("{\"window\": \"%s\", \"waves_per_cu\": %d, \"path\": \"%s\", \"ms\": %.3f, \"chip_tb_s\": %.2f, \"gb_s_per_cu\": %.1f}\n", fname[f], waves,
               mname[mode], best, bytes / best / 1e9, bytes / best / 1e6 / ncu);
        fflush(stdout);
      }
    }
  }
  return 0;
}
