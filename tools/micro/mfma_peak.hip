// Sustained v_mfma_f32_32x32x16_bf16 rate with no memory traffic: the practical MFMA ceiling (clock under matrix load) that the
// GEMM's 2.5 PFLOP/s spec-sheet roofline should be read against.   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ __launch_bounds__(512, 2) void mfma_loop(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(e * 0.5f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks, int threads, int iters) {
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_loop<NACC><<<blocks, threads>>>(out, 100);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    mfma_loop<NACC><<<blocks, threads>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double flops = 2.0 * 32 * 32 * 16 * (double)NACC * iters * (threads / 64) * blocks;
  printf("{\"kernel\": \"mfma_f32_32x32x16_bf16 x%d accumulators\", \"blocks\": %d, \"waves_per_block\": %d, \"ms\": %.3f, \"tflops\": %.1f}\n",
         NACC, blocks, threads / 64, best, flops / best / 1e9);
  hipFree(out);
}

int main() {
  run<4>(256, 256, 20000);    // 1 wave per SIMD
  run<4>(256, 512, 20000);    // 2 waves per SIMD
  run<6>(256, 512, 20000);
  run<12>(512, 256, 10000);   // gemm2 shape: 12 accumulators, 2 workgroups of 4 waves
  return 0;
}
