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

// operands that CHANGE every MFMA (four pseudo-random register sets in rotation): switching activity of real data
__device__ inline unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
__global__ __launch_bounds__(256, 2) void mfma_toggle(float* out, int iters) {
  f32x16 acc[12];
  for (int i = 0; i < 12; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a[4], b[4];
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
  for (int k = 0; k < 4; ++k)
    for (int e = 0; e < 8; ++e) {
      a[k][e] = (__bf16)(((int)(lcg(s) >> 16) - 32768) * (1.0f / 32768.0f));
      b[k][e] = (__bf16)(((int)(lcg(s) >> 16) - 32768) * (1.0f / 32768.0f));
    }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i + (i >> 2)) & 3], acc[i], 0, 0, 0);
  }
  float t = 0.f;
  for (int i = 0; i < 12; ++i)
    for (int r = 0; r < 16; ++r) t += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}
void sustained_toggle(int blocks, int iters, int launches) {
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int seg = 0; seg < 4; ++seg) {
    hipEventRecord(e0);
    for (int l = 0; l < launches; ++l) mfma_toggle<<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 32 * 32 * 16 * 12.0 * iters * 4 * blocks * launches;
    printf("{\"kernel\": \"sustained, random operands rotating, segment %d\", \"seconds\": %.2f, \"tflops\": %.1f}\n", seg, ms / 1e3, flops / ms / 1e9);
  }
  hipFree(out);
}


// the same experiment with v_mfma_f32_16x16x32_bf16 (half the accumulator bytes per flop, twice the A / B operand bytes per flop of the
// 32x32x16 form): does the matrix pipe's energy per flop depend on the instruction shape?  24 accumulators of four registers.
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
__global__ __launch_bounds__(256, 2) void mfma_toggle16(float* out, int iters) {
  f32x4_t acc[24];
  for (int i = 0; i < 24; ++i)
    for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  bf16x8 a[4], b[4];
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
  for (int k = 0; k < 4; ++k)
    for (int e = 0; e < 8; ++e) {
      a[k][e] = (__bf16)(((int)(lcg(s) >> 16) - 32768) * (1.0f / 32768.0f));
      b[k][e] = (__bf16)(((int)(lcg(s) >> 16) - 32768) * (1.0f / 32768.0f));
    }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 24; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i + (i >> 2)) & 3], acc[i], 0, 0, 0);
  }
  float t = 0.f;
  for (int i = 0; i < 24; ++i)
    for (int r = 0; r < 4; ++r) t += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}
void sustained_toggle16(int blocks, int iters, int launches) {
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int seg = 0; seg < 4; ++seg) {
    hipEventRecord(e0);
    for (int l = 0; l < launches; ++l) mfma_toggle16<<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 16 * 16 * 32 * 24.0 * iters * 4 * blocks * launches;
    printf("{\"kernel\": \"sustained 16x16x32, random operands rotating, segment %d\", \"seconds\": %.2f, \"tflops\": %.1f}\n", seg, ms / 1e3, flops / ms / 1e9);
  }
  hipFree(out);
}

// sustained: back-to-back launches for ~seconds (the short runs above finish before power management reacts)
template <int NACC>
void sustained(int blocks, int threads, int iters, int launches) {
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int seg = 0; seg < 4; ++seg) {
    hipEventRecord(e0);
    for (int l = 0; l < launches; ++l) mfma_loop<NACC><<<blocks, threads>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 32 * 32 * 16 * (double)NACC * iters * (threads / 64) * blocks * launches;
    printf("{\"kernel\": \"sustained x%d accumulators segment %d\", \"seconds\": %.2f, \"tflops\": %.1f}\n", NACC, seg, ms / 1e3, flops / ms / 1e9);
  }
  hipFree(out);
}

int main(int argc, char** argv) {
  if (argc > 2) { sustained_toggle16(512, 10000, 300); return 0; }
  if (argc > 1) { sustained_toggle(512, 10000, 300); return 0; }
  sustained<12>(512, 256, 10000, 300);
  run<4>(256, 256, 20000);    // 1 wave per SIMD
  run<4>(256, 512, 20000);    // 2 waves per SIMD
  run<6>(256, 512, 20000);
  run<12>(512, 256, 10000);   // gemm2 shape: 12 accumulators, 2 workgroups of 4 waves
  return 0;
}
