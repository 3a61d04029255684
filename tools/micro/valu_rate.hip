// Issue cost of the softmax's VALU instructions on one gfx950 SIMD, alone and in the gaps of a v_mfma_f32_32x32x16_bf16 stream:
// the numbers behind DESIGN.md §3.2's "what bounds the flash kernels" paragraph.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
// One workgroup per CU, W waves per SIMD (W = 1, 2, 3); every wave runs the same straight-line block of 256 instructions (or 64 MFMAs
// with k fillers behind each) ITERS times between two s_memtime reads; reported: SIMD cycles per instruction (per MFMA + its fillers) =
// (last wave's end - first wave's start) / (units per wave x W), and the same per unit for the fastest wave (the oldest wave of a SIMD
// wins every arbitration, so one wave's own clock says nothing about the SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

template <int MODE>
__global__ void probe(unsigned long long* out, int iters) {
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = -1.0f - 0.01f * (threadIdx.x + i);
  typedef __attribute__((ext_vector_type(16))) float f32x16;
  typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
  f32x16 acc[2];
  for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (threadIdx.x & 15) + e); b[e] = (__bf16)(0.25f * e); }
  unsigned p0, p1, p2, p3;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#define V8(op)                                                                                                     \
  asm volatile(REP32(op " %0, %0\n\t" op " %1, %1\n\t" op " %2, %2\n\t" op " %3, %3\n\t" op " %4, %4\n\t" op " %5, %5\n\t" op " %6, %6\n\t" op " %7, %7\n\t") \
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]))
#define V8B(op)                                                                                                    \
  asm volatile(REP32(op " %0, %0, %0\n\t" op " %1, %1, %1\n\t" op " %2, %2, %2\n\t" op " %3, %3, %3\n\t" op " %4, %4, %4\n\t" op " %5, %5, %5\n\t" op " %6, %6, %6\n\t" op " %7, %7, %7\n\t") \
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]))
    if (MODE == 0) V8("v_exp_f32");
    if (MODE == 1) V8B("v_mul_f32");
    if (MODE == 2) {
      asm volatile(REP32("v_cvt_pk_bf16_f32 %8, %0, %1\n\tv_cvt_pk_bf16_f32 %9, %2, %3\n\tv_cvt_pk_bf16_f32 %10, %4, %5\n\tv_cvt_pk_bf16_f32 %11, %6, %7\n\t"
                         "v_cvt_pk_bf16_f32 %8, %1, %2\n\tv_cvt_pk_bf16_f32 %9, %3, %4\n\tv_cvt_pk_bf16_f32 %10, %5, %6\n\tv_cvt_pk_bf16_f32 %11, %7, %0\n\t")
                   : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3));
    }
    if (MODE == 3) {
      asm volatile(REP32("v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %1, %1, %2, %3\n\tv_max3_f32 %2, %2, %3, %4\n\tv_max3_f32 %3, %3, %4, %5\n\t"
                         "v_max3_f32 %4, %4, %5, %6\n\tv_max3_f32 %5, %5, %6, %7\n\tv_max3_f32 %6, %6, %7, %0\n\tv_max3_f32 %7, %7, %0, %1\n\t")
                   : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
    }
    if (MODE == 9) {   // a DEPENDENT max chain (what a row max is): latency per instruction
      asm volatile(REP32("v_max_f32 %0, %0, %1\n\tv_max_f32 %0, %0, %2\n\tv_max_f32 %0, %0, %3\n\tv_max_f32 %0, %0, %4\n\t"
                         "v_max_f32 %0, %0, %5\n\tv_max_f32 %0, %0, %6\n\tv_max_f32 %0, %0, %7\n\tv_max_f32 %0, %0, %1\n\t")
                   : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
    }
#define MF(fill)                                                                                                              \
  asm volatile(REP32("v_mfma_f32_32x32x16_bf16 %8, %10, %11, %8\n\t" fill "v_mfma_f32_32x32x16_bf16 %9, %10, %11, %9\n\t" fill)      \
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(acc[0]), "+v"(acc[1]) \
               : "v"(a), "v"(b))
    if (MODE == 4) MF("");
    if (MODE == 5) MF("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\t");
    if (MODE == 6) MF("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t");
    if (MODE == 7) MF("v_mul_f32 %0, %0, %0\n\tv_mul_f32 %1, %1, %1\n\tv_mul_f32 %2, %2, %2\n\tv_mul_f32 %3, %3, %3\n\t");
    if (MODE == 8) MF("v_mul_f32 %0, %0, %0\n\tv_mul_f32 %1, %1, %1\n\tv_mul_f32 %2, %2, %2\n\tv_mul_f32 %3, %3, %3\n\tv_mul_f32 %4, %4, %4\n\tv_mul_f32 %5, %5, %5\n\tv_mul_f32 %6, %6, %6\n\tv_mul_f32 %7, %7, %7\n\t");
    if (MODE == 10) MF("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_mul_f32 %2, %2, %2\n\tv_mul_f32 %3, %3, %3\n\tv_mul_f32 %4, %4, %4\n\tv_mul_f32 %5, %5, %5\n\t");
    if (MODE == 11) MF("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\t");
    // a wave-tile of the 32-row flash kernel without memory: 22 MFMAs (two chains), then 32 exps + 16 converts + 24 max3 (the softmax);
    // W waves per SIMD, nothing synchronises them: does the SIMD reach max(matrix, VALU) or their sum?
    if (MODE == 12 || MODE == 14 || MODE == 16) {
      const bool shifted = MODE == 14 && ((threadIdx.x >> 8) & 1);
      if (MODE == 16) asm volatile("s_setprio 1");   // the matrix phase outranks the other waves' VALU phases   // mode 14: every second wave of a SIMD starts with the VALU phase
      if (!shifted)
        asm volatile(REP8("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t")
                     "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
                     "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
                     "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
                     : "+v"(acc[0]), "+v"(acc[1]) : "v"(a), "v"(b));
      if (MODE == 16) asm volatile("s_setprio 0");
      asm volatile(REP8("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
                        "v_cvt_pk_bf16_f32 %8, %0, %1\n\tv_cvt_pk_bf16_f32 %9, %2, %3\n\t"
                        "v_max3_f32 %4, %4, %5, %6\n\tv_max3_f32 %5, %5, %6, %7\n\tv_max3_f32 %6, %6, %7, %4\n\t")
                   : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "=v"(p0), "=v"(p1));
      if (shifted)
        asm volatile(REP8("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t")
                     "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
                     "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
                     "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
                     : "+v"(acc[0]), "+v"(acc[1]) : "v"(a), "v"(b));
    }
    // roles: the waves of the first half of the workgroup run 64 MFMAs per iteration, the others 256 exps / 256 max3
    if (MODE == 13 || MODE == 15 || MODE == 17) {
      if (((threadIdx.x >> 8) & 1) != (MODE == 17)) {   // 17: the VALU waves are the OLDER half
        if (MODE == 13) V8("v_exp_f32");
        else
          asm volatile(REP32("v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %1, %1, %2, %3\n\tv_max3_f32 %2, %2, %3, %4\n\tv_max3_f32 %3, %3, %4, %5\n\t"
                             "v_max3_f32 %4, %4, %5, %6\n\tv_max3_f32 %5, %5, %6, %7\n\tv_max3_f32 %6, %6, %7, %0\n\tv_max3_f32 %7, %7, %0, %1\n\t")
                       : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
      } else {
        MF("");
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
  if (MODE == 2) s += (float)(p0 + p1 + p2 + p3);
  if (MODE == 12 || MODE == 14 || MODE == 16) s += (float)(p0 + p1);
  if (s == 12345.678f) out[1] = 1;   // keep everything live
  // the oldest wave of a SIMD wins every arbitration: report the whole workgroup (first start -> last end), not one wave
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { out[2 + 2 * (threadIdx.x >> 6)] = t0; out[3 + 2 * (threadIdx.x >> 6)] = t1; }
}

template <int MODE>
void run(const char* name, int per_iter, int waves_per_simd) {
  unsigned long long* out;
  hipMalloc(&out, 8 * 64);
  const int iters = 200;
  probe<MODE><<<256, 256 * waves_per_simd>>>(out, iters);
  hipDeviceSynchronize();
  probe<MODE><<<256, 256 * waves_per_simd>>>(out, iters);
  unsigned long long h[64];
  hipMemcpy(h, out, 8 * 64, hipMemcpyDeviceToHost);
  const int nw = 4 * waves_per_simd;
  unsigned long long first = ~0ull, last = 0, fastest = ~0ull;
  for (int w = 0; w < nw; ++w) {
    first = h[2 + 2 * w] < first ? h[2 + 2 * w] : first;
    last = h[3 + 2 * w] > last ? h[3 + 2 * w] : last;
    fastest = h[3 + 2 * w] - h[2 + 2 * w] < fastest ? h[3 + 2 * w] - h[2 + 2 * w] : fastest;
  }
  if (MODE == 17) {
    printf("{\"mode\": \"%s\", \"waves_per_simd\": %d, \"matrix_wave_cycles_per_mfma\": %.2f, \"valu_wave_cycles_per_instruction\": %.2f}\n", name,
           waves_per_simd, (double)(h[3 + 8] - h[2 + 8]) / iters / 64, (double)(h[3] - h[2]) / iters / 256);
  } else
  if (MODE == 13 || MODE == 15) {   // roles: cycles per MFMA of the matrix waves, cycles per VALU instruction of the others (waves 4..7 of 8)
    printf("{\"mode\": \"%s\", \"waves_per_simd\": %d, \"matrix_wave_cycles_per_mfma\": %.2f, \"valu_wave_cycles_per_instruction\": %.2f}\n", name,
           waves_per_simd, (double)(h[3] - h[2]) / iters / 64, (double)(h[3 + 8] - h[2 + 8]) / iters / 256);
  } else
  printf("{\"mode\": \"%s\", \"waves_per_simd\": %d, \"simd_cycles_per_unit\": %.2f, \"fastest_wave_cycles_per_unit\": %.2f}\n", name, waves_per_simd,
         (double)(last - first) / iters / per_iter / waves_per_simd, (double)fastest / iters / per_iter);
  hipFree(out);
}

int main() {
  for (int w = 1; w <= 3; ++w) {
    run<0>("v_exp_f32 (independent)", 256, w);
    run<1>("v_mul_f32 (independent)", 256, w);
    run<2>("v_cvt_pk_bf16_f32", 256, w);
    run<3>("v_max3_f32 (chain of distance 1)", 256, w);
    run<9>("v_max_f32 dependent chain", 256, w);
    run<4>("mfma 32x32x16 bare (per MFMA)", 64, w);
    run<5>("mfma + 2 v_exp (per MFMA)", 64, w);
    run<6>("mfma + 4 v_exp (per MFMA)", 64, w);
    run<11>("mfma + 8 v_exp (per MFMA)", 64, w);
    run<7>("mfma + 4 v_mul (per MFMA)", 64, w);
    run<8>("mfma + 8 v_mul (per MFMA)", 64, w);
    run<10>("mfma + 2 v_exp + 4 v_mul (per MFMA)", 64, w);
    run<12>("wave-tile: 22 mfma THEN 32 exp + 16 cvt + 24 max3 (per wave-tile; matrix 704, VALU port 22*8 + 416)", 1, w);
    if (w == 2) {
      run<14>("the same, every second wave starts with its VALU phase (per wave-tile)", 1, w);
      run<13>("roles: waves 0-3 mfma only, waves 4-7 v_exp only", 1, w);
      run<15>("roles: waves 0-3 mfma only, waves 4-7 v_max3 only", 1, w);
      run<17>("roles reversed: waves 0-3 (older) v_max3 only, waves 4-7 mfma only", 1, w);
    }
    run<16>("wave-tile with s_setprio 1 over its matrix phase (per wave-tile)", 1, w);
  }
  return 0;
}
