// Micro-benchmark: what does one SIMD of gfx950 sustain on fp32 vector instructions, plain and packed, at 1..8 waves per
// SIMD?  (Decides whether packed math can raise the intersection kernels' ceiling.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE> __global__ void __launch_bounds__(256) k(float *out, int iters, float a, float b) {
  float x[8]; float2v y[8];
  for (int i = 0; i < 8; i++) { x[i] = threadIdx.x * 0.001f + i; y[i] = (float2v){ x[i], x[i] + 0.5f }; }
  float2v a2 = { a, a * 1.0001f }, b2 = { b, b * 0.999f };
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(a2), "v"(b2));
        if (MODE == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(a2));
        if (MODE == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "s"(a), "v"(b));     // one scalar operand
        if (MODE == 5) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a), "v"(b) : "vcc");
        if (MODE == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
        if (MODE == 7) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(*(double *)&y[i]) : "v"(*(double *)&a2), "v"(*(double *)&b2));
        if (MODE == 8) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(*(double *)&y[i]) : "v"(x[i]));
        if (MODE == 9) asm volatile("v_add_f64 %0, %0, %1" : "+v"(*(double *)&y[i]) : "v"(*(double *)&a2));
        if (MODE == 10) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(*(double *)&y[i]) : "v"(*(double *)&a2));
        if (MODE == 11) asm volatile("v_cmp_lt_f64 vcc, %1, %2\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(*(int *)&x[i]) : "v"(*(double *)&y[i]), "v"(*(double *)&a2) : "vcc");
        if (MODE == 12) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(*(unsigned *)&x[i]) : "v"(*(unsigned *)&a));
        if (MODE == 13) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(unsigned long long *)&y[i]) : "v"(*(unsigned *)&x[i]), "v"(*(unsigned *)&a) : "vcc");
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += x[i] + y[i].x + y[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> static void run(const char *name, float *d, int flop_per_inst, int inst_per_slot) {
  const int iters = 4000;
  for (int wps = 1; wps <= 8; wps *= 2) {
    const int blocks = 256 * wps;                 // 256 CUs x wps workgroups of 4 waves = wps waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0001f, 0.0001f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)blocks * 4 * iters * 32 * inst_per_slot;      // wave-instructions
    const double per_simd_cycles = ms * 1e-3 * 2.4e9;                             // at 2.4 GHz nominal
    printf("%-28s waves/SIMD %d: %.3f ms  %.2f cycles@2.4GHz per wave-instruction per SIMD  %.1f TFLOP/s\n", name, wps, ms,
           per_simd_cycles / (inst / 1024.0), inst * 64 * flop_per_inst / (ms * 1e-3) / 1e12);
  }
}
int main() {
  float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_fma_f32", d, 2, 1);
  run<1>("v_pk_fma_f32", d, 4, 1);
  run<2>("v_mul_f32", d, 1, 1);
  run<3>("v_pk_mul_f32", d, 2, 1);
  run<4>("v_fma_f32 (sgpr operand)", d, 2, 1);
  run<5>("v_cmp+v_cndmask", d, 0, 2);
  run<6>("v_rcp_f32", d, 1, 1);
  run<7>("v_fma_f64", d, 2, 1);
  run<8>("v_cvt_f64_f32", d, 1, 1);
  run<9>("v_add_f64", d, 1, 1);
  run<10>("v_mul_f64", d, 1, 1);
  run<11>("v_cmp_f64+v_addc", d, 0, 2);
  run<12>("v_mul_hi_u32", d, 0, 1);
  run<13>("v_mad_u64_u32", d, 0, 1);
  return 0;
}
