// VALU issue rates on gfx950 that decide the attention softmax budget (attention.hip): v_exp_f32, v_fma_f32, v_pk_fma_f32, v_pk_mul_f32,
// v_max3_f32, v_cvt_pk_bf16_f32, v_ldexp_f32 -- and whether the transcendental ops of one wave overlap the plain VALU ops of ANOTHER wave
// on the same SIMD (separate pipe) or serialize with them (same pipe).
// hipcc --offload-arch=gfx950 -O2 valubench.hip -o valubench.  256 blocks x 512 threads: 8 waves per CU = 2 per SIMD (w and w+4 share one).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 4096
// 8 independent chains per iteration so the dependent-issue latency never limits
#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 b8v __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, float seed) {
  float v[8], w[8];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[8];
  for (int i = 0; i < 8; i++) { v[i] = seed * (threadIdx.x + i) * 1e-3f; w[i] = v[i] * 0.5f; p[i] = f2{v[i], w[i]}; }
  const int wave = threadIdx.x >> 6;
  int mode = MODE;
  if (MODE == 100) mode = (wave < 4) ? 0 : 1;      // half the waves exp, the other half fma (one of each per SIMD)
  if (MODE == 101) mode = (wave < 4) ? 0 : 2;      // exp + pk_fma
  if (MODE == 102) mode = (wave < 4) ? 11 : 1;     // MFMA waves + fma waves on the same SIMDs
  if (MODE == 103) mode = (wave < 4) ? 11 : 0;     // MFMA waves + exp waves
  if (MODE == 104) mode = (wave < 4) ? 11 : 99;    // MFMA waves alone (one per SIMD), the other four waves idle
  if (MODE == 105) mode = (wave < 4) ? 13 : 99;    // MFMA waves alone, 4 independent accumulator chains
  if (MODE == 106) mode = (wave < 4) ? 13 : 1;     // 4-chain MFMA waves + fma waves
  if (MODE == 107) mode = (wave < 4) ? 13 : 0;     // 4-chain MFMA waves + exp waves
  if (MODE == 108) mode = (wave < 4) ? 99 : 1;     // fma waves alone (one per SIMD)
  f16v mc[2];
  for (int i = 0; i < 16; i++) { mc[0][i] = seed * (i + 2); mc[1][i] = seed * (i + 3); }
  f16v ma[2]; b8v mb;
  for (int i = 0; i < 16; i++) { ma[0][i] = seed * i; ma[1][i] = seed * (i + 1); }
  for (int i = 0; i < 8; i++) mb[i] = (__bf16)(seed * (threadIdx.x & 7));
  if (mode == 0) {
    for (int it = 0; it < ITER; it++) {
#define OP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
      REP8(OP)
#undef OP
    }
  } else if (mode == 1) {
    for (int it = 0; it < ITER; it++) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w[i]));
      REP8(OP)
#undef OP
    }
  } else if (mode == 2) {
    for (int it = 0; it < ITER; it++) {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
      REP8(OP)
#undef OP
    }
  } else if (mode == 3) {
    for (int it = 0; it < ITER; it++) {
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[i]));
      REP8(OP)
#undef OP
    }
  } else if (mode == 4) {
    for (int it = 0; it < ITER; it++) {
#define OP(i) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w[i]));
      REP8(OP)
#undef OP
    }
  } else if (mode == 5) {
    for (int it = 0; it < ITER; it++) {
#define OP(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
      REP8(OP)
#undef OP
    }
  } else if (mode == 6) {
    for (int it = 0; it < ITER; it++) {
#define OP(i) asm volatile("v_ldexp_f32 %0, %0, 1" : "+v"(v[i]));
      REP8(OP)
#undef OP
    }
  } else if (mode == 7) {
    for (int it = 0; it < ITER; it++) {
#define OP(i) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p[i]));
      REP8(OP)
#undef OP
    }
  } else if (mode == 8) {
    for (int it = 0; it < ITER; it++) {
#define OP(i) asm volatile("v_exp_f16 %0, %0" : "+v"(v[i]));
      REP8(OP)
#undef OP
    }
  } else if (mode == 9) {   // exp and fma INTERLEAVED inside one wave (does the issue of a trans op block the wave's next plain op?)
    for (int it = 0; it < ITER; it++) {
#define OP(i) asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(v[i]), "+v"(w[i]), "+v"(p[i].x), "+v"(p[i].y));
      REP8(OP)
#undef OP
    }
  } else if (mode == 10) {
    for (int it = 0; it < ITER; it++) {
#define OP(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
      REP8(OP)
#undef OP
    }
  }
  else if (mode == 11) {  // 8 MFMAs per iteration on two independent accumulators
    for (int it = 0; it < ITER; it++) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        ma[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mb, mb, ma[0], 0, 0, 0);
        ma[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mb, mb, ma[1], 0, 0, 0);
      }
    }
  } else if (mode == 13) {  // 8 MFMAs per iteration on FOUR independent accumulators
    for (int it = 0; it < ITER; it++) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        ma[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mb, mb, ma[0], 0, 0, 0);
        ma[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mb, mb, ma[1], 0, 0, 0);
        mc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mb, mb, mc[0], 0, 0, 0);
        mc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mb, mb, mc[1], 0, 0, 0);
      }
    }
  } else if (mode == 12) {  // ONE wave type: 1 MFMA + 8 independent v_fma_f32 per group, 8 groups per iteration
    for (int it = 0; it < ITER; it++) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        ma[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mb, mb, ma[u & 1], 0, 0, 0);
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w[i]));
        REP8(OP)
#undef OP
      }
    }
  }
  float s = ma[0][0] + ma[1][3] + mc[0][1] + mc[1][2];
  for (int i = 0; i < 8; i++) s += v[i] + w[i] + p[i].x + p[i].y;
  if (s == 123.456f) out[0] = s;
}
template <int MODE>
static float run() {
  float* o; hipMalloc(&o, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, o, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, o, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
// cycles per wave-instruction at 2 waves per SIMD: time * f / (ITER * 8 instr * 2 waves)
#define RUN(title, M, ninstr) { float t = run<M>(); printf("%-64s %8.3f ms  = %6.2f SIMD cycles per wave-instruction (@2.4 GHz)\n", title, t, t * 1e-3 * 2.4e9 / (ITER * 8.0 * 2 * (ninstr))); }
int main() {
  RUN("v_exp_f32", 0, 1);
  RUN("v_fma_f32", 1, 1);
  RUN("v_pk_fma_f32 (2 fma per lane)", 2, 1);
  RUN("v_pk_mul_f32", 3, 1);
  RUN("v_pk_add_f32", 7, 1);
  RUN("v_max3_f32", 4, 1);
  RUN("v_max_f32", 10, 1);
  RUN("v_cvt_pk_bf16_f32", 5, 1);
  RUN("v_ldexp_f32", 6, 1);
  RUN("v_exp_f16", 8, 1);
  { float a = run<0>(), b = run<1>(), c = run<100>(); printf("waves 0-3 v_exp_f32 / waves 4-7 v_fma_f32 on the same SIMDs: %.3f ms; all-exp %.3f, all-fma %.3f -> separate pipes would give %.3f, one pipe %.3f\n", c, a, b, (a > b ? a : b) / 2, (a + b) / 2); }
  { float a = run<0>(), b = run<2>(), c = run<101>(); printf("waves 0-3 v_exp_f32 / waves 4-7 v_pk_fma_f32: %.3f ms; all-exp %.3f, all-pk_fma %.3f -> separate %.3f, one pipe %.3f\n", c, a, b, (a > b ? a : b) / 2, (a + b) / 2); }
  { float a = run<11>(), b = run<1>(), c = run<102>(); printf("MFMA 32x32x16 bf16: all 8 waves %.3f ms = %.2f cycles per MFMA per SIMD (@2.4 GHz); waves 0-3 MFMA / waves 4-7 v_fma_f32: %.3f ms (all-fma %.3f) -> separate pipes %.3f, one pipe %.3f\n", a, a * 1e-3 * 2.4e9 / (ITER * 8.0 * 2), c, b, (a > b ? a : b) / 2, (a + b) / 2); }
  { float a = run<11>(), b = run<0>(), c = run<103>(); printf("waves 0-3 MFMA / waves 4-7 v_exp_f32: %.3f ms (all-MFMA %.3f, all-exp %.3f) -> separate pipes %.3f, one pipe %.3f\n", c, a, b, (a > b ? a : b) / 2, (a + b) / 2); }
  { float a = run<104>(), b = run<105>(), c = run<108>(), d = run<106>(), e = run<107>(), f = run<0>();
    printf("ONE MFMA wave per SIMD alone: 2 chains %.3f ms, 4 chains %.3f ms; one fma wave per SIMD alone %.3f ms; 4-chain MFMA wave + fma wave %.3f ms; 4-chain MFMA wave + exp wave %.3f ms (exp wave alone %.3f)\n", a, b, c, d, e, f / 2); }
  { float t = run<12>(), a = run<11>(), b = run<1>(); printf("every wave: {1 MFMA + 8 independent v_fma_f32} x 8 per iteration: %.3f ms; MFMA part alone %.3f, fma part alone %.3f (8 fma per group = the all-fma run)\n", t, a, b); }
  { float t = run<9>(); printf("one wave: {v_exp_f32 + 3 v_fma_f32} x 8 per iteration: %.3f ms = %.2f cycles per group (serial would be 16 + 12 = 28)\n", t, t * 1e-3 * 2.4e9 / (ITER * 8.0 * 2)); }
  return 0;
}
