// Can the matrix pipe run under VALU work on gfx950?  (valubench.hip showed: MFMAs of one wave and VALU of ANOTHER wave on the same SIMD
// serialize when the MFMA wave issues back to back -- an MFMA waiting for the busy pipe holds the SIMD's VALU issue port.)
// Here every wave interleaves 1 MFMA 32x32x16 with NV independent v_fma_f32 (the attention ratio is ~7-14 VALU per MFMA) and the time
// is compared with the VALU-only and MFMA-only runs, at 1 / 2 / 3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 2048
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 b8v __attribute__((ext_vector_type(8)));
template <int NV, bool MF, bool AG>
__global__ __launch_bounds__(768) void k(float* out, float seed) {
  float v[16];
  for (int i = 0; i < 16; i++) v[i] = seed * (threadIdx.x + i) * 1e-3f;
  f16v ma[4]; b8v mb;
  for (int c = 0; c < 4; c++) for (int i = 0; i < 16; i++) ma[c][i] = seed * (i + c);
  for (int i = 0; i < 8; i++) mb[i] = (__bf16)(seed * (threadIdx.x & 7));
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (MF && !AG) ma[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mb, mb, ma[u], 0, 0, 0);
      if (MF && AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %1, %0" : "+a"(ma[u]) : "v"(mb));   // accumulator in AccVGPRs
#pragma unroll
      for (int j = 0; j < NV; j++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[j & 15]));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = ma[0][0] + ma[1][3] + ma[2][1] + ma[3][2];
  for (int i = 0; i < 16; i++) s += v[i];
  if (s == 123.456f) out[0] = s;
}
template <int NV, bool MF, bool AG>
static float run(int threads) {
  float* o; hipMalloc(&o, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, MF, AG>), dim3(256), dim3(threads), 0, 0, o, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, MF, AG>), dim3(256), dim3(threads), 0, 0, o, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
#define ROW(NV) for (int w = 1; w <= 3; w++) { float a = run<NV, true, false>(256 * w), g = run<NV, true, true>(256 * w), b = run<NV, false, false>(256 * w), c = run<0, true, false>(256 * w), c2 = run<0, true, true>(256 * w); \
  printf("%2d v_fma per MFMA, %d wave(s)/SIMD: VGPR-acc interleaved %7.3f ms | AGPR-acc interleaved %7.3f ms | VALU only %7.3f | MFMA only %7.3f (AGPR %7.3f) | sum %7.3f\n", NV, w, a, g, b, c, c2, b + c); }
int main() {
  ROW(4) ROW(8) ROW(12) ROW(16) ROW(24)
  return 0;
}
