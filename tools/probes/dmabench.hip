// LDS-DMA (buffer_load_dwordx4 ... lds) throughput per CU on gfx950, source resident in L2: 8 waves of a workgroup issue NI instructions
// per iteration (1 KB each) into a 64 KB LDS window, with a counted wait that keeps NI in flight -- the staging pattern of gemm8p.hip
// (72 KB per K-tile per CU against 2560 MFMA cycles).  Also: the same with plain global_load_dwordx4 to registers for comparison.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 512
template <int NI, int MODE>
__global__ __launch_bounds__(512) void k(const char* src, unsigned* out, int span) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = src + (size_t)blockIdx.x * span;          // every workgroup re-reads its own window (L2 / MALL resident)
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x80000000u, 0x00020000);
  unsigned acc = 0;
  for (int it = 0; it < ITER; it++) {
    const unsigned off0 = (unsigned)(((it * 8 + wave) * NI) * 1024) % (unsigned)span;
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < NI; j++)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + ((wave * NI + j) % 64) * 1024), 16, lane * 16, (off0 + j * 1024) % span, 0, 0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
    } else {
      typedef int v4i __attribute__((ext_vector_type(4)));
      v4i r[NI];
#pragma unroll
      for (int j = 0; j < NI; j++) r[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (off0 + j * 1024) % span, 0);
#pragma unroll
      for (int j = 0; j < NI; j++) acc += r[j][0] ^ r[j][3];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (acc == 0x12345 || lds[threadIdx.x] == 77) out[0] = acc;
}
template <int NI, int MODE>
static void run(const char* src, unsigned* out, int span, const char* what) {
  hipFuncSetAttribute((const void*)k<NI, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NI, MODE>), dim3(256), dim3(512), 65536, 0, src, out, span);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NI, MODE>), dim3(256), dim3(512), 65536, 0, src, out, span);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes_per_cu = (double)ITER * 8 * NI * 1024;
  printf("%-46s NI=%d window %4d KB/CU: %7.3f ms  %6.1f B/clk/CU (@2.4 GHz)  %6.2f TB/s chip\n", what, NI, span / 1024, ms, bytes_per_cu / (ms * 1e-3 * 2.4e9), bytes_per_cu * 256 / (ms * 1e-3) / 1e12);
}
int main() {
  char* src; unsigned* out;
  hipMalloc(&src, 256u * 1024 * 1024); hipMemset(src, 1, 256u * 1024 * 1024); hipMalloc(&out, 4);
  for (int span : {64 * 1024, 512 * 1024}) {     // 64 KB per CU: 16 MB in all (L2 4 MB/XCD x 8 = 32 MB); 512 KB per CU: 128 MB (MALL)
    run<2, 0>(src, out, span, "buffer_load_dwordx4 ... lds");
    run<4, 0>(src, out, span, "buffer_load_dwordx4 ... lds");
    run<9, 0>(src, out, span, "buffer_load_dwordx4 ... lds");
    run<4, 1>(src, out, span, "buffer_load_dwordx4 to VGPRs");
    run<8, 1>(src, out, span, "buffer_load_dwordx4 to VGPRs");
  }
  return 0;
}
