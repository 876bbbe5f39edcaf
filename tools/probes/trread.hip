// Empirical lane mapping of ds_read_b64_tr_b16 on gfx950 (the guides give one layout formula only).
// LDS holds 16-bit values equal to their element index; every lane supplies its own byte address and
// we print which 4 elements it gets back.   hipcc --offload-arch=gfx950 -O2 trread.hip -o trread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k(const int* addr, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  unsigned a = (unsigned)(size_t)lds + (unsigned)addr[threadIdx.x];
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  for (int j = 0; j < 4; j++) out[4 * threadIdx.x + j] = (unsigned short)(v >> (16 * j));
}
static void run(const char* title, int (*f)(int)) {
  int h[64]; for (int l = 0; l < 64; l++) h[l] = f(l);
  int* d; unsigned short* o; unsigned short ho[256];
  hipMalloc(&d, 256); hipMalloc(&o, 512);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  hipMemcpy(ho, o, 512, hipMemcpyDeviceToHost);
  printf("== %s\n", title);
  for (int l = 0; l < 64; l++) printf("lane %2d addr_elem %4d -> %4d %4d %4d %4d\n", l, h[l] / 2, ho[4 * l], ho[4 * l + 1], ho[4 * l + 2], ho[4 * l + 3]);
}
int main() {
  run("A: addr = lane*8 bytes (each lane its own 4 contiguous elements)", [](int l) { return l * 8; });
  run("B: addr = 2*((l&15) + (l>>4)*64) (guide formula base, elem j at +16 elements)", [](int l) { return 2 * ((l & 15) + (l >> 4) * 64); });
  run("C: row-major [m][64 cols] tile, lane -> row (l&15)... addr = (l&15)*128 + (l>>4)*8", [](int l) { return (l & 15) * 128 + (l >> 4) * 8; });
  run("D: addr = (l&3)*8 + (l>>2)*128 : 4 lanes per row of 16 elements, 16 rows of 64 elements", [](int l) { return (l & 3) * 8 + (l >> 2) * 128; });
  return 0;
}
