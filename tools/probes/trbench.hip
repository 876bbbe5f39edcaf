// Throughput of ds_read_b64_tr_b16 vs ds_read_b128 / ds_read_b64 for the tile layouts used by attention.hip / wgrad_tr.hip.
// hipcc --offload-arch=gfx950 -O2 trbench.hip -o trbench ; one block per CU x 4 waves, each wave issues ITER x 8 reads.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define ITER 2048
template <int MODE>
__global__ __launch_bounds__(256) void k(const int* addr, unsigned* out, int stride_bytes) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  for (int i = threadIdx.x; i < 16384; i += 256) ((unsigned*)lds)[i] = i;
  __syncthreads();
  unsigned a = (unsigned)(size_t)lds + (unsigned)addr[threadIdx.x & 63];
  unsigned long long acc = 0;
  for (int it = 0; it < ITER; it++) {
    unsigned long long v0, v1, v2, v3, v4, v5, v6, v7;
    if (MODE == 0) {
      asm volatile("ds_read_b64_tr_b16 %0, %8\n ds_read_b64_tr_b16 %1, %8 offset:1024\n ds_read_b64_tr_b16 %2, %8 offset:2048\n ds_read_b64_tr_b16 %3, %8 offset:3072\n"
                   "ds_read_b64_tr_b16 %4, %8 offset:4096\n ds_read_b64_tr_b16 %5, %8 offset:5120\n ds_read_b64_tr_b16 %6, %8 offset:6144\n ds_read_b64_tr_b16 %7, %8 offset:7168\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(a));
    } else {
      asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:1024\n ds_read_b64 %2, %8 offset:2048\n ds_read_b64 %3, %8 offset:3072\n"
                   "ds_read_b64 %4, %8 offset:4096\n ds_read_b64 %5, %8 offset:5120\n ds_read_b64 %6, %8 offset:6144\n ds_read_b64 %7, %8 offset:7168\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(a));
    }
    acc += v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
  }
  if (acc == 0x1234567) out[0] = 1;
}
template <int MODE>
static float run(int (*f)(int), int nblocks) {
  int h[64]; for (int l = 0; l < 64; l++) h[l] = f(l);
  int* d; unsigned* o; hipMalloc(&d, 256); hipMalloc(&o, 4);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(nblocks), dim3(256), 0, 0, d, o, 0);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(nblocks), dim3(256), 0, 0, d, o, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
#define RUN(title, fn) { float t0 = run<0>(fn, 256), t1 = run<1>(fn, 256); \
  printf("%-70s tr_b16 %7.3f ms = %5.1f cyc/wave-instr(@2.4GHz, 4 waves/CU)   b64 %7.3f ms = %5.1f\n", title, t0, t0 * 1e-3 * 2.4e9 / (ITER * 8.0 * 4), t1, t1 * 1e-3 * 2.4e9 / (ITER * 8.0 * 4)); }
static int f_att40(int l) { int s = l & 15, j = s >> 2, q = s & 3, g = l >> 4; return (4 * (g >> 1) + j) * 112 + (16 * (g & 1) + 4 * q) * 2; }
static int f_att64(int l) { int s = l & 15, j = s >> 2, q = s & 3, g = l >> 4; return (4 * (g >> 1) + j) * 144 + (16 * (g & 1) + 4 * q) * 2; }
static int f_128(int l) { int s = l & 15, j = s >> 2, q = s & 3, g = l >> 4; return (8 * (g >> 1) + j) * 128 + (16 * (g & 1) + 4 * q) * 2; }
static int f_128s(int l) { int s = l & 15, j = s >> 2, q = s & 3, g = l >> 4; int ch = (2 * (g & 1) + (q >> 1)) ^ (((j >> 1) & 1) << 2); return (8 * (g >> 1) + j) * 128 + ch * 16 + 8 * (q & 1); }
static int f_256s(int l) { int s = l & 15, j = s >> 2, q = s & 3, g = l >> 4; int ch = (2 * (g & 1) + (q >> 1)) ^ (j << 2); return (8 * (g >> 1) + j) * 256 + ch * 16 + 8 * (q & 1); }
static int f_256(int l) { int s = l & 15, j = s >> 2, q = s & 3, g = l >> 4; return (8 * (g >> 1) + j) * 256 + (16 * (g & 1) + 4 * q) * 2; }
static int f_lin(int l) { return l * 8; }
int main() {
  RUN("attention d=40 rows of 112 B", f_att40);
  RUN("attention d=64 rows of 144 B", f_att64);
  RUN("rows of 128 B, no swizzle (wgrad u window)", f_128);
  RUN("rows of 128 B, chunk ^ ((j>>1)&1)<<2 (wgrad small / x tile)", f_128s);
  RUN("rows of 256 B, chunk ^ (j<<2) (wgrad big tile)", f_256s);
  RUN("rows of 256 B, no swizzle", f_256);
  RUN("linear lane*8 (conflict-free reference)", f_lin);
  return 0;
}
