// pcm_gemm_smallm_kernel -- the M <= 16 member of the pcm_gemm_bf16 family: the projections whose "M" is the BATCH, not the pixels
// (time-embedding MLP and the per-resnet time_emb_proj of the UNets, train_pcm_lora_sd15.py:1192 -> diffusers TimestepEmbedding /
// ResnetBlock2D.time_emb_proj; the adaLN modulation linears of the MMDiT, M = 2 .. 4, N = 9216).  They are pure weight streams -- a
// (16, 1280, K 1280) call reads 3.3 MB and does 0.05 GFLOP -- and ran 13 - 17 us each through the 64 x 64 tile with its LDS staging, K loop
// and (for long K) slab round trip: ~140 launches per SD3 step (batch 2 / 4), fewer in the UNet configs.
// (M <= 16, one MFMA row tile: at M = 17 .. 32 a second row tile worked as well, but bs-16 runs then take this kernel for their 2B = 32
//  row passes and the generic tile for a 64-row one -- per-row results of ONE kernel do not depend on M, those of two kernels differ in the
//  last bit, and tests/test_emu_unet.py holds the forward of a half batch bit-identical to that half of the full batch.)
//
// Here a block owns 16 output columns for ALL rows; its waves split the K steps (32 wide) of both segments round-robin, every lane
// loads its 16-B piece of one weight row and of one activation row straight from global memory (several K steps in flight, no
// LDS staging), the partial 16 x 16 tiles are summed through LDS and one wave's worth of threads applies alpha / bias / SiLU and stores.
// One memory round trip + one reduction: bound by launch latency and the weight stream, not by a tile pipeline.
#include "gemm_dev.h"

template <int NW>
__global__ __launch_bounds__(64 * NW) void pcm_gemm_smallm_kernel(GemmDev g) {
  __shared__ __attribute__((aligned(16))) float red[NW][16][20];         // [wave][row j][channel i] (+4: bank spread)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fk = lane >> 4;
  const int n0 = blockIdx.x * 16;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int U = 4;                                                   // K steps in flight per wave
  for (int si = 0; si < g.nseg; si++) {
    const SegDev& sg = g.seg[si];
    const int nsteps = sg.K >> 5;
    int n = n0 + frow; if (n > g.N - 1) n = g.N - 1;
    int m = frow; if (m > g.M - 1) m = g.M - 1;                          // rows beyond M: a valid row is read, the result is not stored
    const bf16_t* wr = sg.w + (size_t)n * sg.K + 8 * fk;
    const bf16_t* xr = sg.a + (size_t)m * sg.lda + 8 * fk;
    for (int s0 = wave; s0 < nsteps; s0 += NW * U) {
      bf16x8 wf[U], xf[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int s = s0 + u * NW;
        const int k0 = (s < nsteps ? s : s0) << 5;                       // (past the end: re-read a valid step, its MFMA is skipped)
        wf[u] = *(const bf16x8*)(wr + k0);
        xf[u] = *(const bf16x8*)(xr + k0);
      }
      __builtin_amdgcn_sched_barrier(0);                                  // all loads in flight before the first MFMA
#pragma unroll
      for (int u = 0; u < U; u++)
        if (s0 + u * NW < nsteps) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], xf[u], acc, 0, 0, 0);   // D[i = channel][j = row]
    }
  }
  // lane holds row j = frow, channels 4 fk .. 4 fk + 3
  *(float4*)&red[wave][frow][4 * fk] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  __syncthreads();
  if (tid < 64) {
    const int j = tid >> 2, c4 = tid & 3;
    const int m = j, n = n0 + 4 * c4;
    if (m < g.M && n < g.N) {
      float4 v = *(const float4*)&red[0][j][4 * c4];
#pragma unroll
      for (int w = 1; w < NW; w++) {
        const float4 u = *(const float4*)&red[w][j][4 * c4];
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      v.x *= g.alpha; v.y *= g.alpha; v.z *= g.alpha; v.w *= g.alpha;
      if (g.bias) { const float4 b = *(const float4*)(g.bias + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
      if (g.act == PCM_ACT_SILU) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
      if (g.out_f32) *(float4*)((float*)g.out + (size_t)m * g.ldo + n) = v;
      else *(uint2*)((bf16_t*)g.out + (size_t)m * g.ldo + n) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
    }
  }
}

// preconditions (gemm.hip gemm_smallm_ok): M <= 16, plain segments with K % 32 == 0 and 16-B aligned rows, N % 4 == 0, no row vector /
// residual, activation none or SiLU
int pcm_gemm_smallm_launch(const GemmDev& g, void* stream) {
  const dim3 grid((g.N + 15) / 16);
  PCM_LAUNCH((pcm_gemm_smallm_kernel<4>), grid, dim3(256), 0, stream, g);
  return 0;
}
