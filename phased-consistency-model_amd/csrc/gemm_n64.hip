// pcm_gemm_n64_kernel — the rank-64 LoRA-down projection t = x A^T (and d_t = dy (sB) in backward): [M x K] x [64 x K]^T.
//
// 834 launches per bs-16 step, each ONE pass over an activation tensor: pure HBM streaming (2*K bytes in, 128 bytes out per
// row; MFMA time is ~1/10 of the load time).  The tiled kernels of gemm.hip stream it with one or two K-tiles in flight per
// CU and reach ~2.3 TB/s; here every wave issues all of its loads for a K-chunk (up to 320 columns x 32 rows = 20 KB)
// straight into registers in MFMA fragment layout before touching any of them, several blocks per CU, so the CU keeps
// hundreds of KB in flight.  The 64 x KC weight chunk is staged once per block in LDS (padded rows, conflict-free b128 reads).
// v_mfma_f32_16x16x32_bf16 with the weights as the A operand: lane owns 4 consecutive channels of one row (8-byte stores).
#include <stdlib.h>

#include "gemm_dev.h"

template <int NS, int RF>   // NS = 32-column steps per K-chunk (KC = 32*NS); RF = 16-row fragments per wave
__global__ __launch_bounds__(256) void pcm_gemm_n64_kernel(GemmDev g) {
  constexpr int KC = 32 * NS, CPR = KC / 8;          // 16-byte chunks per weight row
  constexpr int PIECES = 64 * CPR;                   // 16-byte pieces of the weight chunk (a multiple of 256 for every NS)
  constexpr int WI = PIECES / 256;
  static_assert(PIECES % 256 == 0 && CPR % 8 == 0, "weight chunk must be whole LDS-DMA instructions / whole 8-chunk groups");
  PCM_DYN_SMEM(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fk = lane >> 4;
  const SegDev& sg = g.seg[0];
  const int m0 = (blockIdx.x * 4 + wave) * (16 * RF);
  const bf16_t* xrow[RF];
#pragma unroll
  for (int j = 0; j < RF; j++) {
    int m = m0 + 16 * j + frow; if (m > g.M - 1) m = g.M - 1;      // clamped: rows beyond M are loaded but never stored
    xrow[j] = sg.a + (size_t)m * sg.lda + 8 * fk;
  }
  f32x4 acc[RF][4];
#pragma unroll
  for (int j = 0; j < RF; j++)
#pragma unroll
    for (int f = 0; f < 4; f++) acc[j][f] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < sg.K; k0 += KC) {
    // this wave's activation fragments for the whole chunk: all loads in flight before the first use
    bf16x8 xf[RF][NS];
#pragma unroll
    for (int j = 0; j < RF; j++)
#pragma unroll
      for (int s = 0; s < NS; s++) xf[j][s] = *(const bf16x8*)(xrow[j] + k0 + 32 * s);
    // weight chunk -> LDS by LDS-DMA (no staging registers, in flight together with the activation loads).  The image is
    // row-major [64][CPR chunks]; the low 3 chunk bits are XOR-swizzled with (row>>1)&7 on the SOURCE side (the LDS side of
    // the DMA is lane-linear), which makes the 16-row b128 fragment reads conflict-free for row strides of 8k chunks.
    if (k0) __syncthreads();
#pragma unroll
    for (int i = 0; i < WI; i++) {
      const int u = 256 * i + tid, n = u / CPR, cp = u - n * CPR;
      const int c = (cp & ~7) | ((cp & 7) ^ ((n >> 1) & 7));
      __builtin_amdgcn_global_load_lds(PCM_AS1(sg.w + (size_t)n * sg.K + k0 + 8 * c), PCM_AS3(smem + (256 * i + 64 * wave) * 16), 16, 0, 0);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NS; s++) {
      bf16x8 wf[4];
#pragma unroll
      for (int f = 0; f < 4; f++) wf[f] = *(const bf16x8*)(smem + ((16 * f + frow) * CPR + (((4 * s + fk) & ~7) | (((4 * s + fk) & 7) ^ ((frow >> 1) & 7)))) * 16);
#pragma unroll
      for (int j = 0; j < RF; j++)
#pragma unroll
        for (int f = 0; f < 4; f++) acc[j][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[f], xf[j][s], acc[j][f], 0, 0, 0);
    }
  }
#pragma unroll
  for (int j = 0; j < RF; j++) {
    const int m = m0 + 16 * j + frow;
    if (m >= g.M) continue;
    bf16_t* orow = (bf16_t*)g.out + (size_t)m * g.ldo + 4 * fk;
#pragma unroll
    for (int f = 0; f < 4; f++)
      *(uint2*)(orow + 16 * f) = make_uint2(pack_bf2(acc[j][f][0] * g.alpha, acc[j][f][1] * g.alpha), pack_bf2(acc[j][f][2] * g.alpha, acc[j][f][3] * g.alpha));
  }
}

// Small-M variant (M <= 16384: the 16x16 / 8x8 / 32x32 feature maps, the 4096-token SDXL / SD3 levels).  There the pass is latency bound,
// not bandwidth bound: a block owns 16 RF rows and splits K over its waves (32 NS columns per wave and sub-chunk, NS = 5 or 4), every wave
// reads its activation AND weight fragments straight from global / L2 (no LDS staging, no chunk loop for K <= 1280), the partial tiles are
// summed through LDS.  One memory round trip instead of one per K-chunk.  RF = 2 (32 rows per block) where that still gives >= 256 blocks:
// every block reads the whole 64 x K weight matrix from L2, which at 16 rows per block is 4x the activation bytes ((8192, K 5120): 335 MB
// of weight reads for 84 MB of activations).
template <int NW, int RF, int NS>
__global__ __launch_bounds__(64 * NW) void pcm_gemm_n64_ksplit_kernel(GemmDev g, int sub) {
  __shared__ __attribute__((aligned(16))) float red[NW][16 * RF][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fk = lane >> 4;
  const SegDev& sg = g.seg[0];
  const bf16_t* xr[RF];
#pragma unroll
  for (int j = 0; j < RF; j++) {
    int m = blockIdx.x * (16 * RF) + 16 * j + frow; if (m > g.M - 1) m = g.M - 1;
    xr[j] = sg.a + (size_t)m * sg.lda + 8 * fk;
  }
  const bf16_t* wr = sg.w + (size_t)frow * sg.K + 8 * fk;
  f32x4 acc[RF][4];
#pragma unroll
  for (int j = 0; j < RF; j++)
#pragma unroll
    for (int f = 0; f < 4; f++) acc[j][f] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int sc = 0; sc < sub; sc++) {
    const int k0 = (wave * sub + sc) * (32 * NS);
    bf16x8 xf[RF][NS], wf[4][NS];
#pragma unroll
    for (int j = 0; j < RF; j++)
#pragma unroll
      for (int s = 0; s < NS; s++) xf[j][s] = *(const bf16x8*)(xr[j] + k0 + 32 * s);
#pragma unroll
    for (int f = 0; f < 4; f++)
#pragma unroll
      for (int s = 0; s < NS; s++) wf[f][s] = *(const bf16x8*)(wr + (size_t)(16 * f) * sg.K + k0 + 32 * s);
    __builtin_amdgcn_sched_barrier(0);   // all loads in flight before the first MFMA (hipcc otherwise sinks them one by one)
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
      for (int f = 0; f < 4; f++)
#pragma unroll
        for (int j = 0; j < RF; j++) acc[j][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[f][s], xf[j][s], acc[j][f], 0, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < RF; j++)
#pragma unroll
    for (int f = 0; f < 4; f++) *(float4*)&red[wave][16 * j + frow][16 * f + 4 * fk] = make_float4(acc[j][f][0], acc[j][f][1], acc[j][f][2], acc[j][f][3]);
  __syncthreads();
  for (int idx = tid; idx < 256 * RF; idx += 64 * NW) {
    const int row = idx >> 4, c4 = idx & 15;
    const int mm = blockIdx.x * (16 * RF) + row;
    if (mm >= g.M) continue;
    float4 t = *(const float4*)&red[0][row][4 * c4];
#pragma unroll
    for (int w = 1; w < NW; w++) {
      const float4 u = *(const float4*)&red[w][row][4 * c4];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    *(uint2*)((bf16_t*)g.out + (size_t)mm * g.ldo + 4 * c4) = make_uint2(pack_bf2(t.x * g.alpha, t.y * g.alpha), pack_bf2(t.z * g.alpha, t.w * g.alpha));
  }
}
template <int RF, int NS>
static void launch_ksplit(const GemmDev& g, int parts, void* stream) {
  const dim3 grid((g.M + 16 * RF - 1) / (16 * RF));
  if (parts % 8 == 0) PCM_LAUNCH((pcm_gemm_n64_ksplit_kernel<8, RF, NS>), grid, dim3(512), 0, stream, g, parts / 8);
  else if (parts % 4 == 0) PCM_LAUNCH((pcm_gemm_n64_ksplit_kernel<4, RF, NS>), grid, dim3(256), 0, stream, g, parts / 4);
  else if (parts % 2 == 0) PCM_LAUNCH((pcm_gemm_n64_ksplit_kernel<2, RF, NS>), grid, dim3(128), 0, stream, g, parts / 2);
  else PCM_LAUNCH((pcm_gemm_n64_ksplit_kernel<1, RF, NS>), grid, dim3(64), 0, stream, g, parts);
}

template <int NS, int RF>
static void launch_n64(const GemmDev& g, void* stream) {
  const int rows = 4 * 16 * RF;
  const size_t smem = 64 * (size_t)(32 * NS * 2);
  PCM_LAUNCH((pcm_gemm_n64_kernel<NS, RF>), dim3((g.M + rows - 1) / rows), dim3(256), smem, stream, g);
}
// preconditions (checked by the planner in gemm.hip): one plain segment, N == 64, K % 64 == 0, bf16 output, no bias / row vector /
// residual / activation
int pcm_gemm_n64_launch(const GemmDev& g, void* stream) {
  const int K = g.seg[0].K;
  // (128-column pieces only up to M = 8192: at (16384, K 1536) the chunked streaming kernel below runs 17.4 us, this one 21.7)
  if (g.M <= 16384 && (K % 160 == 0 || (K % 128 == 0 && g.M <= 8192))) {
    // rows per block: 16 RF.  Every block re-reads the whole weight matrix from L2, so more rows per block cut the L2 traffic -- as long as
    // the grid still covers the chip: measured per shape with RF forced (profiles/r04_t_n64_ksplit_rows_per_block.txt), the fastest form
    // is RF = 4 (224 VGPRs) from 256 blocks of 64 rows, RF = 2 from 256 blocks of 32 rows, RF = 1 below (at M = 4096 RF = 2 is 15 % slower
    // at every K).  PCM_N64_RF overrides (tuning).
#if PCM_HAS_TOOLS
    static const int env_rf = pcm_env_int("PCM_N64_RF", 0);
#else
    constexpr int env_rf = 0;
#endif
    int rf = env_rf ? env_rf : (g.M >= 64 * PCM_GRID_CAP(256) ? 4 : (g.M >= 32 * PCM_GRID_CAP(256) ? 2 : 1));
    if (K % 160 == 0) { if (rf == 4) launch_ksplit<4, 5>(g, K / 160, stream); else if (rf == 2) launch_ksplit<2, 5>(g, K / 160, stream); else launch_ksplit<1, 5>(g, K / 160, stream); }
    else { if (rf == 4) launch_ksplit<4, 4>(g, K / 128, stream); else if (rf == 2) launch_ksplit<2, 4>(g, K / 128, stream); else launch_ksplit<1, 4>(g, K / 128, stream); }
    return 0;
  }
  const bool wide = (g.M + 127) / 128 >= 512;       // enough 128-row blocks for two per CU: 32 rows per wave, else 16
  if (K % 320 == 0) { if (wide) launch_n64<10, 2>(g, stream); else launch_n64<10, 1>(g, stream); }
  else if (K % 256 == 0) { if (wide) launch_n64<8, 2>(g, stream); else launch_n64<8, 1>(g, stream); }
  else { if (wide) launch_n64<2, 2>(g, stream); else launch_n64<2, 1>(g, stream); }
  return 0;
}
