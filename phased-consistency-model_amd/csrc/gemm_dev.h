// Device-side argument block shared by the GEMM kernels (gemm.hip: 4-wave tiles; gemm8p.hip: 256-row phased tile).
#pragma once
#include "pcm_common.h"

struct SegDev {
  const bf16_t* a;
  const bf16_t* w;
  int K, lda, mode, Hs, Ws, C, stride, src_mode, ktiles;
};
struct GemmDev {
  SegDev seg[2];
  int nseg, M, N, Ho, Wo;
  const float* bias;
  const bf16_t* rowvec;
  int rpb;
  const bf16_t* res;
  int ldr;
  void* out;
  int ldo, out_f32, act;
  float alpha;
  int tiles_m, tiles_n;
  int splitk, kt_per_split;   // split-K: blockIdx.y = K slice; raw fp32 partial tiles go to slab ws[slice][M][N]
  float* ws;
  int conv_md, conv_co;       // gemm8p.hip address-path / K-order variants of the stride-1 direct 3x3 view (A/B hooks; defaults 0, 0)
  int conv_auto;              // opt-in: the gemm8p launcher picks the K order by shape (chunk-outer on 8x8 feature maps)
  bf16_t* pre_out;            // PCM_ACT_GEGLU: optional second output, the interleaved pre-activation of rows < pre_rows (row stride ldp)
  int pre_rows, ldp;
  int dbg;                    // ablation mask: only read by -DPCM_ABLATE builds (tools/probes/build_ablate.py), 0 otherwise
  bf16_t* out2;               // abi 5: second copy of the bf16 output rows (row stride ldo2) -- a skip tensor written straight into the channel
  int ldo2;                   // range of its future concat buffer (no concat pass); nullptr: none
  double* chstats;            // abi 5: per-(sample, channel) {sum, sumsq} of the stored output, fp64 atomics into [M / stats_rows][N][2] (pre-zeroed);
  int stats_rows;             // gemm8p's unsplit epilogue (gemm_epilogue.h stats_pass) and the split-K finalize emit it; nullptr: none
  int w4_stagger;             // gemm4w.hip: start delay (units of ~0.85 us) of the odd-numbered workgroup slot of a CU, so that the two
                              // co-resident workgroups do not run their MFMA and their epilogue phases in lockstep; 0 = none
};
// timing ablations for tools/gemm8p_ablate.py (results are wrong by construction): 1 = no global stores in the epilogue, 2 = no epilogue,
// 4 = no MFMAs, 8 = no LDS-DMA after the prologue.  Compiled out of the product library.
#ifdef PCM_ABLATE
#define PCM_ABL(bit) (g.dbg & (bit))
#else
#define PCM_ABL(bit) 0
#endif

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// fused epilogue on 8 consecutive channels of one output row (values already scaled by alpha): + bias + per-batch row
// vector (time embedding), SiLU, + residual, one 16-byte bf16 store.  Requires the 16-B alignment the planner checks.
__device__ __forceinline__ void pcm_epi_store8(const GemmDev& g, int m, int n, float v[8]) {
  if (g.bias) {
    const float4 b0 = *(const float4*)(g.bias + n), b1 = *(const float4*)(g.bias + n + 4);
    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
  }
  if (g.rowvec) {
    const uint4 t = *(const uint4*)(g.rowvec + (size_t)(m / g.rpb) * g.N + n);
    const unsigned tw[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int e = 0; e < 4; e++) { v[2 * e] += bf2f((bf16_t)(tw[e] & 0xffff)); v[2 * e + 1] += bf2f((bf16_t)(tw[e] >> 16)); }
  }
  if (g.act == PCM_ACT_SILU) {
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = silu_f(v[e]);
  }
  if (g.res) {
    const uint4 t = *(const uint4*)(g.res + (size_t)m * g.ldr + n);
    const unsigned tw[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int e = 0; e < 4; e++) { v[2 * e] += bf2f((bf16_t)(tw[e] & 0xffff)); v[2 * e + 1] += bf2f((bf16_t)(tw[e] >> 16)); }
  }
  const uint4 o = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
  *(uint4*)((bf16_t*)g.out + (size_t)m * g.ldo + n) = o;
  if (g.out2) *(uint4*)(g.out2 + (size_t)m * g.ldo2 + n) = o;
}

// The same epilogue split into a LOAD half and a FINISH half so that a kernel can put the global loads (row vector, residual) of several
// 8-channel pieces in flight before it consumes the first one (gemm8p.hip: 4-5 pieces per thread per pass; issued one after the other
// each piece pays the full load latency).
struct EpiAux { uint4 rowvec, res; };
__device__ __forceinline__ EpiAux pcm_epi_load8(const GemmDev& g, int m, int n) {
  EpiAux x;
  x.rowvec = g.rowvec ? *(const uint4*)(g.rowvec + (size_t)(m / g.rpb) * g.N + n) : make_uint4(0u, 0u, 0u, 0u);
  x.res = g.res ? *(const uint4*)(g.res + (size_t)m * g.ldr + n) : make_uint4(0u, 0u, 0u, 0u);
  return x;
}
__device__ __forceinline__ void pcm_epi_finish8(const GemmDev& g, int m, int n, float v[8], const EpiAux& x) {
  if (g.bias) {
    const float4 b0 = *(const float4*)(g.bias + n), b1 = *(const float4*)(g.bias + n + 4);
    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
  }
  if (g.rowvec) {
    const unsigned tw[4] = {x.rowvec.x, x.rowvec.y, x.rowvec.z, x.rowvec.w};
#pragma unroll
    for (int e = 0; e < 4; e++) { v[2 * e] += bf2f((bf16_t)(tw[e] & 0xffff)); v[2 * e + 1] += bf2f((bf16_t)(tw[e] >> 16)); }
  }
  if (g.act == PCM_ACT_SILU) {
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = silu_f(v[e]);
  }
  if (g.res) {
    const unsigned tw[4] = {x.res.x, x.res.y, x.res.z, x.res.w};
#pragma unroll
    for (int e = 0; e < 4; e++) { v[2 * e] += bf2f((bf16_t)(tw[e] & 0xffff)); v[2 * e + 1] += bf2f((bf16_t)(tw[e] >> 16)); }
  }
  const uint4 o = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
  *(uint4*)((bf16_t*)g.out + (size_t)m * g.ldo + n) = o;
  if (g.out2) *(uint4*)(g.out2 + (size_t)m * g.ldo2 + n) = o;
}

// 256 x (64*FN) phased kernel (gemm8p.hip).  fn = 5 -> 256x320, fn = 4 -> 256x256.  grid = (tiles_m*tiles_n, splitk)
int pcm_gemm8p_launch(const GemmDev& g, int fn, void* stream);
size_t pcm_gemm8p_lds_bytes(int fn);
// 128 x (64*FN) short-K kernel at two workgroups per CU (gemm4w.hip).  grid = tiles_m*tiles_n (tiles_m counts 128-row tiles); no split-K
int pcm_gemm4w_launch(const GemmDev& g, int fn, void* stream);
// weights-stationary kernel for the short-K projections of the 64x64 level (gemm_ws.hip): N % 320 == 0, K total 320 / 384, plain segments
int pcm_gemm_ws_launch(const GemmDev& g, void* stream, int form);      // form 1: four waves x 80 columns; 2: eight waves x 40 columns
// rank-64 down-projection of a conv LoRA factor from a halo window (conv_r64.hip): 0 = launched, 1 = not one of its shapes, < 0 error
int pcm_conv_r64_launch(const GemmDev& g, void* stream, bool plan_only = false);
// streaming kernel for the rank-64 projections (gemm_n64.hip)
int pcm_gemm_n64_launch(const GemmDev& g, void* stream);
// weight-streaming kernel for the batch-row projections, M <= 16 (gemm_smallm.hip)
int pcm_gemm_smallm_launch(const GemmDev& g, void* stream);
