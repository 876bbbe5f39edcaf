// GroupNorm(32)(+SiLU) and LayerNorm on channels-last bf16 activations, forward and input-gradient
// (norm affine parameters are frozen in PCM-LoRA distillation, so no dgamma/dbeta).
// HBM-bound: every pass streams 16 B/lane; statistics are reduced in fp32 per thread, per block in
// LDS and across blocks with fp64 atomics (E[x^2]-E[x]^2 is then safe), so one launch covers all
// (batch, group) pairs with >> 256 workgroups.
#include "pcm_common.h"

// ------------------------------------------------------------------------------------------
// group statistics.  MODE 0: (sum x, sum x^2).  MODE 1 (backward): (sum dz*gamma, sum dz*gamma*xhat)
// thread -> fixed 8-channel vector cv, strided over pixels; blockDim.x = CVL * k.
// ------------------------------------------------------------------------------------------
struct GNArgs {
  const bf16_t* x;
  const bf16_t* dy;
  const double* stats;
  const float* gamma;
  const float* beta;
  double* out;
  double* part;   // != nullptr: contention-free mode, partial sums [b][chunk][g][2] instead of atomics into out
  int HW, C, G, cpg, CVL, csplit, ppb, act;
  float eps;
};

__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
  f[0] = bf2f((bf16_t)(v.x & 0xffff)); f[1] = bf2f((bf16_t)(v.x >> 16));
  f[2] = bf2f((bf16_t)(v.y & 0xffff)); f[3] = bf2f((bf16_t)(v.y >> 16));
  f[4] = bf2f((bf16_t)(v.z & 0xffff)); f[5] = bf2f((bf16_t)(v.z >> 16));
  f[6] = bf2f((bf16_t)(v.w & 0xffff)); f[7] = bf2f((bf16_t)(v.w >> 16));
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7]));
}

__device__ __forceinline__ float gn_act(float z, int act) {
  return act == PCM_ACT_SILU ? silu_f(z) : (act == PCM_ACT_LEAKY ? (z > 0.f ? z : 0.01f * z) : z);
}
__device__ __forceinline__ float gn_act_grad(float z, int act) {
  return act == PCM_ACT_SILU ? silu_grad_f(z) : (act == PCM_ACT_LEAKY ? (z > 0.f ? 1.f : 0.01f) : 1.f);
}

template <int MODE>
__global__ __launch_bounds__(256) void gn_stats_kernel(GNArgs a) {
  __shared__ float red[2][2048];                                   // per-channel sums of this block's channel slice (<= 256 vectors)
  __shared__ __attribute__((aligned(16))) float4 pbuf[4][256];     // per-thread partials, [float4 index][thread]: conflict-free b128 traffic
  const int b = blockIdx.y, zc = blockIdx.z;      // batch, channel split
  const int cvl = threadIdx.x % a.CVL, pl = threadIdx.x / a.CVL, k = blockDim.x / a.CVL;
  const int c0 = (zc * a.CVL + cvl) * 8;          // first channel of this thread's vector
  const int Cl = a.CVL * 8;                        // channels in this block's slice
  float ga[8], be[8], mu[8], rs[8];
  if (MODE == 1) {
    __shared__ float gconst[2][32];                // mean / rstd of the groups of this slice (fp64 math once per group, not per thread)
    const int gl0 = Cl / a.cpg;
    if ((int)threadIdx.x < gl0) {
      const int g = zc * gl0 + threadIdx.x;
      const double n = (double)a.HW * a.cpg;
      double s = a.stats[((size_t)b * a.G + g) * 2], ss = a.stats[((size_t)b * a.G + g) * 2 + 1];
      double m = s / n, var = ss / n - m * m;
      gconst[0][threadIdx.x] = (float)m;
      gconst[1][threadIdx.x] = (float)(1.0 / sqrt((var > 0 ? var : 0) + (double)a.eps));
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; e++) {
      int c = c0 + e, gloc = (cvl * 8 + e) / a.cpg;
      mu[e] = gconst[0][gloc]; rs[e] = gconst[1][gloc];
      ga[e] = a.gamma[c]; be[e] = a.beta[c];
    }
  }
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; e++) { s1[e] = 0.f; s2[e] = 0.f; }
  const int p_begin = blockIdx.x * a.ppb;
  int p_end = p_begin + a.ppb; if (p_end > a.HW) p_end = a.HW;
  const bf16_t* xb = a.x + (size_t)b * a.HW * a.C + c0;
  const bf16_t* dyb = MODE == 1 ? a.dy + (size_t)b * a.HW * a.C + c0 : nullptr;
  // several pixels per trip: the loads of a trip are issued together (clamped addresses, masked accumulation) -- with one load
  // per thread in flight this pass ran at ~2.3 TB/s
  constexpr int U = MODE == 0 ? 8 : 4;   // 8 x 16 B in flight per thread either way
  for (int p = p_begin + pl; p < p_end; p += U * k) {
    uint4 xr[U], dr[MODE == 1 ? U : 1];
#pragma unroll
    for (int u = 0; u < U; u++) {
      int pp = p + u * k; if (pp > p_end - 1) pp = p_end - 1;
      xr[u] = *(const uint4*)(xb + (size_t)pp * a.C);
      if (MODE == 1) dr[u] = *(const uint4*)(dyb + (size_t)pp * a.C);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (p + u * k >= p_end) continue;
      float xv[8];
      unpack8(xr[u], xv);
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < 8; e++) { s1[e] += xv[e]; s2[e] += xv[e] * xv[e]; }
      } else {
        float dv[8];
        unpack8(dr[u], dv);
#pragma unroll
        for (int e = 0; e < 8; e++) {
          float xh = (xv[e] - mu[e]) * rs[e];
          float dz = dv[e];
          dz *= gn_act_grad(xh * ga[e] + be[e], a.act);
          float t = dz * ga[e];
          s1[e] += t; s2[e] += t * xh;
        }
      }
    }
  }
  // block reduction over the k pixel lanes of each channel vector, without LDS atomics (256 threads x 16 ds_add_f32 with k-way
  // address conflicts cost ~3 us per block and bounded the whole pass by the LDS pipe): every thread parks its 16 sums as 4 float4,
  // then CVL*4 (vector, float4) pairs are summed over the pixel lanes
  pbuf[0][threadIdx.x] = make_float4(s1[0], s1[1], s1[2], s1[3]);
  pbuf[1][threadIdx.x] = make_float4(s1[4], s1[5], s1[6], s1[7]);
  pbuf[2][threadIdx.x] = make_float4(s2[0], s2[1], s2[2], s2[3]);
  pbuf[3][threadIdx.x] = make_float4(s2[4], s2[5], s2[6], s2[7]);
  __syncthreads();
  for (int t2 = threadIdx.x; t2 < 4 * a.CVL; t2 += blockDim.x) {
    const int cv2 = t2 % a.CVL, j = t2 / a.CVL;
    float4 acc4 = pbuf[j][cv2];
    for (int q = 1; q < k; q++) {
      float4 v = pbuf[j][q * a.CVL + cv2];
      acc4.x += v.x; acc4.y += v.y; acc4.z += v.z; acc4.w += v.w;
    }
    *(float4*)&red[j >> 1][cv2 * 8 + 4 * (j & 1)] = acc4;
  }
  __syncthreads();
  const int gl = Cl / a.cpg;  // groups in this slice
  if ((int)threadIdx.x < gl) {
    float t1 = 0.f, t2 = 0.f;
    for (int i = 0; i < a.cpg; i++) { t1 += red[0][threadIdx.x * a.cpg + i]; t2 += red[1][threadIdx.x * a.cpg + i]; }
    int g = zc * gl + threadIdx.x;
    if (a.part) {   // exactly one block owns (b, chunk, g)
      double* pp = a.part + (((size_t)b * gridDim.x + blockIdx.x) * a.G + g) * 2;
      pp[0] = (double)t1; pp[1] = (double)t2;
    } else {
      atomicAdd(&a.out[((size_t)b * a.G + g) * 2], (double)t1);
      atomicAdd(&a.out[((size_t)b * a.G + g) * 2 + 1], (double)t2);
    }
  }
}

// stats[b][i] = sum over chunks of part[b][chunk][i], i in [0, 2G): one 256-thread block per sample, 4 chunk lanes per output with
// all of a lane's loads issued together (the partials were just written by other XCDs, every load is a trip to memory)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* part, double* out, int nchunks, int n2g) {
  __shared__ double sm[4][64];
  const int b = blockIdx.x, i = threadIdx.x & 63, lane = threadIdx.x >> 6;
  double acc = 0.0;
  if (i < n2g) {
    const double* p = part + (size_t)b * nchunks * n2g + i;
    for (int c0 = lane; c0 < nchunks; c0 += 32) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        int ch = c0 + 4 * u;
        v[u] = p[(size_t)(ch < nchunks ? ch : c0) * n2g];
        if (ch >= nchunks) v[u] = 0.0;
      }
      acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
  }
  sm[lane][i] = acc;
  __syncthreads();
  if (lane == 0 && i < n2g) out[(size_t)b * n2g + i] = (sm[0][i] + sm[1][i]) + (sm[2][i] + sm[3][i]);
}

// apply.  MODE 0: y = act(xhat*gamma+beta).  MODE 1: dx = rstd*(dz*gamma - g1/n - xhat*g2/n)
struct GNApply {
  const bf16_t* x;
  const bf16_t* dy;
  const double* stats;
  const double* chstats;   // MODE 0, abi 5: per-(sample, channel) {sum, sumsq} from the producing GEMM's epilogue INSTEAD of `stats`; the group sums
  double* stats_out;       // are formed here and block 0 of each sample also writes them to stats_out[b][g][2] (what the backward reads)
  const double* chstats2;  // channels [c_split, C) come from a second producer (x = torch.cat([h, skip], dim=1): discriminator_sd15.py:312-342)
  int c_split, ld1, ld2;   // per-sample strides (channels) of the two sources
  const double* bstats;
  const float* gamma;
  const float* beta;
  bf16_t* y;
  const bf16_t* dres;      // MODE 1, abi 5: dx += dres (the gradient arriving over the block's skip path: ResnetBlock2D / Transformer2DModel residual)
  int HW, C, G, cpg, act, vec_per_block;
  float eps;
};

template <int MODE>
__global__ __launch_bounds__(256) void gn_apply_kernel(GNApply a) {
  // per-channel constants in LDS.  MODE 0: y = act(x*sc + sh) with sc = rstd*gamma, sh = beta - mean*sc.
  // MODE 1: with z = x*sc + sh, dz = dy*act'(z), R = rstd^2*g2, Q = rstd*g1 - R*mean:  dx = sc*dz - Q - R*x
  // (algebraically rstd*(dz*gamma - g1 - xhat*g2)); no per-element group lookup or division is left in the stream.
  PCM_DYN_SMEM(smem_raw);
  float* sc = (float*)smem_raw;
  float* sh = sc + a.C;
  float* qq = sh + a.C;
  float* rq = qq + a.C;
  __shared__ float gm[32], gr[32], g1[32], g2[32];
  const int b = blockIdx.y;
  const double n = (double)a.HW * a.cpg;
  __shared__ double gsum[32][8][2];
  if (MODE == 0 && a.chstats) {
    // group sums from the per-channel sums: 8 threads per group, each cpg/8 channels (<= 10 loads of 16 B, L2-resident), partials through LDS
    const int g = threadIdx.x >> 3, j = threadIdx.x & 7;
    if (g < a.G) {
      // (all of a thread's <= 10 loads are issued before the first is used: one round trip, not ten)
      struct __attribute__((aligned(16))) D2 { double x, y; };
      D2 vv[10];
#pragma unroll
      for (int u = 0; u < 10; u++) {
        const int cc = j + 8 * u;
        const int c = g * a.cpg + (cc < a.cpg ? cc : 0);      // (clamped lanes re-read the group's first channel; their value is not added)
        const double* p = c < a.c_split ? a.chstats + ((size_t)b * a.ld1 + c) * 2 : a.chstats2 + ((size_t)b * a.ld2 + (c - a.c_split)) * 2;
        vv[u] = *(const D2*)p;
      }
      double s = 0.0, ss = 0.0;
#pragma unroll
      for (int u = 0; u < 10; u++)
        if (j + 8 * u < a.cpg) { s += vv[u].x; ss += vv[u].y; }
      gsum[g][j][0] = s; gsum[g][j][1] = ss;
    }
    __syncthreads();
  }
  if ((int)threadIdx.x < a.G) {
    int g = threadIdx.x;
    double s, ss;
    if (MODE == 0 && a.chstats) {
      s = ss = 0.0;
#pragma unroll
      for (int j = 0; j < 8; j++) { s += gsum[g][j][0]; ss += gsum[g][j][1]; }
      if (blockIdx.x == 0 && a.stats_out) { a.stats_out[((size_t)b * a.G + g) * 2] = s; a.stats_out[((size_t)b * a.G + g) * 2 + 1] = ss; }
    } else {
      s = a.stats[((size_t)b * a.G + g) * 2]; ss = a.stats[((size_t)b * a.G + g) * 2 + 1];
    }
    double m = s / n, var = ss / n - m * m;
    gm[g] = (float)m;
    gr[g] = (float)(1.0 / sqrt((var > 0 ? var : 0) + (double)a.eps));
    if (MODE == 1) {
      g1[g] = (float)(a.bstats[((size_t)b * a.G + g) * 2] / n);
      g2[g] = (float)(a.bstats[((size_t)b * a.G + g) * 2 + 1] / n);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
    int g = c / a.cpg;
    float s_ = gr[g] * a.gamma[c];
    sc[c] = s_; sh[c] = a.beta[c] - gm[g] * s_;
    if (MODE == 1) {
      float R = gr[g] * gr[g] * g2[g];
      rq[c] = R; qq[c] = gr[g] * g1[g] - R * gm[g];
    }
  }
  __syncthreads();
  const int CV = a.C / 8;
  const int nvec = a.HW * CV;
  const int v0 = blockIdx.x * a.vec_per_block;
  int v1 = v0 + a.vec_per_block; if (v1 > nvec) v1 = nvec;
  const bf16_t* xb = a.x + (size_t)b * a.HW * a.C;
  bf16_t* yb = a.y + (size_t)b * a.HW * a.C;
  const bf16_t* dyb = MODE == 1 ? a.dy + (size_t)b * a.HW * a.C : nullptr;
  const bf16_t* rb = (MODE == 1 && a.dres) ? a.dres + (size_t)b * a.HW * a.C : nullptr;
  // channel-vector index of this thread's vector, advanced incrementally (stride 256 vectors): one modulo per thread
  const int cstep = 256 % CV;
  int cv = (v0 + (int)threadIdx.x) % CV;
  // 4 vectors per trip, loads issued together (clamped addresses; the store is skipped for the clamped duplicates)
  for (int vb = v0 + threadIdx.x; vb < v1; vb += 4 * 256) {
    uint4 xr[4], dr[4], rr[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      int v = vb + u * 256; if (v > v1 - 1) v = v1 - 1;
      xr[u] = *(const uint4*)(xb + (size_t)v * 8);
      if (MODE == 1) dr[u] = *(const uint4*)(dyb + (size_t)v * 8);
      if (MODE == 1 && rb) rr[u] = *(const uint4*)(rb + (size_t)v * 8);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int v = vb + u * 256;
      const int c0 = cv * 8;
      cv += cstep; if (cv >= CV) cv -= CV;
      if (v >= v1) continue;
      float xv[8], o[8], scv[8], shv[8];
      unpack8(xr[u], xv);
      *(float4*)&scv[0] = *(const float4*)&sc[c0]; *(float4*)&scv[4] = *(const float4*)&sc[c0 + 4];
      *(float4*)&shv[0] = *(const float4*)&sh[c0]; *(float4*)&shv[4] = *(const float4*)&sh[c0 + 4];
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = gn_act(fmaf(xv[e], scv[e], shv[e]), a.act);
      } else {
        float dv[8], qv[8], rv[8];
        unpack8(dr[u], dv);
        *(float4*)&qv[0] = *(const float4*)&qq[c0]; *(float4*)&qv[4] = *(const float4*)&qq[c0 + 4];
        *(float4*)&rv[0] = *(const float4*)&rq[c0]; *(float4*)&rv[4] = *(const float4*)&rq[c0 + 4];
#pragma unroll
        for (int e = 0; e < 8; e++) {
          float dz = dv[e] * gn_act_grad(fmaf(xv[e], scv[e], shv[e]), a.act);
          o[e] = fmaf(scv[e], dz, -fmaf(rv[e], xv[e], qv[e]));
        }
        if (rb) {      // (the sum is rounded ONCE: dx = bf16(gn_dx + dres) instead of bf16(bf16(gn_dx) + dres) of the separate add pass)
          float rf[8];
          unpack8(rr[u], rf);
#pragma unroll
          for (int e = 0; e < 8; e++) o[e] += rf[e];
        }
      }
      *(uint4*)(yb + (size_t)v * 8) = pack8(o);
    }
  }
}

static int gn_check(const char* what, int B, int HW, int C, int G) {
  PCM_CHECK(B > 0 && HW > 0 && C > 0 && G > 0 && G <= 32 && (C % G) == 0 && (C % 8) == 0 && C <= 2560 * 8, PCM_EINVAL,
            "%s: need G<=32, C%%G==0, C%%8==0 (B=%d HW=%d C=%d G=%d)", what, B, HW, C, G);
  return PCM_OK;
}

// launch geometry of the statistics pass: channel split so a block's slice has <= 256 vectors and whole groups; `target` blocks
static int gn_stats_geometry(const char* what, GNArgs& a, int B, int target, int* split_, int* chunks_, int* threads_) {
  int CV = a.C / 8;
  int split = 1;
  while (CV / split > 256 || (CV % split) != 0 || ((a.C / split) % a.cpg) != 0) {
    split++;
    PCM_CHECK(split <= a.G, PCM_EUNSUPPORTED, "%s: cannot split C=%d over groups of %d", what, a.C, a.cpg);
  }
  a.CVL = CV / split; a.csplit = split;
  PCM_CHECK(a.CVL * 8 <= 2560, PCM_EUNSUPPORTED, "%s: channel slice too large", what);
  int k = 256 / a.CVL; if (k < 1) k = 1;
  int chunks = (target + B * split - 1) / (B * split);
  int maxchunks = (a.HW + k - 1) / k; if (chunks > maxchunks) chunks = maxchunks; if (chunks < 1) chunks = 1;
  a.ppb = (a.HW + chunks - 1) / chunks;
  chunks = (a.HW + a.ppb - 1) / a.ppb;
  *split_ = split; *chunks_ = chunks; *threads_ = a.CVL * k;
  return PCM_OK;
}

// atomic mode: each block ends with LDS + fp64 global atomics (same-address fp64 atomics serialize at ~0.5 us each on this part):
// ~2 blocks per CU, long pixel runs per thread.  Workspace mode: no atomics, any batch size gets ~1024 blocks (measured flat from 512 to 2048).
PCM_KNOB int g_gn_target_ws = 1024;
PCM_TOOLS_ONLY(extern "C" void pcm_debug_gn_target(int blocks) { g_gn_target_ws = blocks > 0 ? blocks : 1024; })   // tuning hook (tools/gn_probe.py)
#define GN_TARGET_ATOMIC PCM_GRID_CAP(512)
#define GN_TARGET_WS PCM_GRID_CAP(g_gn_target_ws)

template <int MODE>
static int gn_stats_launch(const char* what, GNArgs a, int B, void* stream, bool zero = true, void* ws = nullptr, size_t ws_bytes = 0) {
  int split, chunks, threads;
  if (int rc = gn_stats_geometry(what, a, B, ws ? GN_TARGET_WS : GN_TARGET_ATOMIC, &split, &chunks, &threads)) return rc;
  if (ws) {
    PCM_CHECK(ws_bytes >= sizeof(double) * 2 * a.G * (size_t)B * chunks && ((uintptr_t)ws % 8) == 0, PCM_EINVAL, "%s: workspace too small", what);
    a.part = (double*)ws;
    PCM_LAUNCH((gn_stats_kernel<MODE>), dim3(chunks, B, split), dim3(threads), 0, stream, a);
    PCM_LAUNCH(gn_finalize_kernel, dim3(B), dim3(256), 0, stream, (const double*)ws, a.out, chunks, 2 * a.G);
    return pcm_post_launch(what);
  }
  a.part = nullptr;
  if (zero) pcm_zero_async(a.out, sizeof(double) * 2 * B * a.G, stream);
  PCM_LAUNCH((gn_stats_kernel<MODE>), dim3(chunks, B, split), dim3(threads), 0, stream, a);
  return pcm_post_launch(what);
}

extern "C" size_t pcm_groupnorm_workspace_bytes(int B, int HW, int C, int G) {
  if (B <= 0 || HW <= 0 || C <= 0 || G <= 0 || G > 32 || (C % G) || (C % 8)) return 0;
  GNArgs a; memset(&a, 0, sizeof(a));
  a.HW = HW; a.C = C; a.G = G; a.cpg = C / G;
  int split, chunks, threads;
  if (gn_stats_geometry("pcm_groupnorm_workspace_bytes", a, B, GN_TARGET_WS, &split, &chunks, &threads)) return 0;
  return sizeof(double) * 2 * G * (size_t)B * chunks;
}

extern "C" int pcm_groupnorm_stats_ws(const void* x, double* stats, int B, int HW, int C, int G, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  if (int rc = gn_check("pcm_groupnorm_stats_ws", B, HW, C, G)) return rc;
  PCM_CHECK(x && stats && workspace && PCM_ALIGNED16(x), PCM_EALIGN, "pcm_groupnorm_stats_ws: null/unaligned argument");
  GNArgs a; memset(&a, 0, sizeof(a));
  a.x = (const bf16_t*)x; a.out = stats; a.HW = HW; a.C = C; a.G = G; a.cpg = C / G;
  return gn_stats_launch<0>("pcm_groupnorm_stats_ws", a, B, stream, false, workspace, workspace_bytes);
}

extern "C" int pcm_groupnorm_bwd_stats_ws(const void* x, const void* dy, const double* stats, const float* gamma, const float* beta,
                                          double* bstats, int B, int HW, int C, int G, float eps, int act, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  if (int rc = gn_check("pcm_groupnorm_bwd_stats_ws", B, HW, C, G)) return rc;
  PCM_CHECK(x && dy && stats && gamma && beta && bstats && workspace && PCM_ALIGNED16(x) && PCM_ALIGNED16(dy), PCM_EALIGN,
            "pcm_groupnorm_bwd_stats_ws: null/unaligned argument");
  GNArgs a; memset(&a, 0, sizeof(a));
  a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.stats = stats; a.gamma = gamma; a.beta = beta;
  a.out = bstats; a.HW = HW; a.C = C; a.G = G; a.cpg = C / G; a.eps = eps; a.act = act;
  return gn_stats_launch<1>("pcm_groupnorm_bwd_stats_ws", a, B, stream, false, workspace, workspace_bytes);
}

extern "C" int pcm_groupnorm_stats(const void* x, double* stats, int B, int HW, int C, int G, void* stream) {
  if (int rc = gn_check("pcm_groupnorm_stats", B, HW, C, G)) return rc;
  PCM_CHECK(x && stats && PCM_ALIGNED16(x), PCM_EALIGN, "pcm_groupnorm_stats: x must be 16-byte aligned");
  GNArgs a; memset(&a, 0, sizeof(a));
  a.x = (const bf16_t*)x; a.out = stats; a.HW = HW; a.C = C; a.G = G; a.cpg = C / G;
  return gn_stats_launch<0>("pcm_groupnorm_stats", a, B, stream);
}

extern "C" int pcm_groupnorm_bwd_stats(const void* x, const void* dy, const double* stats, const float* gamma,
                                       const float* beta, double* bstats, int B, int HW, int C, int G,
                                       float eps, int act, void* stream) {
  if (int rc = gn_check("pcm_groupnorm_bwd_stats", B, HW, C, G)) return rc;
  PCM_CHECK(x && dy && stats && gamma && beta && bstats && PCM_ALIGNED16(x) && PCM_ALIGNED16(dy), PCM_EALIGN,
            "pcm_groupnorm_bwd_stats: null/unaligned argument");
  GNArgs a; memset(&a, 0, sizeof(a));
  a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.stats = stats; a.gamma = gamma; a.beta = beta;
  a.out = bstats; a.HW = HW; a.C = C; a.G = G; a.cpg = C / G; a.eps = eps; a.act = act;
  return gn_stats_launch<1>("pcm_groupnorm_bwd_stats", a, B, stream);
}

// _acc variants: the caller hands in an ALREADY ZEROED statistics buffer (one memset for a whole arena of layers instead of one per call)
extern "C" int pcm_groupnorm_stats_acc(const void* x, double* stats, int B, int HW, int C, int G, void* stream) {
  if (int rc = gn_check("pcm_groupnorm_stats", B, HW, C, G)) return rc;
  PCM_CHECK(x && stats && PCM_ALIGNED16(x), PCM_EALIGN, "pcm_groupnorm_stats: x must be 16-byte aligned");
  GNArgs a; memset(&a, 0, sizeof(a));
  a.x = (const bf16_t*)x; a.out = stats; a.HW = HW; a.C = C; a.G = G; a.cpg = C / G;
  return gn_stats_launch<0>("pcm_groupnorm_stats_acc", a, B, stream, false);
}

extern "C" int pcm_groupnorm_bwd_stats_acc(const void* x, const void* dy, const double* stats, const float* gamma,
                                       const float* beta, double* bstats, int B, int HW, int C, int G,
                                       float eps, int act, void* stream) {
  if (int rc = gn_check("pcm_groupnorm_bwd_stats", B, HW, C, G)) return rc;
  PCM_CHECK(x && dy && stats && gamma && beta && bstats && PCM_ALIGNED16(x) && PCM_ALIGNED16(dy), PCM_EALIGN,
            "pcm_groupnorm_bwd_stats: null/unaligned argument");
  GNArgs a; memset(&a, 0, sizeof(a));
  a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.stats = stats; a.gamma = gamma; a.beta = beta;
  a.out = bstats; a.HW = HW; a.C = C; a.G = G; a.cpg = C / G; a.eps = eps; a.act = act;
  return gn_stats_launch<1>("pcm_groupnorm_bwd_stats_acc", a, B, stream, false);
}

template <int MODE>
static int gn_apply_launch(const char* what, GNApply a, int B, void* stream) {
  size_t nvec = (size_t)a.HW * (a.C / 8);
  int blocks = (int)((nvec + 256 * 8 - 1) / (256 * 8));
  int cap = (PCM_GRID_CAP(4096) + B - 1) / B; if (blocks > cap) blocks = cap; if (blocks < 1) blocks = 1;
  a.vec_per_block = (int)((nvec + blocks - 1) / blocks);
  blocks = (int)((nvec + a.vec_per_block - 1) / a.vec_per_block);
  PCM_LAUNCH((gn_apply_kernel<MODE>), dim3(blocks, B), dim3(256), (MODE == 1 ? 4 : 2) * a.C * sizeof(float), stream, a);
  return pcm_post_launch(what);
}

extern "C" int pcm_groupnorm_apply(const void* x, const double* stats, const float* gamma, const float* beta,
                                   void* y, int B, int HW, int C, int G, float eps, int act, void* stream) {
  if (int rc = gn_check("pcm_groupnorm_apply", B, HW, C, G)) return rc;
  PCM_CHECK(x && stats && gamma && beta && y && PCM_ALIGNED16(x) && PCM_ALIGNED16(y) && C <= 2560, PCM_EALIGN,
            "pcm_groupnorm_apply: null/unaligned argument or C>2560");
  GNApply a; memset(&a, 0, sizeof(a));
  a.x = (const bf16_t*)x; a.stats = stats; a.gamma = gamma; a.beta = beta; a.y = (bf16_t*)y;
  a.HW = HW; a.C = C; a.G = G; a.cpg = C / G; a.act = act; a.eps = eps;
  return gn_apply_launch<0>("pcm_groupnorm_apply", a, B, stream);
}

/* abi 5: the same apply with the statistics taken from per-(sample, channel) sums accumulated by the producing contractions' epilogues
 * (pcm_gemm_epi.chstats); also writes the group statistics to stats_out[B][G][2] for the backward */
extern "C" int pcm_groupnorm_apply_chstats(const void* x, const double* chstats, int stats_ld, const double* chstats2, int stats_ld2, int c_split,
                                           double* stats_out, const float* gamma, const float* beta, void* y, int B, int HW, int C, int G,
                                           float eps, int act, void* stream) {
  if (int rc = gn_check("pcm_groupnorm_apply_chstats", B, HW, C, G)) return rc;
  PCM_CHECK(x && chstats && gamma && beta && y && PCM_ALIGNED16(x) && PCM_ALIGNED16(y) && ((uintptr_t)chstats % 8) == 0 && C <= 2560, PCM_EALIGN,
            "pcm_groupnorm_apply_chstats: null/unaligned argument or C>2560");
  if (!chstats2) c_split = C;
  PCM_CHECK(c_split > 0 && c_split <= C && stats_ld >= c_split && (!chstats2 || (stats_ld2 >= C - c_split && ((uintptr_t)chstats2 % 8) == 0)), PCM_EINVAL,
            "pcm_groupnorm_apply_chstats: bad channel split / strides (C=%d split=%d ld=%d ld2=%d)", C, c_split, stats_ld, stats_ld2);
  GNApply a; memset(&a, 0, sizeof(a));
  a.x = (const bf16_t*)x; a.chstats = chstats; a.chstats2 = chstats2; a.c_split = c_split; a.ld1 = stats_ld; a.ld2 = stats_ld2;
  a.stats_out = stats_out; a.gamma = gamma; a.beta = beta; a.y = (bf16_t*)y;
  a.HW = HW; a.C = C; a.G = G; a.cpg = C / G; a.act = act; a.eps = eps;
  return gn_apply_launch<0>("pcm_groupnorm_apply_chstats", a, B, stream);
}

extern "C" int pcm_groupnorm_bwd_apply_res(const void* x, const void* dy, const double* stats, const double* bstats,
                                           const float* gamma, const float* beta, const void* dres, void* dx, int B, int HW, int C,
                                           int G, float eps, int act, void* stream) {
  if (int rc = gn_check("pcm_groupnorm_bwd_apply", B, HW, C, G)) return rc;
  PCM_CHECK(x && dy && stats && bstats && gamma && beta && dx && PCM_ALIGNED16(x) && PCM_ALIGNED16(dy) && (!dres || PCM_ALIGNED16(dres)) &&
                PCM_ALIGNED16(dx) && C <= 2560, PCM_EALIGN, "pcm_groupnorm_bwd_apply: null/unaligned argument or C>2560");
  GNApply a; memset(&a, 0, sizeof(a));
  a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.stats = stats; a.bstats = bstats; a.gamma = gamma; a.dres = (const bf16_t*)dres;
  a.beta = beta; a.y = (bf16_t*)dx; a.HW = HW; a.C = C; a.G = G; a.cpg = C / G; a.act = act; a.eps = eps;
  return gn_apply_launch<1>("pcm_groupnorm_bwd_apply", a, B, stream);
}
extern "C" int pcm_groupnorm_bwd_apply(const void* x, const void* dy, const double* stats, const double* bstats,
                                       const float* gamma, const float* beta, void* dx, int B, int HW, int C,
                                       int G, float eps, int act, void* stream) {
  return pcm_groupnorm_bwd_apply_res(x, dy, stats, bstats, gamma, beta, nullptr, dx, B, HW, C, G, eps, act, stream);
}

// ------------------------------------------------------------------------------------------
// LayerNorm over the last dim: one wave per row, 16 B/lane, row kept in registers (C <= 1536*... see VPL)
// ------------------------------------------------------------------------------------------
template <int VPL, int R>  // 16-byte vectors per lane: C <= 512*VPL; R rows per wave with all their loads issued together
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16_t* x, const float* gamma, const float* beta, bf16_t* y,
                                                     float* mean, float* rstd, int M, int C, float eps, int rpb) {
  const int lane = threadIdx.x & 63, row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= M) return;
  const int CV = C / 8;
  uint4 raw[R][VPL];
#pragma unroll
  for (int r = 0; r < R; r++) {
    int row = row0 + r; if (row > M - 1) row = M - 1;
#pragma unroll
    for (int i = 0; i < VPL; i++) {
      int cv = lane + 64 * i; if (cv > CV - 1) cv = CV - 1;      // clamped (unconditional) loads; lanes beyond CV are masked below
      raw[r][i] = *(const uint4*)(x + (size_t)row * C + cv * 8);
    }
  }
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int row = row0 + r;
    if (row >= M) break;
    float v[VPL][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; i++) {
      unpack8(raw[r][i], v[i]);
      if (lane + 64 * i < CV) {
#pragma unroll
        for (int e = 0; e < 8; e++) s += v[i][e];
      }
    }
    const float mu = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; i++)
      if (lane + 64 * i < CV) {
#pragma unroll
        for (int e = 0; e < 8; e++) { float d = v[i][e] - mu; q += d * d; }
      }
    const float rs = rsqrtf(wave_sum(q) / C + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int i = 0; i < VPL; i++) {
      int cv = lane + 64 * i;
      if (cv < CV) {
        float o[8];
        const size_t po = (rpb > 0 ? (size_t)(row / rpb) * C : 0) + cv * 8;   // rpb > 0: per-sample affine rows (adaLN modulation)
        float4 g0 = *(const float4*)(gamma + po), g1 = *(const float4*)(gamma + po + 4);
        float4 b0 = *(const float4*)(beta + po), b1 = *(const float4*)(beta + po + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = (v[i][e] - mu) * rs * gg[e] + bb[e];
        *(uint4*)(y + (size_t)row * C + cv * 8) = pack8(o);
      }
    }
  }
}

// dx = rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat)) (+ dres)
template <int VPL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* x, const bf16_t* dy, const float* gamma, const float* mean,
                                                     const float* rstd, const bf16_t* dres, bf16_t* dx, int M, int C, int rpb) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int CV = C / 8;
  const float mu = mean[row], rs = rstd[row];
  float xh[VPL][8], dg[VPL][8];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; i++) {
    int cv = lane + 64 * i;
    if (cv < CV) {
      float xv[8], dv[8];
      unpack8(*(const uint4*)(x + (size_t)row * C + cv * 8), xv);
      unpack8(*(const uint4*)(dy + (size_t)row * C + cv * 8), dv);
      const size_t po = (rpb > 0 ? (size_t)(row / rpb) * C : 0) + cv * 8;
      float4 g0 = *(const float4*)(gamma + po), g1 = *(const float4*)(gamma + po + 4);
      const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
      for (int e = 0; e < 8; e++) {
        xh[i][e] = (xv[e] - mu) * rs;
        dg[i][e] = dv[e] * gg[e];
        s1 += dg[i][e]; s2 += dg[i][e] * xh[i][e];
      }
    }
  }
  s1 = wave_sum(s1) / C; s2 = wave_sum(s2) / C;
#pragma unroll
  for (int i = 0; i < VPL; i++) {
    int cv = lane + 64 * i;
    if (cv < CV) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = rs * (dg[i][e] - s1 - xh[i][e] * s2);
      if (dres) {
        float r[8];
        unpack8(*(const uint4*)(dres + (size_t)row * C + cv * 8), r);
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] += r[e];
      }
      *(uint4*)(dx + (size_t)row * C + cv * 8) = pack8(o);
    }
  }
}

static int ln_fwd_launch(const char* what, const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int M,
                         int C, float eps, int rpb, void* stream) {
  PCM_CHECK(x && gamma && beta && y && mean && rstd && M > 0 && C > 0 && (C % 8) == 0 && C <= 2048 && rpb >= 0, PCM_EINVAL,
            "%s: need C%%8==0, C<=2048 (M=%d C=%d)", what, M, C);
  PCM_CHECK(PCM_ALIGNED16(x) && PCM_ALIGNED16(y) && PCM_ALIGNED16(gamma) && PCM_ALIGNED16(beta), PCM_EALIGN, "%s: alignment", what);
  dim3 block(256);
  if (C <= 512) PCM_LAUNCH((ln_fwd_kernel<1, 4>), dim3((M + 15) / 16), block, 0, stream, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, M, C, eps, rpb);
  else if (C <= 1024) PCM_LAUNCH((ln_fwd_kernel<2, 4>), dim3((M + 15) / 16), block, 0, stream, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, M, C, eps, rpb);
  else PCM_LAUNCH((ln_fwd_kernel<4, 2>), dim3((M + 7) / 8), block, 0, stream, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, M, C, eps, rpb);
  return pcm_post_launch(what);
}
static int ln_bwd_launch(const char* what, const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd,
                         const void* dres, void* dx, int M, int C, int rpb, void* stream) {
  PCM_CHECK(x && dy && gamma && mean && rstd && dx && M > 0 && C > 0 && (C % 8) == 0 && C <= 2048 && rpb >= 0, PCM_EINVAL,
            "%s: need C%%8==0, C<=2048 (M=%d C=%d)", what, M, C);
  PCM_CHECK(PCM_ALIGNED16(x) && PCM_ALIGNED16(dy) && PCM_ALIGNED16(dx) && PCM_ALIGNED16(gamma) && (!dres || PCM_ALIGNED16(dres)), PCM_EALIGN, "%s: alignment", what);
  dim3 grid((M + 3) / 4), block(256);
  if (C <= 512) PCM_LAUNCH((ln_bwd_kernel<1>), grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)dy, gamma, mean, rstd, (const bf16_t*)dres, (bf16_t*)dx, M, C, rpb);
  else if (C <= 1024) PCM_LAUNCH((ln_bwd_kernel<2>), grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)dy, gamma, mean, rstd, (const bf16_t*)dres, (bf16_t*)dx, M, C, rpb);
  else PCM_LAUNCH((ln_bwd_kernel<4>), grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)dy, gamma, mean, rstd, (const bf16_t*)dres, (bf16_t*)dx, M, C, rpb);
  return pcm_post_launch(what);
}

extern "C" int pcm_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                 float* rstd, int M, int C, float eps, void* stream) {
  return ln_fwd_launch("pcm_layernorm_fwd", x, gamma, beta, y, mean, rstd, M, C, eps, 0, stream);
}
extern "C" int pcm_layernorm_bwd(const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd,
                                 const void* dres, void* dx, int M, int C, void* stream) {
  return ln_bwd_launch("pcm_layernorm_bwd", x, dy, gamma, mean, rstd, dres, dx, M, C, 0, stream);
}
// adaLN (AdaLayerNormZero / AdaLayerNormContinuous of the MMDiT blocks): LayerNorm without learned affine followed by a per-SAMPLE
// modulation y = xhat * gamma[b] + beta[b] with gamma = 1 + scale, beta = shift ([B][C] fp32, rows_per_batch rows share one pair)
extern "C" int pcm_layernorm_mod_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int M, int C,
                                     float eps, int rows_per_batch, void* stream) {
  PCM_CHECK(rows_per_batch > 0 && (M % rows_per_batch) == 0, PCM_EINVAL, "pcm_layernorm_mod_fwd: M must be B*rows_per_batch");
  return ln_fwd_launch("pcm_layernorm_mod_fwd", x, gamma, beta, y, mean, rstd, M, C, eps, rows_per_batch, stream);
}
extern "C" int pcm_layernorm_mod_bwd(const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd, const void* dres,
                                     void* dx, int M, int C, int rows_per_batch, void* stream) {
  PCM_CHECK(rows_per_batch > 0 && (M % rows_per_batch) == 0, PCM_EINVAL, "pcm_layernorm_mod_bwd: M must be B*rows_per_batch");
  return ln_bwd_launch("pcm_layernorm_mod_bwd", x, dy, gamma, mean, rstd, dres, dx, M, C, rows_per_batch, stream);
}

// ---- GroupNorm affine-parameter gradients (discriminator heads: their norms ARE trainable) ----
// dgamma[c] += sum_{b,hw} dz * xhat ; dbeta[c] += sum dz ; dz = dy * act'(z)
__global__ __launch_bounds__(256) void gn_param_grad_kernel(GNArgs a, float* dgamma, float* dbeta) {
  const int b = blockIdx.y, zc = blockIdx.z;
  const int cvl = threadIdx.x % a.CVL, pl = threadIdx.x / a.CVL, k = blockDim.x / a.CVL;
  const int c0 = (zc * a.CVL + cvl) * 8;
  float ga[8], be[8], mu[8], rs[8], s1[8], s2[8];
  const double n = (double)a.HW * a.cpg;
#pragma unroll
  for (int e = 0; e < 8; e++) {
    int c = c0 + e, g = c / a.cpg;
    double s = a.stats[((size_t)b * a.G + g) * 2], ss = a.stats[((size_t)b * a.G + g) * 2 + 1];
    double m = s / n, var = ss / n - m * m;
    mu[e] = (float)m; rs[e] = (float)(1.0 / sqrt((var > 0 ? var : 0) + (double)a.eps));
    ga[e] = a.gamma[c]; be[e] = a.beta[c]; s1[e] = 0.f; s2[e] = 0.f;
  }
  const int p_begin = blockIdx.x * a.ppb;
  int p_end = p_begin + a.ppb; if (p_end > a.HW) p_end = a.HW;
  const bf16_t* xb = a.x + (size_t)b * a.HW * a.C + c0;
  const bf16_t* dyb = a.dy + (size_t)b * a.HW * a.C + c0;
  auto add_row = [&](const uint4& xr, const uint4& dr) {
    float xv[8], dv[8];
    unpack8(xr, xv);
    unpack8(dr, dv);
#pragma unroll
    for (int e = 0; e < 8; e++) {
      float xh = (xv[e] - mu[e]) * rs[e];
      float dz = dv[e] * gn_act_grad(xh * ga[e] + be[e], a.act);
      s1[e] += dz * xh; s2[e] += dz;
    }
  };
  int p = p_begin + pl;
  for (; p + 2 * k < p_end; p += 3 * k) {       // three rows (six loads) in flight per thread; same summation order as one row at a time
    uint4 xr[3], dr[3];
#pragma unroll
    for (int u = 0; u < 3; u++) { xr[u] = *(const uint4*)(xb + (size_t)(p + u * k) * a.C); dr[u] = *(const uint4*)(dyb + (size_t)(p + u * k) * a.C); }
#pragma unroll
    for (int u = 0; u < 3; u++) add_row(xr[u], dr[u]);
  }
  for (; p < p_end; p += k) add_row(*(const uint4*)(xb + (size_t)p * a.C), *(const uint4*)(dyb + (size_t)p * a.C));
  // the k pixel-lanes of the block are reduced in LDS first: ONE atomic per (block, channel, parameter).  Per-thread atomics put
  // k x blocks same-address operations on every channel (measured on the 36-head discriminator step: 320 us per call, 23 ms per step).
  __shared__ float red[256 * 16];
#pragma unroll
  for (int e = 0; e < 8; e++) { red[(pl * a.CVL + cvl) * 16 + e] = s1[e]; red[(pl * a.CVL + cvl) * 16 + 8 + e] = s2[e]; }
  __syncthreads();
  for (int i = threadIdx.x; i < a.CVL * 16; i += blockDim.x) {
    float t = 0.f;
    for (int j = 0; j < k; j++) t += red[j * a.CVL * 16 + i];
    const int cv = i >> 4, e = i & 15;
    const int col = (zc * a.CVL + cv) * 8 + (e & 7);
    // reproducible form: block (chunk, b) of every channel split stores to part[(b * chunks + chunk)][dgamma C | dbeta C]
    if (a.part) ((float*)a.part)[((size_t)b * gridDim.x + blockIdx.x) * (2 * a.C) + (e < 8 ? 0 : a.C) + col] = t;
    else atomicAdd((e < 8 ? dgamma : dbeta) + col, t);
  }
}
static void gn_param_grad_geometry(GNArgs& a, int B, int* split_, int* k_, int* chunks_) {
  int CV = a.C / 8, split = 1;
  while (CV / split > 256 || (CV % split) != 0) split++;
  a.CVL = CV / split;
  int k = 256 / a.CVL; if (k < 1) k = 1;
  int chunks = (PCM_GRID_CAP(512) + B * split - 1) / (B * split);
  int maxchunks = (a.HW + k - 1) / k; if (chunks > maxchunks) chunks = maxchunks; if (chunks < 1) chunks = 1;
  a.ppb = (a.HW + chunks - 1) / chunks; chunks = (a.HW + a.ppb - 1) / a.ppb;
  *split_ = split; *k_ = k; *chunks_ = chunks;
}
extern "C" int pcm_groupnorm_param_grad(const void* x, const void* dy, const double* stats, const float* gamma, const float* beta,
                                        float* dgamma, float* dbeta, int B, int HW, int C, int G, float eps, int act, void* stream) {
  if (int rc = gn_check("pcm_groupnorm_param_grad", B, HW, C, G)) return rc;
  PCM_CHECK(x && dy && stats && gamma && beta && dgamma && dbeta && PCM_ALIGNED16(x) && PCM_ALIGNED16(dy), PCM_EALIGN, "pcm_groupnorm_param_grad: null/unaligned");
  GNArgs a; memset(&a, 0, sizeof(a));
  a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.stats = stats; a.gamma = gamma; a.beta = beta;
  a.HW = HW; a.C = C; a.G = G; a.cpg = C / G; a.eps = eps; a.act = act;
  int split, k, chunks;
  gn_param_grad_geometry(a, B, &split, &k, &chunks);
  a.part = nullptr;
  PCM_LAUNCH(gn_param_grad_kernel, dim3(chunks, B, split), dim3(a.CVL * k), 0, stream, a, dgamma, dbeta);
  return pcm_post_launch("pcm_groupnorm_param_grad");
}
// reproducible form (abi 5): per-block partials + an ordered finalize that ADDS into dgamma / dbeta like the atomic form
extern "C" size_t pcm_groupnorm_param_grad_workspace_bytes(int B, int HW, int C, int G) {
  if (B <= 0 || HW <= 0 || C <= 0 || G <= 0 || G > 32 || (C % G) || (C % 8)) return 0;
  GNArgs a; memset(&a, 0, sizeof(a));
  a.HW = HW; a.C = C; a.G = G; a.cpg = C / G;
  int split, k, chunks;
  gn_param_grad_geometry(a, B, &split, &k, &chunks);
  return sizeof(float) * (size_t)B * chunks * 2 * C;
}
extern "C" int pcm_groupnorm_param_grad_ws(const void* x, const void* dy, const double* stats, const float* gamma, const float* beta,
                                           float* dgamma, float* dbeta, int B, int HW, int C, int G, float eps, int act, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  if (int rc = gn_check("pcm_groupnorm_param_grad_ws", B, HW, C, G)) return rc;
  PCM_CHECK(x && dy && stats && gamma && beta && dgamma && dbeta && workspace && PCM_ALIGNED16(x) && PCM_ALIGNED16(dy), PCM_EALIGN, "pcm_groupnorm_param_grad_ws: null/unaligned");
  GNArgs a; memset(&a, 0, sizeof(a));
  a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.stats = stats; a.gamma = gamma; a.beta = beta;
  a.HW = HW; a.C = C; a.G = G; a.cpg = C / G; a.eps = eps; a.act = act;
  int split, k, chunks;
  gn_param_grad_geometry(a, B, &split, &k, &chunks);
  PCM_CHECK(workspace_bytes >= sizeof(float) * (size_t)B * chunks * 2 * C, PCM_EINVAL, "pcm_groupnorm_param_grad_ws: workspace too small");
  a.part = (double*)workspace;
  PCM_LAUNCH(gn_param_grad_kernel, dim3(chunks, B, split), dim3(a.CVL * k), 0, stream, a, dgamma, dbeta);
  pcm_partials_finalize((const float*)workspace, 2 * C, dgamma, B * chunks, C, 1, stream);
  pcm_partials_finalize((const float*)workspace + C, 2 * C, dbeta, B * chunks, C, 1, stream);
  return pcm_post_launch("pcm_groupnorm_param_grad_ws");
}
