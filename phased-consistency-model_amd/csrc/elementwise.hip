// Data-movement and small-operator kernels of the UNet wiring (channels-last bf16, 16 B/lane):
// nearest-2x upsample and its adjoint, channel concat/split for the skip connections, residual add,
// SiLU, GEGLU fwd/bwd, pixel-sum (gradient of the broadcast time-embedding add), the 4-channel
// edge convolutions conv_in / conv_out (+ conv_out input-gradient) and the sinusoidal timestep
// projection.  All HBM-bound; grid-stride with ~2048 workgroups.
#include "pcm_common.h"

__device__ __forceinline__ void ew_unpack8(const uint4& v, float (&f)[8]) {
  f[0] = bf2f((bf16_t)(v.x & 0xffff)); f[1] = bf2f((bf16_t)(v.x >> 16));
  f[2] = bf2f((bf16_t)(v.y & 0xffff)); f[3] = bf2f((bf16_t)(v.y >> 16));
  f[4] = bf2f((bf16_t)(v.z & 0xffff)); f[5] = bf2f((bf16_t)(v.z >> 16));
  f[6] = bf2f((bf16_t)(v.w & 0xffff)); f[7] = bf2f((bf16_t)(v.w >> 16));
}
__device__ __forceinline__ uint4 ew_pack8(const float (&f)[8]) {
  return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7]));
}
static inline int ew_blocks(long nvec) {
  long b = (nvec + 255) / 256;
  if (b > PCM_GRID_CAP(4096)) b = PCM_GRID_CAP(4096);
  if (b < 1) b = 1;
  return (int)b;
}
#define EW_LOOP(v, nvec) for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < (nvec); v += (long)gridDim.x * blockDim.x)

// ---- upsample / pool ----
__global__ __launch_bounds__(256) void upsample2x_kernel(const uint4* x, uint4* y, int B, int H, int W, int CV) {
  long nvec = (long)B * 4 * H * W * CV;
  EW_LOOP(v, nvec) {
    int cv = (int)(v % CV); long p = v / CV;
    int ox = (int)(p % (2 * W)); long q = p / (2 * W);
    int oy = (int)(q % (2 * H)); int b = (int)(q / (2 * H));
    y[v] = x[(((long)b * H + (oy >> 1)) * W + (ox >> 1)) * CV + cv];
  }
}
__global__ __launch_bounds__(256) void pool2x_kernel(const uint4* dy, uint4* dx, int B, int H, int W, int CV) {
  long nvec = (long)B * H * W * CV;
  EW_LOOP(v, nvec) {
    int cv = (int)(v % CV); long p = v / CV;
    int x_ = (int)(p % W); long q = p / W;
    int y_ = (int)(q % H); int b = (int)(q / H);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dy_ = 0; dy_ < 2; dy_++)
#pragma unroll
      for (int dx_ = 0; dx_ < 2; dx_++) {
        float f[8];
        ew_unpack8(dy[(((long)b * 2 * H + 2 * y_ + dy_) * 2 * W + 2 * x_ + dx_) * CV + cv], f);
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] += f[e];
      }
    dx[v] = ew_pack8(acc);
  }
}
extern "C" int pcm_upsample2x_nhwc(const void* x, void* y, int B, int H, int W, int C, void* stream) {
  PCM_CHECK(x && y && (C % 8) == 0 && PCM_ALIGNED16(x) && PCM_ALIGNED16(y), PCM_EALIGN, "pcm_upsample2x_nhwc: C%%8, alignment");
  PCM_LAUNCH(upsample2x_kernel, dim3(ew_blocks((long)B * 4 * H * W * (C / 8))), dim3(256), 0, stream, (const uint4*)x, (uint4*)y, B, H, W, C / 8);
  return pcm_post_launch("pcm_upsample2x_nhwc");
}
extern "C" int pcm_pool2x_sum_nhwc(const void* dy, void* dx, int B, int H, int W, int C, void* stream) {
  PCM_CHECK(dy && dx && (C % 8) == 0 && PCM_ALIGNED16(dy) && PCM_ALIGNED16(dx), PCM_EALIGN, "pcm_pool2x_sum_nhwc: C%%8, alignment");
  PCM_LAUNCH(pool2x_kernel, dim3(ew_blocks((long)B * H * W * (C / 8))), dim3(256), 0, stream, (const uint4*)dy, (uint4*)dx, B, H, W, C / 8);
  return pcm_post_launch("pcm_pool2x_sum_nhwc");
}

// ---- concat / split ----
// Row/column walker for the [rows][CV] vector grids below: a thread visits vectors v, v+S, v+2S, ... (S = grid size); the
// (row, column) pair is advanced incrementally -- one 32-bit division per thread instead of a 64-bit div+mod per vector --
// and EW_U vectors are fetched per trip so their loads are in flight together.
#define EW_U 4
struct EwWalk {
  unsigned v, r, c, S, sr, sc, CV, nvec;
  __device__ __forceinline__ EwWalk(unsigned rows, unsigned CV_) {
    CV = CV_; nvec = rows * CV_; S = gridDim.x * blockDim.x; sr = S / CV_; sc = S - sr * CV_;
    v = blockIdx.x * blockDim.x + threadIdx.x; r = v / CV_; c = v - r * CV_;
  }
  __device__ __forceinline__ void step() { v += S; c += sc; r += sr; if (c >= CV) { c -= CV; r++; } }
};
static inline int ew_blocks_u(long nvec) { return ew_blocks((nvec + EW_U - 1) / EW_U); }   // EW_U vectors per thread and trip
static inline bool ew_fits32(long rows, long CV) { return rows > 0 && CV > 0 && rows * CV < (1L << 31) - (1L << 22); }

__global__ __launch_bounds__(256) void concat_kernel(const uint4* a, int CVa, const uint4* b, int CVb, uint4* out, int rows) {
  EwWalk w(rows, CVa + CVb);
  while (w.v < w.nvec) {
    uint4 val[EW_U]; unsigned vv[EW_U];
#pragma unroll
    for (int u = 0; u < EW_U; u++) {
      vv[u] = w.v;
      if (w.v < w.nvec) val[u] = (int)w.c < CVa ? a[(size_t)w.r * CVa + w.c] : b[(size_t)w.r * CVb + (w.c - CVa)];
      w.step();
    }
#pragma unroll
    for (int u = 0; u < EW_U; u++)
      if (vv[u] < w.nvec) out[vv[u]] = val[u];
  }
}
__global__ __launch_bounds__(256) void split_kernel(const uint4* in, uint4* a, int CVa, uint4* b, int CVb, int rows, int acc_a) {
  EwWalk w(rows, CVa + CVb);
  while (w.v < w.nvec) {
    uint4 val[EW_U], old[EW_U]; unsigned rr[EW_U], cc[EW_U]; bool ok[EW_U];
#pragma unroll
    for (int u = 0; u < EW_U; u++) {
      rr[u] = w.r; cc[u] = w.c; ok[u] = w.v < w.nvec;
      if (ok[u]) {
        val[u] = in[w.v];
        if (acc_a && (int)w.c < CVa) old[u] = a[(size_t)w.r * CVa + w.c];
      }
      w.step();
    }
#pragma unroll
    for (int u = 0; u < EW_U; u++) {
      if (!ok[u]) continue;
      if ((int)cc[u] < CVa) {
        if (acc_a) {
          float f[8], g[8];
          ew_unpack8(val[u], f); ew_unpack8(old[u], g);
#pragma unroll
          for (int e = 0; e < 8; e++) f[e] += g[e];
          val[u] = ew_pack8(f);
        }
        a[(size_t)rr[u] * CVa + cc[u]] = val[u];
      } else {
        b[(size_t)rr[u] * CVb + (cc[u] - CVa)] = val[u];
      }
    }
  }
}
extern "C" int pcm_concat_channels(const void* a, int Ca, const void* b, int Cb, void* out, long rows, void* stream) {
  PCM_CHECK(a && b && out && (Ca % 8) == 0 && (Cb % 8) == 0 && ew_fits32(rows, (Ca + Cb) / 8), PCM_EINVAL, "pcm_concat_channels: C%%8, rows*C/8 < 2^31");
  PCM_LAUNCH(concat_kernel, dim3(ew_blocks_u(rows * ((Ca + Cb) / 8))), dim3(256), 0, stream, (const uint4*)a, Ca / 8, (const uint4*)b, Cb / 8, (uint4*)out, (int)rows);
  return pcm_post_launch("pcm_concat_channels");
}
extern "C" int pcm_split_channels(const void* in, void* a, int Ca, void* b, int Cb, long rows, int accumulate_a, void* stream) {
  PCM_CHECK(in && a && b && (Ca % 8) == 0 && (Cb % 8) == 0 && ew_fits32(rows, (Ca + Cb) / 8), PCM_EINVAL, "pcm_split_channels: C%%8, rows*C/8 < 2^31");
  PCM_LAUNCH(split_kernel, dim3(ew_blocks_u(rows * ((Ca + Cb) / 8))), dim3(256), 0, stream, (const uint4*)in, (uint4*)a, Ca / 8, (uint4*)b, Cb / 8, (int)rows, accumulate_a);
  return pcm_post_launch("pcm_split_channels");
}

// ---- add / silu / geglu ----
__global__ __launch_bounds__(256) void add_kernel(const uint4* a, const uint4* b, uint4* o, long nvec) {
  EW_LOOP(v, nvec) {
    float f[8], g[8];
    ew_unpack8(a[v], f); ew_unpack8(b[v], g);
#pragma unroll
    for (int e = 0; e < 8; e++) f[e] += g[e];
    o[v] = ew_pack8(f);
  }
}
__global__ __launch_bounds__(256) void silu_kernel(const uint4* a, uint4* o, long nvec) {
  EW_LOOP(v, nvec) {
    float f[8];
    ew_unpack8(a[v], f);
#pragma unroll
    for (int e = 0; e < 8; e++) f[e] = silu_f(f[e]);
    o[v] = ew_pack8(f);
  }
}
// dx = dy * silu'(x)
__global__ __launch_bounds__(256) void silu_bwd_kernel(const uint4* x, const uint4* dy, uint4* dx, long nvec) {
  EW_LOOP(v, nvec) {
    float f[8], d[8];
    ew_unpack8(x[v], f); ew_unpack8(dy[v], d);
#pragma unroll
    for (int e = 0; e < 8; e++) d[e] *= silu_grad_f(f[e]);
    dx[v] = ew_pack8(d);
  }
}
extern "C" int pcm_silu_bwd_bf16(const void* x, const void* dy, void* dx, long n, void* stream) {
  PCM_CHECK(x && dy && dx && (n % 8) == 0, PCM_EINVAL, "pcm_silu_bwd_bf16: n%%8");
  PCM_LAUNCH(silu_bwd_kernel, dim3(ew_blocks(n / 8)), dim3(256), 0, stream, (const uint4*)x, (const uint4*)dy, (uint4*)dx, n / 8);
  return pcm_post_launch("pcm_silu_bwd_bf16");
}
extern "C" int pcm_add_bf16(const void* a, const void* b, void* out, long n, void* stream) {
  PCM_CHECK(a && b && out && (n % 8) == 0, PCM_EINVAL, "pcm_add_bf16: n%%8");
  PCM_LAUNCH(add_kernel, dim3(ew_blocks(n / 8)), dim3(256), 0, stream, (const uint4*)a, (const uint4*)b, (uint4*)out, n / 8);
  return pcm_post_launch("pcm_add_bf16");
}
extern "C" int pcm_silu_bf16(const void* x, void* y, long n, void* stream) {
  PCM_CHECK(x && y && (n % 8) == 0, PCM_EINVAL, "pcm_silu_bf16: n%%8");
  PCM_LAUNCH(silu_kernel, dim3(ew_blocks(n / 8)), dim3(256), 0, stream, (const uint4*)x, (uint4*)y, n / 8);
  return pcm_post_launch("pcm_silu_bf16");
}
// hg [M][2*C4]: h = cols [0,C4), g = cols [C4, 2*C4)
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const uint4* hg, uint4* out, int M, int CV4) {
  EwWalk w(M, CV4);
  while (w.v < w.nvec) {
    uint4 hr[EW_U], gr[EW_U]; unsigned vv[EW_U];
#pragma unroll
    for (int u = 0; u < EW_U; u++) {
      vv[u] = w.v;
      if (w.v < w.nvec) { const uint4* p = hg + (size_t)w.r * 2 * CV4 + w.c; hr[u] = p[0]; gr[u] = p[CV4]; }
      w.step();
    }
#pragma unroll
    for (int u = 0; u < EW_U; u++) {
      if (vv[u] >= w.nvec) continue;
      float h[8], g[8];
      ew_unpack8(hr[u], h); ew_unpack8(gr[u], g);
#pragma unroll
      for (int e = 0; e < 8; e++) h[e] *= gelu_erf_f(g[e]);
      out[vv[u]] = ew_pack8(h);
    }
  }
}
// IL: the saved pre-activation is in the INTERLEAVED column order of the fused projection (2 values, their 2 gates, ...: pcm_hip.h
// PCM_ACT_GEGLU with pre_out); the gradient is written in the standard [values | gates] order the dgrad / wgrad GEMMs expect
template <bool IL>
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const uint4* hg, const uint4* dout, uint4* dhg, int M, int CV4, int ldp4) {
  EwWalk w(M, CV4);
  while (w.v < w.nvec) {
    uint4 hr[EW_U], gr[EW_U], dr[EW_U]; unsigned rr[EW_U], cc[EW_U]; bool ok[EW_U];
#pragma unroll
    for (int u = 0; u < EW_U; u++) {
      rr[u] = w.r; cc[u] = w.c; ok[u] = w.v < w.nvec;
      if (ok[u]) {
        const uint4* p = IL ? hg + (size_t)w.r * ldp4 + 2 * w.c : hg + (size_t)w.r * 2 * CV4 + w.c;
        hr[u] = p[0]; gr[u] = p[IL ? 1 : CV4]; dr[u] = dout[w.v];
      }
      w.step();
    }
#pragma unroll
    for (int u = 0; u < EW_U; u++) {
      if (!ok[u]) continue;
      float h[8], g[8], d[8], dh[8], dg[8];
      ew_unpack8(hr[u], h); ew_unpack8(gr[u], g); ew_unpack8(dr[u], d);
      if (IL) {   // 16 interleaved columns [v0 v1 g0 g1 | v2 v3 g2 g3 | v4 v5 g4 g5 | v6 v7 g6 g7] -> values h[0..7], gates g[0..7]
        const float a[8] = {h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]}, b[8] = {g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7]};
        h[0] = a[0]; h[1] = a[1]; h[2] = a[4]; h[3] = a[5]; h[4] = b[0]; h[5] = b[1]; h[6] = b[4]; h[7] = b[5];
        g[0] = a[2]; g[1] = a[3]; g[2] = a[6]; g[3] = a[7]; g[4] = b[2]; g[5] = b[3]; g[6] = b[6]; g[7] = b[7];
      }
#pragma unroll
      for (int e = 0; e < 8; e++) { dh[e] = d[e] * gelu_erf_f(g[e]); dg[e] = d[e] * h[e] * gelu_erf_grad_f(g[e]); }
      uint4* q = dhg + (size_t)rr[u] * 2 * CV4 + cc[u];
      q[0] = ew_pack8(dh);
      q[CV4] = ew_pack8(dg);
    }
  }
}
extern "C" int pcm_geglu_fwd(const void* hg, void* out, int M, int C4, void* stream) {
  PCM_CHECK(hg && out && (C4 % 8) == 0 && ew_fits32(M, C4 / 8), PCM_EINVAL, "pcm_geglu_fwd: C4%%8, M*C4/8 < 2^31");
  PCM_LAUNCH(geglu_fwd_kernel, dim3(ew_blocks_u((long)M * (C4 / 8))), dim3(256), 0, stream, (const uint4*)hg, (uint4*)out, M, C4 / 8);
  return pcm_post_launch("pcm_geglu_fwd");
}
extern "C" int pcm_geglu_bwd(const void* hg, const void* dout, void* dhg, int M, int C4, void* stream) {
  PCM_CHECK(hg && dout && dhg && (C4 % 8) == 0 && ew_fits32(M, C4 / 8), PCM_EINVAL, "pcm_geglu_bwd: C4%%8, M*C4/8 < 2^31");
  PCM_LAUNCH(geglu_bwd_kernel<false>, dim3(ew_blocks_u((long)M * (C4 / 8))), dim3(256), 0, stream, (const uint4*)hg, (const uint4*)dout, (uint4*)dhg, M, C4 / 8, 0);
  return pcm_post_launch("pcm_geglu_bwd");
}
extern "C" int pcm_geglu_bwd_interleaved(const void* pre, int ldp, const void* dout, void* dhg, int M, int C4, void* stream) {
  PCM_CHECK(pre && dout && dhg && (C4 % 8) == 0 && (ldp % 8) == 0 && ldp >= 2 * C4 && ew_fits32(M, C4 / 8) && PCM_ALIGNED16(pre), PCM_EINVAL,
            "pcm_geglu_bwd_interleaved: C4%%8, ldp%%8, ldp >= 2*C4, M*C4/8 < 2^31");
  PCM_LAUNCH(geglu_bwd_kernel<true>, dim3(ew_blocks_u((long)M * (C4 / 8))), dim3(256), 0, stream, (const uint4*)pre, (const uint4*)dout, (uint4*)dhg, M, C4 / 8, ldp / 8);
  return pcm_post_launch("pcm_geglu_bwd_interleaved");
}

// ---- pixel sum: out[b][c] = sum_hw x[b][hw][c]  (fp32, zeroed here) ----
// part != nullptr (reproducible form): block (chunk, b, zc) stores its sums to part[chunk][b][C]; colsum_finalize_kernel adds the chunks in order
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* x, float* out, int HW, int C, int CVL, int ppb, float* part) {
  const int b = blockIdx.y, zc = blockIdx.z;
  const int cvl = threadIdx.x % CVL, pl = threadIdx.x / CVL, k = blockDim.x / CVL;
  const int c0 = (zc * CVL + cvl) * 8;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int p0 = blockIdx.x * ppb, p1 = p0 + ppb; if (p1 > HW) p1 = HW;
  // 4 rows in flight per thread (a plain one-row loop is one memory round trip per row: 30 us for 64 rows per block)
  const bf16_t* col = x + (size_t)b * HW * C + c0;
  int p = p0 + pl;
  for (; p + 3 * k < p1; p += 4 * k) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = *(const uint4*)(col + (size_t)(p + u * k) * C);
#pragma unroll
    for (int u = 0; u < 4; u++) {
      float f[8];
      ew_unpack8(v[u], f);
#pragma unroll
      for (int e = 0; e < 8; e++) s[e] += f[e];
    }
  }
  for (; p < p1; p += k) {
    float f[8];
    ew_unpack8(*(const uint4*)(col + (size_t)p * C), f);
#pragma unroll
    for (int e = 0; e < 8; e++) s[e] += f[e];
  }
  // reduce the k pixel-lanes of the block in LDS first: one atomic per (block, channel) instead of one per thread
  __shared__ float red[256 * 8];
#pragma unroll
  for (int e = 0; e < 8; e++) red[(pl * CVL + cvl) * 8 + e] = s[e];
  __syncthreads();
  for (int i = threadIdx.x; i < CVL * 8; i += blockDim.x) {
    float t = 0.f;
    for (int j = 0; j < k; j++) t += red[j * CVL * 8 + i];
    if (part) part[((size_t)blockIdx.x * gridDim.y + b) * C + zc * CVL * 8 + i] = t;
    else atomicAdd(&out[(size_t)b * C + zc * CVL * 8 + i], t);
  }
}
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* part, float* out, int chunks, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float t = 0.f;
    for (int c = 0; c < chunks; c++) t += part[(size_t)c * n + i];
    out[i] = t;
  }
}
static void colsum_geometry(int B, int HW, int C, int* split_, int* CVL_, int* k_, int* chunks_, int* ppb_) {
  int CV = C / 8, split = 1;
  while (CV / split > 256 || (CV % split) != 0) split++;
  int CVL = CV / split, k = 256 / CVL;
  int chunks = (PCM_GRID_CAP(1024) + B * split - 1) / (B * split);
  int maxc = (HW + k - 1) / k; if (chunks > maxc) chunks = maxc; if (chunks < 1) chunks = 1;
  int ppb = (HW + chunks - 1) / chunks; chunks = (HW + ppb - 1) / ppb;
  *split_ = split; *CVL_ = CVL; *k_ = k; *chunks_ = chunks; *ppb_ = ppb;
}
extern "C" int pcm_colsum_bf16(const void* x, void* out, int B, int HW, int C, void* stream) {
  PCM_CHECK(x && out && B > 0 && HW > 0 && C > 0 && (C % 8) == 0 && PCM_ALIGNED16(x), PCM_EALIGN, "pcm_colsum_bf16: C%%8, alignment");
  int split, CVL, k, chunks, ppb;
  colsum_geometry(B, HW, C, &split, &CVL, &k, &chunks, &ppb);
  pcm_zero_async(out, sizeof(float) * (size_t)B * C, stream);
  PCM_LAUNCH(colsum_kernel, dim3(chunks, B, split), dim3(CVL * k), 0, stream, (const bf16_t*)x, (float*)out, HW, C, CVL, ppb, (float*)nullptr);
  return pcm_post_launch("pcm_colsum_bf16");
}
extern "C" size_t pcm_colsum_workspace_bytes(int B, int HW, int C) {
  if (B <= 0 || HW <= 0 || C <= 0 || (C % 8)) return 0;
  int split, CVL, k, chunks, ppb;
  colsum_geometry(B, HW, C, &split, &CVL, &k, &chunks, &ppb);
  return sizeof(float) * (size_t)chunks * B * C;
}
extern "C" int pcm_colsum_bf16_ws(const void* x, void* out, int B, int HW, int C, void* workspace, size_t workspace_bytes, void* stream) {
  PCM_CHECK(x && out && workspace && B > 0 && HW > 0 && C > 0 && (C % 8) == 0 && PCM_ALIGNED16(x), PCM_EALIGN, "pcm_colsum_bf16_ws: C%%8, alignment");
  int split, CVL, k, chunks, ppb;
  colsum_geometry(B, HW, C, &split, &CVL, &k, &chunks, &ppb);
  PCM_CHECK(workspace_bytes >= sizeof(float) * (size_t)chunks * B * C, PCM_EINVAL, "pcm_colsum_bf16_ws: workspace too small");
  PCM_LAUNCH(colsum_kernel, dim3(chunks, B, split), dim3(CVL * k), 0, stream, (const bf16_t*)x, (float*)out, HW, C, CVL, ppb, (float*)workspace);
  const long n = (long)B * C;
  long blocks = (n + 255) / 256; if (blocks > PCM_GRID_CAP(1024)) blocks = PCM_GRID_CAP(1024);
  PCM_LAUNCH(colsum_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)workspace, (float*)out, chunks, n);
  return pcm_post_launch("pcm_colsum_bf16_ws");
}

// Stage n fp32 weights into LDS through a per-element index map.  The global loads are issued in batches of 8 per thread
// before any LDS store: written as a plain load/store loop hipcc waits for every load before its store, i.e. one memory
// round trip per element per thread (45 serial round trips ~ 90 us per block for the 36x320 edge-conv weights).
template <typename F>
__device__ __forceinline__ void stage_weights(const float* w, float* wl, int n, F map) {
  for (int base = 0; base < n; base += 8 * (int)blockDim.x) {
    float r[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { int i = base + u * (int)blockDim.x + (int)threadIdx.x; r[u] = i < n ? w[i] : 0.f; }
#pragma unroll
    for (int u = 0; u < 8; u++) { int i = base + u * (int)blockDim.x + (int)threadIdx.x; if (i < n) wl[map(i)] = r[u]; }
  }
}

// ---- 4-channel edge convolutions -------------------------------------------------------
// conv4: in NCHW fp32 [B][4][H][W] -> out NHWC bf16 [B][H][W][C0], 3x3 pad 1.
// flip=0: w [C0][4][3][3] (conv_in).  flip=1: w [4][C0][3][3] read transposed with flipped taps
// (input gradient of conv_out).  Weights staged in LDS as wl[j=(ci,tap)][c].
__global__ __launch_bounds__(256) void conv4_kernel(const float* x, const float* w, const float* bias, bf16_t* y,
                                                    int B, int H, int W, int C0, int flip, bf16_t* y2, int ld2) {
  PCM_DYN_SMEM(smem);
  float* wl = (float*)smem;  // [36][C0]
  stage_weights(w, wl, 36 * C0, [&](int i) {
    int tap = i % 9, r = i / 9;
    if (flip) { int ci = r / C0, c = r - ci * C0; return (ci * 9 + (8 - tap)) * C0 + c; }
    int c = r >> 2, ci = r & 3;
    return (ci * 9 + tap) * C0 + c;
  });
  __syncthreads();
  const int CV = C0 / 8;
  long nvec = (long)B * H * W * CV;
  EW_LOOP(v, nvec) {
    int cv = (int)(v % CV); long p = v / CV;
    int px = (int)(p % W); long q = p / W;
    int py = (int)(q % H); int b = (int)(q / H);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] = bias ? bias[cv * 8 + e] : 0.f;
    for (int ci = 0; ci < 4; ci++)
#pragma unroll
      for (int tap = 0; tap < 9; tap++) {
        int iy = py + tap / 3 - 1, ix = px + tap % 3 - 1;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        float xv = x[(((size_t)b * 4 + ci) * H + iy) * W + ix];
        const float* wr = wl + (ci * 9 + tap) * C0 + cv * 8;
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] += xv * wr[e];
      }
    const uint4 o = ew_pack8(acc);
    *(uint4*)(y + v * 8) = o;
    if (y2) *(uint4*)(y2 + (size_t)p * ld2 + cv * 8) = o;      // second copy, row stride ld2 (abi 5: the skip's slot in its concat buffer)
  }
}
// W % 4 == 0 variant: one item = 4 consecutive pixels of a row x 8 channels, so every staged weight vector feeds 4 pixels
// and the 3x6 input window is loaded once; weights are staged with coalesced global reads (the scatter is on the LDS side).
__global__ __launch_bounds__(256) void conv4x4_kernel(const float* x, const float* w, const float* bias, bf16_t* y,
                                                      int B, int H, int W, int C0, int flip, bf16_t* y2, int ld2) {
  PCM_DYN_SMEM(smem);
  float* wl = (float*)smem;  // [36][C0]
  stage_weights(w, wl, 36 * C0, [&](int i) {
    int tap = i % 9, r = i / 9;
    if (flip) { int ci = r / C0, c = r - ci * C0; return (ci * 9 + (8 - tap)) * C0 + c; }
    int c = r >> 2, ci = r & 3;
    return (ci * 9 + tap) * C0 + c;
  });
  __syncthreads();
  const int CV = C0 / 8, WQ = W / 4;
  const long nitem = (long)B * H * WQ * CV;
  EW_LOOP(v, nitem) {
    int cv = (int)(v % CV); long p = v / CV;
    int px0 = (int)(p % WQ) * 4; long q = p / WQ;
    int py = (int)(q % H); int b = (int)(q / H);
    float acc[4][8];
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int e = 0; e < 8; e++) acc[k][e] = bias ? bias[cv * 8 + e] : 0.f;
#pragma unroll 1
    for (int ci = 0; ci < 4; ci++) {
      const float* xc = x + ((size_t)b * 4 + ci) * H * W;
#pragma unroll 1
      for (int dy = 0; dy < 3; dy++) {
        const int iy = py + dy - 1;
        if (iy < 0 || iy >= H) continue;
        float xr[6];
#pragma unroll
        for (int k = 0; k < 6; k++) { int ix = px0 + k - 1; xr[k] = (ix >= 0 && ix < W) ? xc[(size_t)iy * W + ix] : 0.f; }
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
          const float4 w0 = *(const float4*)(wl + (ci * 9 + dy * 3 + dx) * C0 + cv * 8), w1 = *(const float4*)(wl + (ci * 9 + dy * 3 + dx) * C0 + cv * 8 + 4);
          const float wr[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
          for (int k = 0; k < 4; k++)
#pragma unroll
            for (int e = 0; e < 8; e++) acc[k][e] += xr[k + dx] * wr[e];
        }
      }
    }
    const size_t row0 = ((size_t)b * H + py) * W + px0;
    bf16_t* yo = y + row0 * C0 + cv * 8;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint4 o = ew_pack8(acc[k]);
      *(uint4*)(yo + (size_t)k * C0) = o;
      if (y2) *(uint4*)(y2 + (row0 + k) * ld2 + cv * 8) = o;
    }
  }
}
static void conv4_launch(const float* x, const float* w, const float* bias, bf16_t* y, int B, int H, int W, int C0, int flip, void* stream,
                         bf16_t* y2 = nullptr, int ld2 = 0) {
  if ((W % 4) == 0) {
    long blocks = ((long)B * H * (W / 4) * (C0 / 8) + 255) / 256; if (blocks > PCM_GRID_CAP(768)) blocks = PCM_GRID_CAP(768);
    PCM_LAUNCH(conv4x4_kernel, dim3((int)blocks), dim3(256), 36 * C0 * 4, stream, x, w, bias, y, B, H, W, C0, flip, y2, ld2);
  } else {
    PCM_LAUNCH(conv4_kernel, dim3(ew_blocks((long)B * H * W * (C0 / 8))), dim3(256), 36 * C0 * 4, stream, x, w, bias, y, B, H, W, C0, flip, y2, ld2);
  }
}
extern "C" int pcm_conv_in_fwd(const float* x, const float* w, const float* bias, void* y, int B, int H, int W, int C0, void* stream) {
  PCM_CHECK(x && w && y && (C0 % 8) == 0 && C0 <= 1024, PCM_EINVAL, "pcm_conv_in_fwd: C0%%8, C0<=1024");
  conv4_launch(x, w, bias, (bf16_t*)y, B, H, W, C0, 0, stream);
  return pcm_post_launch("pcm_conv_in_fwd");
}
// abi 5: the same with a second copy of the output rows, y2[pixel][ld2] (conv_in's output is the first skip tensor: its slot in the last
// up-block resnet's concatenated input, discriminator_sd15.py:312-342)
extern "C" int pcm_conv_in_fwd2(const float* x, const float* w, const float* bias, void* y, void* y2, int ld2, int B, int H, int W, int C0, void* stream) {
  PCM_CHECK(x && w && y && (C0 % 8) == 0 && C0 <= 1024, PCM_EINVAL, "pcm_conv_in_fwd2: C0%%8, C0<=1024");
  PCM_CHECK(!y2 || (PCM_ALIGNED16(y2) && (ld2 % 8) == 0 && ld2 >= C0), PCM_EALIGN, "pcm_conv_in_fwd2: y2 alignment / ld2");
  conv4_launch(x, w, bias, (bf16_t*)y, B, H, W, C0, 0, stream, (bf16_t*)y2, ld2);
  return pcm_post_launch("pcm_conv_in_fwd2");
}
extern "C" int pcm_conv_out_bwd(const float* dy, const float* w, void* dx, int B, int H, int W, int C0, void* stream) {
  PCM_CHECK(dy && w && dx && (C0 % 8) == 0 && C0 <= 1024, PCM_EINVAL, "pcm_conv_out_bwd: C0%%8, C0<=1024");
  conv4_launch(dy, w, (const float*)nullptr, (bf16_t*)dx, B, H, W, C0, 1, stream);
  return pcm_post_launch("pcm_conv_out_bwd");
}
// conv_out: x NHWC bf16 [B][H][W][C0] -> y NCHW fp32 [B][4][H][W]; one wave per output pixel
__global__ __launch_bounds__(256) void conv_out_kernel(const bf16_t* x, const float* w, const float* bias, float* y,
                                                       int B, int H, int W, int C0) {
  PCM_DYN_SMEM(smem);
  float* wl = (float*)smem;  // [9][4][C0]
  stage_weights(w, wl, 36 * C0, [&](int i) { int tap = i % 9, r = i / 9; return tap * 4 * C0 + r; });   // r = o*C0 + c
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, CV = C0 / 8;
  long npix = (long)B * H * W;
  for (long p = (long)blockIdx.x * 4 + wv; p < npix; p += (long)gridDim.x * 4) {
    int px = (int)(p % W); long q = p / W;
    int py = (int)(q % H); int b = (int)(q / H);
    float acc[4] = {0, 0, 0, 0};
    for (int tap = 0; tap < 9; tap++) {
      int iy = py + tap / 3 - 1, ix = px + tap % 3 - 1;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      const bf16_t* xr = x + (((size_t)b * H + iy) * W + ix) * C0;
      for (int cv = lane; cv < CV; cv += 64) {
        float f[8];
        ew_unpack8(*(const uint4*)(xr + cv * 8), f);
#pragma unroll
        for (int o = 0; o < 4; o++) {
          const float* wr = wl + (tap * 4 + o) * C0 + cv * 8;
#pragma unroll
          for (int e = 0; e < 8; e++) acc[o] += f[e] * wr[e];
        }
      }
    }
#pragma unroll
    for (int o = 0; o < 4; o++) acc[o] = wave_sum(acc[o]);
    if (lane < 4) {
      float v = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3];
      y[(((size_t)b * 4 + lane) * H + py) * W + px] = v + (bias ? bias[lane] : 0.f);
    }
  }
}
// W % 8 == 0, C0 <= 512 variant: one wave per 8 consecutive pixels of a row; lane = 8-channel group.  The 4x8 weights of a tap
// are read from LDS once per 8 pixels; the 32 (pixel, out-channel) partial sums are reduced across the wave by a halving
// butterfly (32 shuffles instead of 32 full wave reductions): lane l ends up with value index l>>1.
__global__ __launch_bounds__(256) void conv_out8_kernel(const bf16_t* x, const float* w, const float* bias, float* y,
                                                        int B, int H, int W, int C0) {
  PCM_DYN_SMEM(smem);
  float* wl = (float*)smem;  // [9][4][C0]
  stage_weights(w, wl, 36 * C0, [&](int i) { int tap = i % 9, r = i / 9; return tap * 4 * C0 + r; });   // r = o*C0 + c
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, CV = C0 / 8, W8 = W / 8;
  const long nitem = (long)B * H * W8;
  for (long it = (long)blockIdx.x * 4 + wv; it < nitem; it += (long)gridDim.x * 4) {
    int px0 = (int)(it % W8) * 8; long q = it / W8;
    int py = (int)(q % H); int b = (int)(q / H);
    float vals[32];
#pragma unroll
    for (int i = 0; i < 32; i++) vals[i] = 0.f;
    for (int cv = lane; cv < CV; cv += 64) {
#pragma unroll 1   // one tap at a time: fully unrolled this kernel needs 400+ VGPRs (one wave per SIMD, latency bound)
      for (int tap = 0; tap < 9; tap++) {
        const int iy = py + tap / 3 - 1;
        if (iy < 0 || iy >= H) continue;
        float wr[4][8];
#pragma unroll
        for (int o = 0; o < 4; o++) {
          const float4 w0 = *(const float4*)(wl + (tap * 4 + o) * C0 + cv * 8), w1 = *(const float4*)(wl + (tap * 4 + o) * C0 + cv * 8 + 4);
          wr[o][0] = w0.x; wr[o][1] = w0.y; wr[o][2] = w0.z; wr[o][3] = w0.w; wr[o][4] = w1.x; wr[o][5] = w1.y; wr[o][6] = w1.z; wr[o][7] = w1.w;
        }
        const bf16_t* xr = x + (((size_t)b * H + iy) * W) * C0 + cv * 8;
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int ix = px0 + k + tap % 3 - 1;
          if (ix < 0 || ix >= W) continue;
          float f[8];
          ew_unpack8(*(const uint4*)(xr + (size_t)ix * C0), f);
#pragma unroll
          for (int o = 0; o < 4; o++)
#pragma unroll
            for (int e = 0; e < 8; e++) vals[k * 4 + o] += f[e] * wr[o][e];
        }
      }
    }
#pragma unroll
    for (int s = 32, n = 16; n >= 1; s >>= 1, n >>= 1) {
      const bool up = (lane & s) != 0;
#pragma unroll
      for (int i = 0; i < n; i++) {
        const float lo = vals[i], hi = vals[i + n];
        const float recv = __shfl_xor(up ? lo : hi, s);
        vals[i] = (up ? hi : lo) + recv;
      }
    }
    const float tot = vals[0] + __shfl_xor(vals[0], 1);
    if (!(lane & 1)) {
      const int idx = lane >> 1, k = idx >> 2, o = idx & 3;
      y[(((size_t)b * 4 + o) * H + py) * W + px0 + k] = tot + (bias ? bias[o] : 0.f);
    }
  }
}
extern "C" int pcm_conv_out_fwd(const void* x, const float* w, const float* bias, float* y, int B, int H, int W, int C0, void* stream) {
  PCM_CHECK(x && w && y && (C0 % 8) == 0 && C0 <= 1024 && PCM_ALIGNED16(x), PCM_EINVAL, "pcm_conv_out_fwd: C0%%8, C0<=1024");
  long npix = (long)B * H * W;
  if ((W % 8) == 0) {
    long blocks = (npix / 8 + 3) / 4; if (blocks > PCM_GRID_CAP(768)) blocks = PCM_GRID_CAP(768);
    PCM_LAUNCH(conv_out8_kernel, dim3((int)blocks), dim3(256), 36 * C0 * 4, stream, (const bf16_t*)x, w, bias, y, B, H, W, C0);
  } else {
    long blocks = (npix + 3) / 4; if (blocks > PCM_GRID_CAP(768)) blocks = PCM_GRID_CAP(768);
    PCM_LAUNCH(conv_out_kernel, dim3((int)blocks), dim3(256), 36 * C0 * 4, stream, (const bf16_t*)x, w, bias, y, B, H, W, C0);
  }
  return pcm_post_launch("pcm_conv_out_fwd");
}

// ---- timestep projection: [cos(t f_i) | sin(t f_i)], f_i = exp(-ln(1e4) i / half) ----
__global__ __launch_bounds__(256) void temb_kernel(const int64_t* t, bf16_t* out, int B, int dim) {
  int half = dim / 2;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  int b = i / half, j = i - b * half;
  float f = expf(-9.210340371976184f * (float)j / (float)half);
  float arg = (float)t[b] * f;
  out[(size_t)b * dim + j] = f2bf(cosf(arg));
  out[(size_t)b * dim + half + j] = f2bf(sinf(arg));
}
extern "C" int pcm_timestep_embedding(const int64_t* t, void* out, int B, int dim, void* stream) {
  PCM_CHECK(t && out && (dim % 2) == 0, PCM_EINVAL, "pcm_timestep_embedding: dim even");
  PCM_LAUNCH(temb_kernel, dim3((B * dim / 2 + 255) / 256), dim3(256), 0, stream, t, (bf16_t*)out, B, dim);
  return pcm_post_launch("pcm_timestep_embedding");
}

// ---- conv1x1 to ONE channel (DiscriminatorHead.conv_out, discriminator_sd15.py:362): out[m] = x[m,:].w + b ----
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const bf16_t* x, const float* w, const float* bias, float* out, long M, int C) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, CV = C / 8;
  for (long m = (long)blockIdx.x * 4 + wv; m < M; m += (long)gridDim.x * 4) {
    float acc = 0.f;
    for (int cv = lane; cv < CV; cv += 64) {
      float f[8];
      ew_unpack8(*(const uint4*)(x + m * C + cv * 8), f);
#pragma unroll
      for (int e = 0; e < 8; e++) acc += f[e] * w[cv * 8 + e];
    }
    acc = wave_sum(acc);
    if (lane == 0) out[m] = acc + (bias ? bias[0] : 0.f);
  }
}
// dx[m][c] = dy[m] * w[c] ; dw[c] += sum_m dy[m] x[m][c] ; db += sum_m dy[m]
// part != nullptr (reproducible form): block b stores its C weight sums and its bias sum to part[b][C + 1]; an ordered finalize adds them
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const bf16_t* x, const float* w, const float* dy, bf16_t* dx, float* dw, float* db,
                                                         long M, int C, int rows_per_block, float* part) {
  const int CV = C / 8;
  const int cvl = threadIdx.x % CV, pl = threadIdx.x / CV, k = blockDim.x / CV;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wv[8];
#pragma unroll
  for (int e = 0; e < 8; e++) wv[e] = w[cvl * 8 + e];
  long m0 = (long)blockIdx.x * rows_per_block, m1 = m0 + rows_per_block; if (m1 > M) m1 = M;
  float sb = 0.f;
  auto row = [&](long m, float d, const uint4& xr) {
    float f[8], o[8];
    ew_unpack8(xr, f);
#pragma unroll
    for (int e = 0; e < 8; e++) { s[e] += d * f[e]; o[e] = d * wv[e]; }
    if (dx) *(uint4*)(dx + m * C + cvl * 8) = ew_pack8(o);
    if (cvl == 0) sb += d;
  };
  long m = m0 + pl;
  for (; m + 3 * k < m1; m += 4 * k) {          // four rows in flight per thread (same summation order as one row at a time)
    float d[4]; uint4 xr[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { d[u] = dy[m + u * k]; xr[u] = *(const uint4*)(x + (m + u * k) * C + cvl * 8); }
#pragma unroll
    for (int u = 0; u < 4; u++) row(m + u * k, d[u], xr[u]);
  }
  for (; m < m1; m += k) row(m, dy[m], *(const uint4*)(x + m * C + cvl * 8));
  // block-level reduction of the k row-lanes in LDS, then one atomic per (block, channel) -- see gn_param_grad_kernel
  __shared__ float red[256 * 8 + 256];   // bias partials: one per row-lane, k = blockDim/CV <= 256 (C = 8)
#pragma unroll
  for (int e = 0; e < 8; e++) red[(pl * CV + cvl) * 8 + e] = s[e];
  if (cvl == 0) red[256 * 8 + pl] = sb;
  __syncthreads();
  for (int i = threadIdx.x; i < CV * 8; i += blockDim.x) {
    float t = 0.f;
    for (int j = 0; j < k; j++) t += red[j * CV * 8 + i];
    if (part) part[(size_t)blockIdx.x * (C + 1) + i] = t; else atomicAdd(&dw[i], t);
  }
  if (threadIdx.x == 0 && (db || part)) {
    float t = 0.f;
    for (int j = 0; j < k; j++) t += red[256 * 8 + j];
    if (part) part[(size_t)blockIdx.x * (C + 1) + C] = t; else atomicAdd(db, t);
  }
}
static void rowdot_bwd_geometry(long M, int C, int* k_, long* blocks_, long* rpb_) {
  int CV = C / 8, k = 256 / CV; if (k < 1) k = 1;
  long blocks = PCM_GRID_CAP(512); long rpb = (M + blocks - 1) / blocks; if (rpb < k) rpb = k;
  blocks = (M + rpb - 1) / rpb;
  *k_ = k; *blocks_ = blocks; *rpb_ = rpb;
}
extern "C" int pcm_rowdot_fwd(const void* x, const float* w, const float* bias, float* out, long M, int C, void* stream) {
  PCM_CHECK(x && w && out && M > 0 && (C % 8) == 0 && PCM_ALIGNED16(x), PCM_EINVAL, "pcm_rowdot_fwd: C%%8, alignment");
  long blocks = (M + 3) / 4; if (blocks > PCM_GRID_CAP(2048)) blocks = PCM_GRID_CAP(2048);
  PCM_LAUNCH(rowdot_fwd_kernel, dim3((int)blocks), dim3(256), 0, stream, (const bf16_t*)x, w, bias, out, M, C);
  return pcm_post_launch("pcm_rowdot_fwd");
}
extern "C" int pcm_rowdot_bwd(const void* x, const float* w, const float* dy, void* dx, float* dw, float* db, long M, int C, void* stream) {
  PCM_CHECK(x && w && dy && dw && M > 0 && (C % 8) == 0 && C <= 2048 && PCM_ALIGNED16(x), PCM_EINVAL, "pcm_rowdot_bwd: C%%8, C<=2048, alignment");
  int k; long blocks, rpb;
  rowdot_bwd_geometry(M, C, &k, &blocks, &rpb);
  PCM_LAUNCH(rowdot_bwd_kernel, dim3((int)blocks), dim3((C / 8) * k), 0, stream, (const bf16_t*)x, w, dy, (bf16_t*)dx, dw, db, M, C, (int)rpb, (float*)nullptr);
  return pcm_post_launch("pcm_rowdot_bwd");
}
// reproducible form (abi 5): per-block partials in the caller's workspace + an ordered finalize that ADDS into dw / db like the atomic form
extern "C" size_t pcm_rowdot_bwd_workspace_bytes(long M, int C) {
  if (M <= 0 || C <= 0 || (C % 8) || C > 2048) return 0;
  int k; long blocks, rpb;
  rowdot_bwd_geometry(M, C, &k, &blocks, &rpb);
  return sizeof(float) * (size_t)blocks * (C + 1);
}
extern "C" int pcm_rowdot_bwd_ws(const void* x, const float* w, const float* dy, void* dx, float* dw, float* db, long M, int C, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  PCM_CHECK(x && w && dy && dw && workspace && M > 0 && (C % 8) == 0 && C <= 2048 && PCM_ALIGNED16(x), PCM_EINVAL, "pcm_rowdot_bwd_ws: C%%8, C<=2048, alignment");
  int k; long blocks, rpb;
  rowdot_bwd_geometry(M, C, &k, &blocks, &rpb);
  PCM_CHECK(workspace_bytes >= sizeof(float) * (size_t)blocks * (C + 1), PCM_EINVAL, "pcm_rowdot_bwd_ws: workspace too small");
  PCM_LAUNCH(rowdot_bwd_kernel, dim3((int)blocks), dim3((C / 8) * k), 0, stream, (const bf16_t*)x, w, dy, (bf16_t*)dx, dw, db, M, C, (int)rpb, (float*)workspace);
  pcm_partials_finalize((const float*)workspace, C + 1, dw, (int)blocks, C, 1, stream);
  if (db) pcm_partials_finalize((const float*)workspace + C, C + 1, db, (int)blocks, 1, 1, stream);
  return pcm_post_launch("pcm_rowdot_bwd_ws");
}
