// Element-wise pieces of the MMDiT (SD3Transformer2DModel) blocks that the UNet path does not have:
// gated residual of the adaLN-Zero blocks, tanh-GELU, 2x2 patchify / unpatchify, sinusoidal projection of FLOAT timesteps.
// All HBM-bound: 16 B per lane, grid-stride.
#include "pcm_common.h"

#define MM_LOOP(v, n) for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < (n); v += (long)gridDim.x * blockDim.x)
static inline int mm_blocks(long n) { long b = (n + 255) / 256; return (int)(b > PCM_GRID_CAP(4096) ? PCM_GRID_CAP(4096) : (b < 1 ? 1 : b)); }

__device__ __forceinline__ void mm_unpack8(const uint4& v, float (&f)[8]) {
  f[0] = bf2f((bf16_t)(v.x & 0xffff)); f[1] = bf2f((bf16_t)(v.x >> 16));
  f[2] = bf2f((bf16_t)(v.y & 0xffff)); f[3] = bf2f((bf16_t)(v.y >> 16));
  f[4] = bf2f((bf16_t)(v.z & 0xffff)); f[5] = bf2f((bf16_t)(v.z >> 16));
  f[6] = bf2f((bf16_t)(v.w & 0xffff)); f[7] = bf2f((bf16_t)(v.w >> 16));
}
__device__ __forceinline__ uint4 mm_pack8(const float (&f)[8]) {
  return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7]));
}

// out[m][c] = (res ? res[m][c] : 0) + gate[m / rows_per_batch][c] * y[m][c]      (JointTransformerBlock: x + gate_msa * attn, x + gate_mlp * ff;
// with res == nullptr the same launch is the backward of the gated branch, d_y = gate * d_out)
__global__ __launch_bounds__(256) void rowgate_kernel(const uint4* y, const float* gate, const uint4* res, uint4* out, int M, int CV, int rpb) {
  const unsigned nvec = (unsigned)M * CV, S = gridDim.x * blockDim.x;
  unsigned v = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned r = v / CV, c = v - r * CV;
  const unsigned sr = S / CV, sc = S - sr * CV;
  for (; v < nvec; v += S) {
    float f[8], o[8];
    mm_unpack8(y[v], f);
    const float* g = gate + ((size_t)(r / rpb) * CV + c) * 8;
    const float4 g0 = *(const float4*)g, g1 = *(const float4*)(g + 4);
    const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    if (res) {
      float q[8];
      mm_unpack8(res[v], q);
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = q[e] + gg[e] * f[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = gg[e] * f[e];
    }
    out[v] = mm_pack8(o);
    c += sc; r += sr; if (c >= (unsigned)CV) { c -= CV; r++; }
  }
}
extern "C" int pcm_rowgate_fma(const void* y, const float* gate, const void* res, void* out, int M, int C, int rows_per_batch, void* stream) {
  PCM_CHECK(y && gate && out && M > 0 && C > 0 && (C % 8) == 0 && rows_per_batch > 0 && (M % rows_per_batch) == 0 && (long)M * (C / 8) < (1L << 31) - (1L << 22),
            PCM_EINVAL, "pcm_rowgate_fma: need C%%8==0, M == B*rows_per_batch");
  PCM_CHECK(PCM_ALIGNED16(y) && PCM_ALIGNED16(out) && PCM_ALIGNED16(gate) && (!res || PCM_ALIGNED16(res)), PCM_EALIGN, "pcm_rowgate_fma: alignment");
  PCM_LAUNCH(rowgate_kernel, dim3(mm_blocks((long)M * (C / 8))), dim3(256), 0, stream, (const uint4*)y, gate, (const uint4*)res, (uint4*)out, M, C / 8, rows_per_batch);
  return pcm_post_launch("pcm_rowgate_fma");
}

// Gradients of the per-sample modulation vectors (needed when the adaLN projections norm1(.context).linear carry LoRA factors,
// train_pcm_lora_sd3_adv.py:992-1015): column reductions over the rows of each sample,
//   out_a[b][c] = sum_l dy[b,l,c] * u[b,l,c],   u = (x - mean[row]) * rstd[row]  (LayerNorm scale: d gamma)  or  u = x (gate: d gate)
//   out_b[b][c] = sum_l dy[b,l,c]                (LayerNorm shift: d beta; optional)
// thread -> fixed 8-channel vector, strided over rows; block reduction through a float4 LDS image; fp32 atomics into the zeroed outputs.
// part != nullptr (reproducible form): block (chunk, b, zc) stores to part[chunk][a | b][B][C]; an ordered finalize adds the chunks
__global__ __launch_bounds__(256) void mod_grad_kernel(const bf16_t* x, const bf16_t* dy, const float* mean, const float* rstd, float* out_a,
                                                       float* out_b, int L, int C, int CVL, int rpb_blk, float* part) {
  __shared__ __attribute__((aligned(16))) float4 pbuf[4][256];
  const int b = blockIdx.y, zc = blockIdx.z;
  const int cvl = threadIdx.x % CVL, pl = threadIdx.x / CVL, k = blockDim.x / CVL;
  const int c0 = (zc * CVL + cvl) * 8;
  const int l_begin = blockIdx.x * rpb_blk;
  int l_end = l_begin + rpb_blk; if (l_end > L) l_end = L;
  float sa[8], sb[8];
#pragma unroll
  for (int e = 0; e < 8; e++) { sa[e] = 0.f; sb[e] = 0.f; }
  for (int l = l_begin + pl; l < l_end; l += k) {
    const size_t row = (size_t)b * L + l;
    float xv[8], dv[8];
    mm_unpack8(*(const uint4*)(x + row * C + c0), xv);
    mm_unpack8(*(const uint4*)(dy + row * C + c0), dv);
    float mu = 0.f, rs = 1.f;
    if (mean) { mu = mean[row]; rs = rstd[row]; }
#pragma unroll
    for (int e = 0; e < 8; e++) { sa[e] += dv[e] * ((xv[e] - mu) * rs); sb[e] += dv[e]; }
  }
  pbuf[0][threadIdx.x] = make_float4(sa[0], sa[1], sa[2], sa[3]);
  pbuf[1][threadIdx.x] = make_float4(sa[4], sa[5], sa[6], sa[7]);
  pbuf[2][threadIdx.x] = make_float4(sb[0], sb[1], sb[2], sb[3]);
  pbuf[3][threadIdx.x] = make_float4(sb[4], sb[5], sb[6], sb[7]);
  __syncthreads();
  for (int t2 = threadIdx.x; t2 < 4 * CVL; t2 += blockDim.x) {
    const int cv2 = t2 % CVL, j = t2 / CVL;
    if (j >= 2 && !out_b) continue;
    float4 acc = pbuf[j][cv2];
    for (int q = 1; q < k; q++) {
      const float4 v = pbuf[j][q * CVL + cv2];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const size_t col = (size_t)b * C + (zc * CVL + cv2) * 8 + 4 * (j & 1);
    if (part) {
      *(float4*)(part + ((size_t)blockIdx.x * 2 + (j < 2 ? 0 : 1)) * gridDim.y * C + col) = acc;
      continue;
    }
    float* o = (j < 2 ? out_a : out_b) + col;
    atomicAdd(o + 0, acc.x); atomicAdd(o + 1, acc.y); atomicAdd(o + 2, acc.z); atomicAdd(o + 3, acc.w);
  }
}
static void mod_grad_geometry(int B, int L, int C, int* split_, int* CVL_, int* k_, int* chunks_, int* rpb_) {
  const int CV = C / 8;
  int split = 1;
  while (CV / split > 256 || (CV % split) != 0) split++;
  const int CVL = CV / split, k = 256 / CVL > 0 ? 256 / CVL : 1;
  int chunks = (PCM_GRID_CAP(1024) + B * split - 1) / (B * split);
  const int maxc = (L + k - 1) / k; if (chunks > maxc) chunks = maxc; if (chunks < 1) chunks = 1;
  const int rpb = (L + chunks - 1) / chunks; chunks = (L + rpb - 1) / rpb;
  *split_ = split; *CVL_ = CVL; *k_ = k; *chunks_ = chunks; *rpb_ = rpb;
}
extern "C" int pcm_mod_grad(const void* x, const void* dy, const float* mean, const float* rstd, float* out_a, float* out_b, int B, int L, int C,
                            void* stream) {
  PCM_CHECK(x && dy && out_a && B > 0 && L > 0 && C > 0 && (C % 8) == 0 && (!mean == !rstd), PCM_EINVAL, "pcm_mod_grad: null/empty, C%%8");
  PCM_CHECK(PCM_ALIGNED16(x) && PCM_ALIGNED16(dy), PCM_EALIGN, "pcm_mod_grad: alignment");
  int split, CVL, k, chunks, rpb;
  mod_grad_geometry(B, L, C, &split, &CVL, &k, &chunks, &rpb);
  pcm_zero_async(out_a, sizeof(float) * (size_t)B * C, stream);
  if (out_b) pcm_zero_async(out_b, sizeof(float) * (size_t)B * C, stream);
  PCM_LAUNCH(mod_grad_kernel, dim3(chunks, B, split), dim3(CVL * k), 0, stream, (const bf16_t*)x, (const bf16_t*)dy, mean, rstd, out_a, out_b, L, C, CVL, rpb, (float*)nullptr);
  return pcm_post_launch("pcm_mod_grad");
}
// reproducible form (abi 5): per-block partials in the caller's workspace + an ordered finalize (the outputs are overwritten, as above)
extern "C" size_t pcm_mod_grad_workspace_bytes(int B, int L, int C) {
  if (B <= 0 || L <= 0 || C <= 0 || (C % 8)) return 0;
  int split, CVL, k, chunks, rpb;
  mod_grad_geometry(B, L, C, &split, &CVL, &k, &chunks, &rpb);
  return sizeof(float) * (size_t)chunks * 2 * B * C;
}
extern "C" int pcm_mod_grad_ws(const void* x, const void* dy, const float* mean, const float* rstd, float* out_a, float* out_b, int B, int L, int C,
                               void* workspace, size_t workspace_bytes, void* stream) {
  PCM_CHECK(x && dy && out_a && workspace && B > 0 && L > 0 && C > 0 && (C % 8) == 0 && (!mean == !rstd), PCM_EINVAL, "pcm_mod_grad_ws: null/empty, C%%8");
  PCM_CHECK(PCM_ALIGNED16(x) && PCM_ALIGNED16(dy) && PCM_ALIGNED16(workspace), PCM_EALIGN, "pcm_mod_grad_ws: alignment");
  int split, CVL, k, chunks, rpb;
  mod_grad_geometry(B, L, C, &split, &CVL, &k, &chunks, &rpb);
  PCM_CHECK(workspace_bytes >= sizeof(float) * (size_t)chunks * 2 * B * C, PCM_EINVAL, "pcm_mod_grad_ws: workspace too small");
  PCM_LAUNCH(mod_grad_kernel, dim3(chunks, B, split), dim3(CVL * k), 0, stream, (const bf16_t*)x, (const bf16_t*)dy, mean, rstd, out_a, out_b, L, C, CVL, rpb, (float*)workspace);
  const long n = (long)B * C;
  pcm_partials_finalize((const float*)workspace, 2 * n, out_a, chunks, n, 0, stream);
  if (out_b) pcm_partials_finalize((const float*)workspace + n, 2 * n, out_b, chunks, n, 0, stream);
  return pcm_post_launch("pcm_mod_grad_ws");
}

// FeedForward(activation_fn="gelu-approximate"): y = 0.5 x (1 + tanh(k (x + 0.044715 x^3))), k = sqrt(2/pi)
__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.0f + tanhf(u));
}
__device__ __forceinline__ float gelu_tanh_grad_f(float x) {
  const float x2 = x * x;
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x2);
  const float t = tanhf(u);
  const float du = 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * x2);
  return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * du;
}
__global__ __launch_bounds__(256) void gelu_tanh_fwd_kernel(const uint4* x, uint4* y, long nvec) {
  MM_LOOP(v, nvec) {
    float f[8];
    mm_unpack8(x[v], f);
#pragma unroll
    for (int e = 0; e < 8; e++) f[e] = gelu_tanh_f(f[e]);
    y[v] = mm_pack8(f);
  }
}
__global__ __launch_bounds__(256) void gelu_tanh_bwd_kernel(const uint4* x, const uint4* dy, uint4* dx, long nvec) {
  MM_LOOP(v, nvec) {
    float f[8], d[8];
    mm_unpack8(x[v], f); mm_unpack8(dy[v], d);
#pragma unroll
    for (int e = 0; e < 8; e++) d[e] *= gelu_tanh_grad_f(f[e]);
    dx[v] = mm_pack8(d);
  }
}
extern "C" int pcm_gelu_tanh_fwd(const void* x, void* y, long n, void* stream) {
  PCM_CHECK(x && y && n > 0 && (n % 8) == 0 && PCM_ALIGNED16(x) && PCM_ALIGNED16(y), PCM_EINVAL, "pcm_gelu_tanh_fwd: n%%8, alignment");
  PCM_LAUNCH(gelu_tanh_fwd_kernel, dim3(mm_blocks(n / 8)), dim3(256), 0, stream, (const uint4*)x, (uint4*)y, n / 8);
  return pcm_post_launch("pcm_gelu_tanh_fwd");
}
extern "C" int pcm_gelu_tanh_bwd(const void* x, const void* dy, void* dx, long n, void* stream) {
  PCM_CHECK(x && dy && dx && n > 0 && (n % 8) == 0 && PCM_ALIGNED16(x) && PCM_ALIGNED16(dy) && PCM_ALIGNED16(dx), PCM_EINVAL, "pcm_gelu_tanh_bwd: n%%8, alignment");
  PCM_LAUNCH(gelu_tanh_bwd_kernel, dim3(mm_blocks(n / 8)), dim3(256), 0, stream, (const uint4*)x, (const uint4*)dy, (uint4*)dx, n / 8);
  return pcm_post_launch("pcm_gelu_tanh_bwd");
}

// 2x2 patches of an fp32 NCHW image <-> token rows.  Token row = (b, hp, wp), hp = h/2, wp = w/2; K = 4*C columns in one of two orders:
//   order 0 (c, p, q): the flattened Conv2d(k=2, s=2) weight of PatchEmbed.proj ([N][C][2][2])           -- forward input
//   order 1 (p, q, c): the layout proj_out produces for the unpatchify einsum "nhwpqc->nchpwq"          -- output / its gradient
__global__ __launch_bounds__(256) void patchify_kernel(const float* img, bf16_t* tok, int B, int C, int H, int W, int order) {
  const int Hp = H / 2, Wp = W / 2, K = 4 * C;
  const long n = (long)B * Hp * Wp * K;
  MM_LOOP(i, n) {
    const int k = (int)(i % K); long t = i / K;
    const int wp = (int)(t % Wp); t /= Wp;
    const int hp = (int)(t % Hp); const int b = (int)(t / Hp);
    int c, p, q;
    if (order == 0) { c = k >> 2; p = (k >> 1) & 1; q = k & 1; } else { p = k / (2 * C); q = (k / C) & 1; c = k % C; }
    tok[i] = f2bf(img[(((size_t)b * C + c) * H + 2 * hp + p) * W + 2 * wp + q]);
  }
}
__global__ __launch_bounds__(256) void unpatchify_kernel(const float* tok, float* img, int B, int C, int H, int W, int order) {
  const int Hp = H / 2, Wp = W / 2, K = 4 * C;
  const long n = (long)B * C * H * W;
  MM_LOOP(i, n) {
    const int w = (int)(i % W); long t = i / W;
    const int h = (int)(t % H); t /= H;
    const int c = (int)(t % C); const int b = (int)(t / C);
    const int k = order == 0 ? c * 4 + (h & 1) * 2 + (w & 1) : ((h & 1) * 2 + (w & 1)) * C + c;
    img[i] = tok[(((size_t)b * Hp + (h >> 1)) * Wp + (w >> 1)) * K + k];
  }
}
extern "C" int pcm_patchify2x2(const float* img, void* tokens, int B, int C, int H, int W, int order, void* stream) {
  PCM_CHECK(img && tokens && B > 0 && C > 0 && H > 0 && W > 0 && (H % 2) == 0 && (W % 2) == 0 && (order == 0 || order == 1), PCM_EINVAL,
            "pcm_patchify2x2: even H, W; order 0|1");
  PCM_LAUNCH(patchify_kernel, dim3(mm_blocks((long)B * C * H * W)), dim3(256), 0, stream, img, (bf16_t*)tokens, B, C, H, W, order);
  return pcm_post_launch("pcm_patchify2x2");
}
extern "C" int pcm_unpatchify2x2(const float* tokens, float* img, int B, int C, int H, int W, int order, void* stream) {
  PCM_CHECK(img && tokens && B > 0 && C > 0 && H > 0 && W > 0 && (H % 2) == 0 && (W % 2) == 0 && (order == 0 || order == 1), PCM_EINVAL,
            "pcm_unpatchify2x2: even H, W; order 0|1");
  PCM_LAUNCH(unpatchify_kernel, dim3(mm_blocks((long)B * C * H * W)), dim3(256), 0, stream, tokens, img, B, C, H, W, order);
  return pcm_post_launch("pcm_unpatchify2x2");
}

// Timesteps(num_channels=dim, flip_sin_to_cos=True, downscale_freq_shift=0) on FLOAT timesteps (sigma * 1000 of the flow-matching
// trainer, train_pcm_lora_sd3.py:1295-1300): [cos(t f_i) | sin(t f_i)], f_i = exp(-ln(1e4) i / half)
__global__ __launch_bounds__(256) void temb_f32_kernel(const float* t, bf16_t* out, int B, int dim) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, j = i - b * half;
  const float f = expf(-9.210340371976184f * (float)j / (float)half);
  const float arg = t[b] * f;
  out[(size_t)b * dim + j] = f2bf(cosf(arg));
  out[(size_t)b * dim + half + j] = f2bf(sinf(arg));
}
extern "C" int pcm_timestep_embedding_f32(const float* t, void* out, int B, int dim, void* stream) {
  PCM_CHECK(t && out && B > 0 && dim > 0 && (dim % 2) == 0, PCM_EINVAL, "pcm_timestep_embedding_f32: dim even");
  PCM_LAUNCH(temb_f32_kernel, dim3((B * dim / 2 + 255) / 256), dim3(256), 0, stream, t, (bf16_t*)out, B, dim);
  return pcm_post_launch("pcm_timestep_embedding_f32");
}
