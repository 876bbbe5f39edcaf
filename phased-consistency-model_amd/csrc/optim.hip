// Optimizer + parameter packing for the 67 M LoRA parameters (one flat fp32 buffer):
// global-norm clip + AdamW in a single pass with the clip coefficient read from device memory
// (no host sync: accelerate.clip_grad_norm_ + optimizer.step(), train_pcm_lora_sd15.py:1297-1301),
// EMA (the reference's dead update_ema, :344-355) and the fp32 -> bf16 MFMA-operand packers.
#include "pcm_common.h"

#define OP_LOOP(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)
static inline int op_blocks(long n) { long b = (n + 255) / 256; return (int)(b > PCM_GRID_CAP(2048) ? PCM_GRID_CAP(2048) : (b < 1 ? 1 : b)); }

// part != nullptr (reproducible form): block b stores its sum to part[b]; pcm_reduce_partials_kernel adds them in order
__global__ __launch_bounds__(256) void sumsq_kernel(const float* g, double* out, long n, double* part) {
  __shared__ double red[4];
  double acc = 0.0;
  OP_LOOP(i, n) { double v = (double)g[i]; acc += v * v; }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = red[0] + red[1] + red[2] + red[3];
    if (part) part[blockIdx.x] = t; else atomicAdd(out, t);
  }
}
// out[0] = scale * sum_i part[i], one block, fixed order (thread t sums i = t, t + 256, ...; then a fixed tree)
__global__ __launch_bounds__(256) void pcm_reduce_partials_kernel(const double* part, int n, double* out, double scale) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += part[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] * scale;
}
void pcm_reduce_partials_launch(const double* part, int n, double* out, double scale, void* stream) {
  PCM_LAUNCH(pcm_reduce_partials_kernel, dim3(1), dim3(256), 0, stream, part, n, out, scale);
}
extern "C" int pcm_sumsq_f32(const float* g, double* out, long n, void* stream) {
  PCM_CHECK(g && out && n > 0, PCM_EINVAL, "pcm_sumsq_f32: null/empty");
  pcm_zero_async(out, sizeof(double), stream);
  PCM_LAUNCH(sumsq_kernel, dim3(op_blocks(n)), dim3(256), 0, stream, g, out, n, (double*)nullptr);
  return pcm_post_launch("pcm_sumsq_f32");
}
extern "C" int pcm_sumsq_f32_ws(const float* g, double* out, long n, void* workspace, size_t workspace_bytes, void* stream) {
  PCM_CHECK(g && out && n > 0 && workspace && ((uintptr_t)workspace % 8) == 0 && workspace_bytes >= PCM_REDUCE_WS_BYTES, PCM_EINVAL,
            "pcm_sumsq_f32_ws: null/empty or workspace < PCM_REDUCE_WS_BYTES");
  const int nb = op_blocks(n);          // <= 2048 partials = 16 KB
  PCM_LAUNCH(sumsq_kernel, dim3(nb), dim3(256), 0, stream, g, out, n, (double*)workspace);
  pcm_reduce_partials_launch((const double*)workspace, nb, out, 1.0, stream);
  return pcm_post_launch("pcm_sumsq_f32_ws");
}

// torch.optim.AdamW single-tensor semantics; g' = g * grad_scale * min(1, max_norm/(||g*grad_scale|| + 1e-6)).
// step / lr may come from DEVICE memory (step_dev, lr_dev) so that a captured hipGraph of the training step
// stays valid while the step count and the learning-rate schedule advance.
__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, const double* gradsq,
                                                    float max_norm, float lr, float b1, float b2, float eps, float wd,
                                                    int step, float gscale, long n, const int64_t* step_dev, const float* lr_dev,
                                                    const float* loss_scale_dev) {
  __shared__ float hyp[3];
  // loss-scaled gradients (half build, pcm_adamw_clip_step_scaled): g holds S * grad with S in device memory.  A non-finite global norm
  // means some backward value overflowed: the whole update is SKIPPED, as torch.cuda.amp.GradScaler.step does; pcm_loss_scale_update
  // (next launch) then lowers S and takes the step count back.
  if (loss_scale_dev) {
    const double gs = *gradsq;
    if (!(gs == gs) || gs > 1.7e308) return;
    gscale /= *loss_scale_dev;
  }
  if (threadIdx.x == 0) {
    float st = step_dev ? (float)(*step_dev) : (float)step;
    hyp[0] = 1.0f - powf(b1, st);
    hyp[1] = sqrtf(1.0f - powf(b2, st));
    hyp[2] = lr_dev ? *lr_dev : lr;
  }
  __syncthreads();
  const float bc1 = hyp[0], bc2_sqrt = hyp[1];
  lr = hyp[2];
  float coef = gscale;
  if (gradsq && max_norm > 0.f) {
    float norm = (float)sqrt(*gradsq) * gscale;
    float c = max_norm / (norm + 1e-6f);
    coef *= c < 1.0f ? c : 1.0f;
  }
  const float step_size = lr / bc1;
  OP_LOOP(i, n) {
    float gi = g[i] * coef;
    float pi = p[i] * (1.0f - lr * wd);
    float mi = m[i] + (gi - m[i]) * (1.0f - b1);
    float vi = v[i] * b2 + gi * gi * (1.0f - b2);
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
    m[i] = mi; v[i] = vi;
  }
}
extern "C" int pcm_adamw_clip_step(float* p, const float* g, float* m, float* v, const double* gradsq, float max_norm, float lr,
                                   float beta1, float beta2, float eps, float wd, int step, float grad_scale, long n,
                                   const int64_t* step_dev, const float* lr_dev, void* stream) {
  PCM_CHECK(p && g && m && v && n > 0 && (step >= 1 || step_dev), PCM_EINVAL, "pcm_adamw_clip_step: null/empty or step<1");
  PCM_LAUNCH(adamw_kernel, dim3(op_blocks(n)), dim3(256), 0, stream, p, g, m, v, gradsq, max_norm, lr, beta1, beta2, eps, wd, step, grad_scale, n, step_dev, lr_dev,
             (const float*)nullptr);
  return pcm_post_launch("pcm_adamw_clip_step");
}

// ---- dynamic loss scaling (the reference's fp16 runs go through accelerate's torch.cuda.amp.GradScaler: train_pcm_lora_sd15.py:1034 with
// --mixed_precision=fp16, :1296-1299 backward / clip / step).  All state lives in device memory so a captured hipGraph of the step stays valid:
//   scale[0] = S (the loss gradient is multiplied by it before the backward: pcm_scale_f32_dev), good[0] = finite steps since the last change.
extern "C" int pcm_adamw_clip_step_scaled(float* p, const float* g, float* m, float* v, const double* gradsq, float max_norm, float lr,
                                          float beta1, float beta2, float eps, float wd, float grad_scale, long n,
                                          const int64_t* step_dev, const float* lr_dev, const float* loss_scale_dev, void* stream) {
  PCM_CHECK(p && g && m && v && gradsq && n > 0 && step_dev && loss_scale_dev, PCM_EINVAL, "pcm_adamw_clip_step_scaled: null/empty");
  PCM_LAUNCH(adamw_kernel, dim3(op_blocks(n)), dim3(256), 0, stream, p, g, m, v, gradsq, max_norm, lr, beta1, beta2, eps, wd, 1, grad_scale, n, step_dev, lr_dev,
             loss_scale_dev);
  return pcm_post_launch("pcm_adamw_clip_step_scaled");
}
__global__ void loss_scale_update_kernel(float* scale, int* good, int64_t* step_dev, const double* gradsq, float growth, float backoff, int interval) {
  const double gs = *gradsq;
  if ((gs == gs) && gs <= 1.7e308) {
    if (++good[0] >= interval) { good[0] = 0; scale[0] *= growth; }
  } else {
    good[0] = 0; scale[0] *= backoff;
    if (step_dev) step_dev[0] -= 1;          // the skipped update does not count as an optimizer step
  }
}
extern "C" int pcm_loss_scale_update(float* scale, int* good_steps, int64_t* step_dev, const double* gradsq, float growth, float backoff,
                                     int interval, void* stream) {
  PCM_CHECK(scale && good_steps && gradsq && growth >= 1.f && backoff > 0.f && backoff <= 1.f && interval > 0, PCM_EINVAL, "pcm_loss_scale_update: bad arguments");
  PCM_LAUNCH(loss_scale_update_kernel, dim3(1), dim3(1), 0, stream, scale, good_steps, step_dev, gradsq, growth, backoff, interval);
  return pcm_post_launch("pcm_loss_scale_update");
}
__global__ __launch_bounds__(256) void scale_dev_kernel(float* x, const float* s, long n) {
  const float f = *s;
  OP_LOOP(i, n) x[i] *= f;
}
extern "C" int pcm_scale_f32_dev(float* x, const float* scale_dev, long n, void* stream) {
  PCM_CHECK(x && scale_dev && n > 0, PCM_EINVAL, "pcm_scale_f32_dev: null/empty");
  PCM_LAUNCH(scale_dev_kernel, dim3(op_blocks(n)), dim3(256), 0, stream, x, scale_dev, n);
  return pcm_post_launch("pcm_scale_f32_dev");
}

// gradsq != nullptr (half build): the optimizer step this EMA follows was SKIPPED when the global gradient norm is not finite
// (adamw_kernel above, GradScaler.step semantics) -- the EMA of an unchanged parameter vector must not advance either
__global__ __launch_bounds__(256) void ema_kernel(float* t, const float* s, float rate, long n, const double* gradsq) {
  if (gradsq) {
    const double gs = *gradsq;
    if (!(gs == gs) || gs > 1.7e308) return;
  }
  OP_LOOP(i, n) t[i] = t[i] * rate + s[i] * (1.0f - rate);  // targ.mul_(rate).add_(src, alpha=1-rate)
}
extern "C" int pcm_ema_update(float* target, const float* source, float rate, long n, void* stream) {
  PCM_CHECK(target && source && n > 0, PCM_EINVAL, "pcm_ema_update: null/empty");
  PCM_LAUNCH(ema_kernel, dim3(op_blocks(n)), dim3(256), 0, stream, target, source, rate, n, (const double*)nullptr);
  return pcm_post_launch("pcm_ema_update");
}
extern "C" int pcm_ema_update_gated(float* target, const float* source, float rate, long n, const double* gradsq, void* stream) {
  PCM_CHECK(target && source && n > 0, PCM_EINVAL, "pcm_ema_update_gated: null/empty");
  PCM_LAUNCH(ema_kernel, dim3(op_blocks(n)), dim3(256), 0, stream, target, source, rate, n, gradsq);
  return pcm_post_launch("pcm_ema_update_gated");
}

__global__ __launch_bounds__(256) void cast_f2b_kernel(const float* x, bf16_t* y, long n) { OP_LOOP(i, n) y[i] = f2bf(x[i]); }
__global__ __launch_bounds__(256) void cast_b2f_kernel(const bf16_t* x, float* y, long n) { OP_LOOP(i, n) y[i] = bf2f(x[i]); }
extern "C" int pcm_cast_f32_bf16(const float* x, void* y, long n, void* stream) {
  PCM_CHECK(x && y && n > 0, PCM_EINVAL, "pcm_cast_f32_bf16: null/empty");
  PCM_LAUNCH(cast_f2b_kernel, dim3(op_blocks(n)), dim3(256), 0, stream, x, (bf16_t*)y, n);
  return pcm_post_launch("pcm_cast_f32_bf16");
}
extern "C" int pcm_cast_bf16_f32(const void* x, float* y, long n, void* stream) {
  PCM_CHECK(x && y && n > 0, PCM_EINVAL, "pcm_cast_bf16_f32: null/empty");
  PCM_LAUNCH(cast_b2f_kernel, dim3(op_blocks(n)), dim3(256), 0, stream, (const bf16_t*)x, y, n);
  return pcm_post_launch("pcm_cast_bf16_f32");
}

// w [N][K] fp32 -> w_nk bf16 [N][K] * scale  and/or  w_kn bf16 [K][N] * scale
__global__ __launch_bounds__(256) void pack_linear_kernel(const float* w, bf16_t* w_nk, bf16_t* w_kn, int N, int K, float scale) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  for (int r = ty; r < 32; r += 8) {
    int n = n0 + r, k = k0 + tx;
    float v = (n < N && k < K) ? w[(size_t)n * K + k] * scale : 0.f;
    tile[r][tx] = v;
    if (w_nk && n < N && k < K) w_nk[(size_t)n * K + k] = f2bf(v);
  }
  __syncthreads();
  if (w_kn)
    for (int r = ty; r < 32; r += 8) {
      int k = k0 + r, n = n0 + tx;
      if (k < K && n < N) w_kn[(size_t)k * N + n] = f2bf(tile[tx][r]);
    }
}
extern "C" int pcm_pack_linear(const float* w, void* w_nk, void* w_kn, int N, int K, float scale, void* stream) {
  PCM_CHECK(w && (w_nk || w_kn) && N > 0 && K > 0, PCM_EINVAL, "pcm_pack_linear: null/empty");
  PCM_LAUNCH(pack_linear_kernel, dim3((K + 31) / 32, (N + 31) / 32), dim3(256), 0, stream, w, (bf16_t*)w_nk, (bf16_t*)w_kn, N, K, scale);
  return pcm_post_launch("pcm_pack_linear");
}

// w [N][C][3][3] fp32 -> fwd [N][tap][c] ; dgrad [C][tap'][n] with tap' = 8 - tap
__global__ __launch_bounds__(256) void pack_conv_kernel(const float* w, bf16_t* wf, bf16_t* wd, int N, int C, float scale, int khwc) {
  long total = (long)N * C * 9;
  OP_LOOP(i, total) {
    // iterate in fwd-output order so the fwd store is coalesced
    int c = (int)(i % C); long r = i / C;
    int tap = (int)(r % 9); int n = (int)(r / 9);
    float v = (khwc ? w[i] : w[((size_t)n * C + c) * 9 + tap]) * scale;
    bf16_t h = f2bf(v);
    if (wf) wf[i] = h;
    if (wd) wd[((size_t)c * 9 + (8 - tap)) * N + n] = h;
  }
}
// the same pack for a SOURCE already in [N][tap][C] order (the discriminator heads' internal layout, repacked after every discriminator
// step): a 32 x 32 (n, c) tile per tap goes through LDS, so both operand copies are written in 64-byte runs.  The element-order kernel
// above scatters 2-byte stores N*2 bytes apart into the dgrad copy (162 us per head conv on MI355X, 18 ms per discriminator step).
__global__ __launch_bounds__(256) void pack_conv_khwc_kernel(const float* w, bf16_t* wf, bf16_t* wd, int N, int C, float scale) {
  __shared__ float tile[32][33];
  const int tap = blockIdx.z, n0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int n = n0 + r, c = c0 + tx;
    float v = 0.f;
    if (n < N && c < C) {
      const size_t i = ((size_t)n * 9 + tap) * C + c;
      v = w[i] * scale;
      if (wf) wf[i] = f2bf(v);
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  if (wd)
    for (int r = ty; r < 32; r += 8) {
      const int c = c0 + r, n = n0 + tx;
      if (c < C && n < N) wd[((size_t)c * 9 + (8 - tap)) * N + n] = f2bf(tile[tx][r]);
    }
}
extern "C" int pcm_pack_conv3x3(const float* w, void* w_fwd, void* w_dgrad, int N, int C, float scale, int src_khwc, void* stream) {
  PCM_CHECK(w && (w_fwd || w_dgrad) && N > 0 && C > 0, PCM_EINVAL, "pcm_pack_conv3x3: null/empty");
  if (src_khwc && (N + 31) / 32 <= 65535) {
    PCM_LAUNCH(pack_conv_khwc_kernel, dim3((C + 31) / 32, (N + 31) / 32, 9), dim3(256), 0, stream, w, (bf16_t*)w_fwd, (bf16_t*)w_dgrad, N, C, scale);
    return pcm_post_launch("pcm_pack_conv3x3");
  }
  PCM_LAUNCH(pack_conv_kernel, dim3(op_blocks((long)N * C * 9)), dim3(256), 0, stream, w, (bf16_t*)w_fwd, (bf16_t*)w_dgrad, N, C, scale, src_khwc);
  return pcm_post_launch("pcm_pack_conv3x3");
}

// ---- segmented pack: ALL LoRA operand copies refreshed by ONE launch after the optimizer step ----
// desc d describes a strided 2-D fp32 matrix src[R][Cc] (row stride lds) inside the flat parameter
// buffer; it is written as bf16 (x scale) to dst_copy[R][Cc] (row stride ldc) and/or transposed to
// dst_t[Cc][R] (row stride ldt) inside the flat operand buffer.  Blocks map to (desc, 32x32 tile)
// through the prefix table blk_start.
__global__ __launch_bounds__(256) void pack_segmented_kernel(const float* src_base, bf16_t* dst_base, const pcm_pack_desc* descs,
                                                             const int* blk_start, int ndesc) {
  __shared__ float tile[32][33];
  const int bid = blockIdx.x;
  int lo = 0, hi = ndesc - 1;
  while (lo < hi) {  // last desc with blk_start[d] <= bid
    int mid = (lo + hi + 1) >> 1;
    if (blk_start[mid] <= bid) lo = mid; else hi = mid - 1;
  }
  const pcm_pack_desc d = descs[lo];
  const int t = bid - blk_start[lo];
  const int tiles_c = (d.Cc + 31) >> 5;
  const int r0 = (t / tiles_c) * 32, c0 = (t % tiles_c) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* src = src_base + d.src_off;
  for (int r = ty; r < 32; r += 8) {
    int rr = r0 + r, cc = c0 + tx;
    float v = (rr < d.R && cc < d.Cc) ? src[(size_t)rr * d.lds + cc] * d.scale : 0.f;
    tile[r][tx] = v;
    if (d.dst_copy_off >= 0 && rr < d.R && cc < d.Cc) dst_base[d.dst_copy_off + (size_t)rr * d.ldc + cc] = f2bf(v);
  }
  __syncthreads();
  if (d.dst_t_off >= 0)
    for (int r = ty; r < 32; r += 8) {
      int cc = c0 + r, rr = r0 + tx;
      if (cc < d.Cc && rr < d.R) dst_base[d.dst_t_off + (size_t)cc * d.ldt + rr] = f2bf(tile[tx][r]);
    }
}
extern "C" int pcm_pack_segmented(const float* src_base, void* dst_base, const pcm_pack_desc* descs, const int* blk_start,
                                  int ndesc, int total_blocks, void* stream) {
  PCM_CHECK(src_base && dst_base && descs && blk_start && ndesc > 0 && total_blocks > 0, PCM_EINVAL, "pcm_pack_segmented: null/empty");
  PCM_LAUNCH(pack_segmented_kernel, dim3(total_blocks), dim3(256), 0, stream, src_base, (bf16_t*)dst_base, descs, blk_start, ndesc);
  return pcm_post_launch("pcm_pack_segmented");
}
