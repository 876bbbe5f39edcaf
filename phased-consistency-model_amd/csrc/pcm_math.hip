// The reference-owned phased-consistency math on [B,4,H,W] latents, fused into four launches
// and free of host round-trips (the reference rebuilds the phase-edge table with numpy and
// copies it H2D three times per step: train_pcm_lora_sd15.py:1157-1163, :322-328).
// Precision follows the reference op by op: fp32 ops are rounded individually (no FMA
// contraction) and the DDIM jump / x_prev / target are fp64 like the reference's float64
// ddim_alpha_cumprods_prev table makes them.
#include "pcm_common.h"

// Individually rounded IEEE ops: FMA contraction is switched off for this file and sqrt / divide
// are the correctly rounded forms (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt), so the
// fp32 chain is bit-identical to the reference's eager torch ops.  (HIP's __fmul_rn & co. are plain
// operators without OCML_BASIC_ROUNDED_OPERATIONS and __fsqrt_rn is the native approximate sqrt.)
#pragma clang fp contract(off)
#define __fmul_rn(a, b) pm_mul(a, b)
#define __fadd_rn(a, b) pm_add(a, b)
#define __fsub_rn(a, b) pm_sub(a, b)
#define __fdiv_rn(a, b) pm_div(a, b)
#define __fsqrt_rn(a) pm_sqrt(a)
#define __dmul_rn(a, b) pm_mul(a, b)
#define __dadd_rn(a, b) pm_add(a, b)
#define __dsub_rn(a, b) pm_sub(a, b)
#define __ddiv_rn(a, b) pm_div(a, b)
#define __dsqrt_rn(a) pm_sqrt(a)
template <typename T> __device__ __forceinline__ T pm_mul(T a, T b) { return a * b; }
template <typename T> __device__ __forceinline__ T pm_add(T a, T b) { return a + b; }
template <typename T> __device__ __forceinline__ T pm_sub(T a, T b) { return a - b; }
template <typename T> __device__ __forceinline__ T pm_div(T a, T b) { return a / b; }
__device__ __forceinline__ float pm_sqrt(float a) { return __builtin_sqrtf(a); }
__device__ __forceinline__ double pm_sqrt(double a) { return __builtin_sqrt(a); }

#define PM_LOOP(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)
static inline int pm_blocks(long n) { long b = (n + 255) / 256; return (int)(b > PCM_GRID_CAP(1024) ? PCM_GRID_CAP(1024) : (b < 1 ? 1 : b)); }

// scheduling_ddpm_modified.py:513-523: sa = acp[t]**0.5 ; sb = (1-acp[t])**0.5 ; sa*x + sb*noise
__global__ __launch_bounds__(256) void add_noise_kernel(const float* x, const float* noise, const float* acp, const int64_t* t,
                                                        float* out, int B, int ps) {
  long n = (long)B * ps;
  PM_LOOP(i, n) {
    int b = (int)(i / ps);
    float a = acp[t[b]];
    float sa = __fsqrt_rn(a), sb = __fsqrt_rn(__fsub_rn(1.0f, a));
    out[i] = __fadd_rn(__fmul_rn(sa, x[i]), __fmul_rn(sb, noise[i]));
  }
}
extern "C" int pcm_add_noise(const float* x, const float* noise, const float* acp, const int64_t* t, float* out, int B,
                             int per_sample, void* stream) {
  PCM_CHECK(x && noise && acp && t && out && B > 0 && per_sample > 0, PCM_EINVAL, "pcm_add_noise: null/empty");
  PCM_LAUNCH(add_noise_kernel, dim3(pm_blocks((long)B * per_sample)), dim3(256), 0, stream, x, noise, acp, t, out, B, per_sample);
  return pcm_post_launch("pcm_add_noise");
}

template <bool S64>
__global__ __launch_bounds__(256) void phase_jump_kernel(const float* eps, const void* sample_, const int64_t* t, const int64_t* index,
                                                         const float* acp, const double* acp_prev, const int64_t* t_prev,
                                                         const int64_t* edges, int n_edges, int target_mode, float* out,
                                                         float* coef, int64_t* end_t, int B, int ps) {
  long n = (long)B * ps;
  PM_LOOP(i, n) {
    int b = (int)(i / ps);
    int64_t idx = index[b];
    // largest phase edge <= idx (train_pcm_lora_sd15.py:329-335); c_skip = idx in edges (:250-253)
    int64_t e = edges[0];
    bool is_edge = false;
    for (int k = 0; k < n_edges; k++) {
      if (edges[k] <= idx) e = edges[k];
      is_edge = is_edge || (edges[k] == idx);
    }
    float a = acp[t[b]];
    float alpha_t = __fsqrt_rn(a), sigma_t = __fsqrt_rn(__fsub_rn(1.0f, a));  // :812-813 fp32 tables
    double ap = acp_prev[e];
    double sa = __dsqrt_rn(ap), sb = __dsqrt_rn(__dsub_rn(1.0, ap));
    float ep = eps[i];
    float prod = __fmul_rn(sigma_t, ep);
    double jump, smp;
    if (S64) {
      smp = ((const double*)sample_)[i];
      double x0 = __ddiv_rn(__dsub_rn(smp, (double)prod), (double)alpha_t);
      jump = __dadd_rn(__dmul_rn(sa, x0), __dmul_rn(sb, (double)ep));
    } else {
      float s32 = ((const float*)sample_)[i];
      smp = (double)s32;
      float x0 = __fdiv_rn(__fsub_rn(s32, prod), alpha_t);
      jump = __dadd_rn(__dmul_rn(sa, (double)x0), __dmul_rn(sb, (double)ep));
    }
    out[i] = (float)((target_mode && is_edge) ? smp : jump);
    if (i % ps == 0) {
      if (coef) coef[b] = (float)(sb - sa * (double)sigma_t / (double)alpha_t);
      if (end_t) end_t[b] = t_prev[e];
    }
  }
}
extern "C" int pcm_phase_jump(const float* eps, const void* sample, int sample_f64, const int64_t* t, const int64_t* index,
                              const float* acp, const double* acp_prev, const int64_t* t_prev, const int64_t* edges,
                              int n_edges, int target_mode, float* out, float* coef, int64_t* end_t, int B,
                              int per_sample, void* stream) {
  PCM_CHECK(eps && sample && t && index && acp && acp_prev && t_prev && edges && out && n_edges > 0 && B > 0, PCM_EINVAL,
            "pcm_phase_jump: null/empty argument");
  dim3 grid(pm_blocks((long)B * per_sample)), block(256);
  if (sample_f64) PCM_LAUNCH((phase_jump_kernel<true>), grid, block, 0, stream, eps, sample, t, index, acp, acp_prev, t_prev, edges, n_edges, target_mode, out, coef, end_t, B, per_sample);
  else PCM_LAUNCH((phase_jump_kernel<false>), grid, block, 0, stream, eps, sample, t, index, acp, acp_prev, t_prev, edges, n_edges, target_mode, out, coef, end_t, B, per_sample);
  return pcm_post_launch("pcm_phase_jump");
}

// train_pcm_lora_sd15.py:1224-1258
__global__ __launch_bounds__(256) void cfg_ddim_kernel(const float* ec, const float* eu, const float* sample, const int64_t* t,
                                                       const int64_t* index, const float* w, const float* acp,
                                                       const double* acp_prev, double* xp, float* xp32, int B, int ps) {
  long n = (long)B * ps;
  PM_LOOP(i, n) {
    int b = (int)(i / ps);
    float a = acp[t[b]];
    float alpha_t = __fsqrt_rn(a), sigma_t = __fsqrt_rn(__fsub_rn(1.0f, a));
    float s = sample[i], c = ec[i], u = eu[i], wb = w[b];
    float x0c = __fdiv_rn(__fsub_rn(s, __fmul_rn(sigma_t, c)), alpha_t);
    float x0u = __fdiv_rn(__fsub_rn(s, __fmul_rn(sigma_t, u)), alpha_t);
    float x0 = __fadd_rn(x0c, __fmul_rn(wb, __fsub_rn(x0c, x0u)));   // :1254
    float pn = __fadd_rn(c, __fmul_rn(wb, __fsub_rn(c, u)));          // :1255-1257
    double ap = acp_prev[index[b]];
    double r = __dadd_rn(__dmul_rn(__dsqrt_rn(ap), (double)x0), __dmul_rn(__dsqrt_rn(__dsub_rn(1.0, ap)), (double)pn));  // :313-319
    xp[i] = r;
    if (xp32) xp32[i] = (float)r;
  }
}
extern "C" int pcm_cfg_ddim_step(const float* eps_c, const float* eps_u, const float* sample, const int64_t* t, const int64_t* index,
                                 const float* w, const float* acp, const double* acp_prev, double* x_prev, float* x_prev_f32,
                                 int B, int per_sample, void* stream) {
  PCM_CHECK(eps_c && eps_u && sample && t && index && w && acp && acp_prev && x_prev && B > 0, PCM_EINVAL, "pcm_cfg_ddim_step: null/empty");
  PCM_LAUNCH(cfg_ddim_kernel, dim3(pm_blocks((long)B * per_sample)), dim3(256), 0, stream, eps_c, eps_u, sample, t, index, w, acp, acp_prev, x_prev, x_prev_f32, B, per_sample);
  return pcm_post_launch("pcm_cfg_ddim_step");
}

// Inference: one DDIM step of the validation sampler (log_validation, train_pcm_lora_sd15.py:120-145: DDIMScheduler with
// timestep_spacing="trailing", clip_sample=False, set_alpha_to_one=False, eta 0) fused with the pipeline's classifier-free
// guidance combine  eps = eps_u + g (eps_c - eps_u)  (eps_u == nullptr: no guidance).
//   x0 = (x - sqrt(1-a_t) eps) / sqrt(a_t);   x_prev = sqrt(a_prev) x0 + sqrt(1-a_prev) eps
__global__ __launch_bounds__(256) void sampler_ddim_kernel(const float* eps_c, const float* eps_u, const float* x, float a_t, float a_prev,
                                                           float guidance, float* out, long n) {
  const float sa = __builtin_sqrtf(a_t), sb = __builtin_sqrtf(1.0f - a_t), pa = __builtin_sqrtf(a_prev), pb = __builtin_sqrtf(1.0f - a_prev);
  PM_LOOP(i, n) {
    float e = eps_c[i];
    if (eps_u) { const float u = eps_u[i]; e = u + guidance * (e - u); }
    const float x0 = (x[i] - sb * e) / sa;
    out[i] = pa * x0 + pb * e;
  }
}
extern "C" int pcm_sampler_ddim_step(const float* eps_c, const float* eps_u, const float* x, float alpha_t, float alpha_prev, float guidance,
                                     float* out, long n, void* stream) {
  PCM_CHECK(eps_c && x && out && n > 0 && alpha_t > 0.f && alpha_t <= 1.f && alpha_prev > 0.f && alpha_prev <= 1.f, PCM_EINVAL, "pcm_sampler_ddim_step: null/empty/alpha");
  PCM_LAUNCH(sampler_ddim_kernel, dim3(pm_blocks(n)), dim3(256), 0, stream, eps_c, eps_u, x, alpha_t, alpha_prev, guidance, out, n);
  return pcm_post_launch("pcm_sampler_ddim_step");
}

// train_pcm_lora_sd15.py:1283-1293 + d loss / d eps of the online branch
__global__ __launch_bounds__(256) void loss_kernel(const float* mp, const float* tg, const float* coef, int huber, float hc,
                                                   double* loss, float* d_eps, float gscale, int B, int ps, double* part) {
  __shared__ double red[4];
  long n = (long)B * ps;
  double acc = 0.0;
  const float inv_n = 1.0f / (float)n;
  PM_LOOP(i, n) {
    float d = __fsub_rn(mp[i], tg[i]);
    float g;
    if (huber) {
      float r = __fsqrt_rn(__fadd_rn(__fmul_rn(d, d), __fmul_rn(hc, hc)));
      acc += (double)__fsub_rn(r, hc);
      g = d / r;
    } else {
      acc += (double)__fmul_rn(d, d);
      g = 2.0f * d;
    }
    if (d_eps) d_eps[i] = g * inv_n * coef[i / ps] * gscale;
  }
  // block reduce (fp64 via two fp32-pair shuffles is overkill here: go through LDS)
  double v = acc;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    if (part) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];     // reproducible form: ordered finalize (optim.hip), scaled by 1/n there
    else atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) / (double)n);
  }
}
void pcm_reduce_partials_launch(const double* part, int n, double* out, double scale, void* stream);   // optim.hip
extern "C" int pcm_consistency_loss_ws(const float* model_pred, const float* target, const float* coef, int huber, float huber_c,
                                       double* loss, float* d_eps, float grad_scale, int B, int per_sample, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  PCM_CHECK(model_pred && target && loss && B > 0 && per_sample > 0 && (!d_eps || coef), PCM_EINVAL, "pcm_consistency_loss_ws: null/empty");
  const int nb = pm_blocks((long)B * per_sample);
  PCM_CHECK(workspace && ((uintptr_t)workspace % 8) == 0 && workspace_bytes >= PCM_REDUCE_WS_BYTES && (size_t)nb * 8 <= PCM_REDUCE_WS_BYTES,
            PCM_EINVAL, "pcm_consistency_loss_ws: workspace < PCM_REDUCE_WS_BYTES");
  PCM_LAUNCH(loss_kernel, dim3(nb), dim3(256), 0, stream, model_pred, target, coef, huber, huber_c, loss, d_eps, grad_scale, B, per_sample,
             (double*)workspace);
  pcm_reduce_partials_launch((const double*)workspace, nb, loss, 1.0 / (double)((long)B * per_sample), stream);
  return pcm_post_launch("pcm_consistency_loss_ws");
}
extern "C" int pcm_consistency_loss(const float* model_pred, const float* target, const float* coef, int huber, float huber_c,
                                    double* loss, float* d_eps, float grad_scale, int B, int per_sample, void* stream) {
  PCM_CHECK(model_pred && target && loss && B > 0 && per_sample > 0 && (!d_eps || coef), PCM_EINVAL, "pcm_consistency_loss: null/empty");
  pcm_zero_async(loss, sizeof(double), stream);
  PCM_LAUNCH(loss_kernel, dim3(pm_blocks((long)B * per_sample)), dim3(256), 0, stream, model_pred, target, coef, huber, huber_c, loss, d_eps, grad_scale, B, per_sample, (double*)nullptr);
  return pcm_post_launch("pcm_consistency_loss");
}

// noise_travel (scheduling_ddpm_modified.py:526-554): r = acp[t_tgt]/acp[t_cur]; sqrt(r) x + sqrt(1-r) noise (fp32);
// also writes sqrt(r) per sample (d out / d x for the generator step's backward)
__global__ __launch_bounds__(256) void noise_travel_kernel(const float* x, const float* noise, const float* acp, const int64_t* t_cur,
                                                           const int64_t* t_tgt, float* out, float* sr, int B, int ps) {
  long n = (long)B * ps;
  PM_LOOP(i, n) {
    int b = (int)(i / ps);
    float r = __fdiv_rn(acp[t_tgt[b]], acp[t_cur[b]]);
    float sa = __fsqrt_rn(r), sb = __fsqrt_rn(__fsub_rn(1.0f, r));
    out[i] = __fadd_rn(__fmul_rn(sa, x[i]), __fmul_rn(sb, noise[i]));
    if (sr && i % ps == 0) sr[b] = sa;
  }
}
extern "C" int pcm_noise_travel(const float* x, const float* noise, const float* acp, const int64_t* t_cur, const int64_t* t_tgt,
                                float* out, float* sqrt_r, int B, int per_sample, void* stream) {
  PCM_CHECK(x && noise && acp && t_cur && t_tgt && out && B > 0 && per_sample > 0, PCM_EINVAL, "pcm_noise_travel: null/empty");
  PCM_LAUNCH(noise_travel_kernel, dim3(pm_blocks((long)B * per_sample)), dim3(256), 0, stream, x, noise, acp, t_cur, t_tgt, out, sqrt_r, B, per_sample);
  return pcm_post_launch("pcm_noise_travel");
}

// hinge losses of the latent discriminator (discriminator_sd15.py:412-434) on ONE head's logit map:
//   mode 0 (D): loss += scale * (mean relu(f + 1) + mean relu(1 - r)) ; mode 1 (G): loss += scale * mean relu(1 - f)
// and the gradients wrt the logits (scaled by gscale).  `loss` accumulates across heads (zero it once).
__global__ __launch_bounds__(256) void hinge_kernel(const float* f, const float* r, int mode, float scale, double* loss, float* df,
                                                    float* dr, float gscale, long n) {
  __shared__ double red[4];
  double acc = 0.0;
  const float inv = scale / (float)n;
  PM_LOOP(i, n) {
    float fv = f[i];
    if (mode == 0) {
      float a = fv + 1.0f, b2 = 1.0f - r[i];
      acc += (double)(a > 0.f ? a : 0.f) + (double)(b2 > 0.f ? b2 : 0.f);
      if (df) df[i] = a > 0.f ? inv * gscale : 0.f;
      if (dr) dr[i] = b2 > 0.f ? -inv * gscale : 0.f;
    } else {
      float a = 1.0f - fv;
      acc += (double)(a > 0.f ? a : 0.f);
      if (df) df[i] = a > 0.f ? -inv * gscale : 0.f;
    }
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * (double)scale / (double)n);
}
extern "C" int pcm_hinge_loss(const float* fake, const float* real, int mode, float scale, double* loss, float* d_fake, float* d_real,
                              float grad_scale, long n, void* stream) {
  PCM_CHECK(fake && loss && n > 0 && (mode == 1 || real), PCM_EINVAL, "pcm_hinge_loss: null/empty");
  PCM_LAUNCH(hinge_kernel, dim3(pm_blocks(n)), dim3(256), 0, stream, fake, real, mode, scale, loss, d_fake, d_real, grad_scale, n);
  return pcm_post_launch("pcm_hinge_loss");
}

// reproducible form (abi 5): ONE workgroup -- the launch's contribution to `loss` is a fixed-order sum, and successive heads add in stream order
extern "C" int pcm_hinge_loss_ordered(const float* fake, const float* real, int mode, float scale, double* loss, float* d_fake, float* d_real,
                                      float grad_scale, long n, void* stream) {
  PCM_CHECK(fake && loss && n > 0 && (mode == 1 || real), PCM_EINVAL, "pcm_hinge_loss_ordered: null/empty");
  PCM_LAUNCH(hinge_kernel, dim3(1), dim3(256), 0, stream, fake, real, mode, scale, loss, d_fake, d_real, grad_scale, n);
  return pcm_post_launch("pcm_hinge_loss_ordered");
}

// out[b][i] += x[b][i] * s1[b] * s2[b]
__global__ __launch_bounds__(256) void scale_add_rows_kernel(float* out, const float* x, const float* s1, const float* s2, int B, int ps) {
  long n = (long)B * ps;
  PM_LOOP(i, n) { int b = (int)(i / ps); out[i] += x[i] * s1[b] * s2[b]; }
}
extern "C" int pcm_scale_add_rows(float* out, const float* x, const float* s1, const float* s2, int B, int per_sample, void* stream) {
  PCM_CHECK(out && x && s1 && s2 && B > 0 && per_sample > 0, PCM_EINVAL, "pcm_scale_add_rows: null/empty");
  PCM_LAUNCH(scale_add_rows_kernel, dim3(pm_blocks((long)B * per_sample)), dim3(256), 0, stream, out, x, s1, s2, B, per_sample);
  return pcm_post_launch("pcm_scale_add_rows");
}
