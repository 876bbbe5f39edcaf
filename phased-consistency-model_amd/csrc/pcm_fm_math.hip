// The reference-owned flow-matching phased-consistency math of the SD3 variant on [B,16,H,W] latents:
// four element-wise launches without host round-trips (the reference rebuilds the phase-edge table with numpy and copies it
// H2D on every call of euler_style_multiphase_pred: text_to_image_sd3/train_pcm_lora_sd3.py:200-206).
// Precision follows the reference op by op: fp32 ops are rounded individually (no FMA contraction) and everything that
// touches sigma_prev is fp64, because the reference builds that table from python floats (np.asarray -> float64, :166-168)
// and torch promotes (sigma_prev - sigma) * model_pred and the sum with the sample to float64.
#include "pcm_common.h"

#pragma clang fp contract(off)

#define FM_LOOP(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)
static inline int fm_blocks(long n) { long b = (n + 255) / 256; return (int)(b > PCM_GRID_CAP(1024) ? PCM_GRID_CAP(1024) : (b < 1 ? 1 : b)); }

// train_pcm_lora_sd3.py:1291,:1301: sigmas = sigmas[index]; noisy = sigmas * noise + (1.0 - sigmas) * model_input
__global__ __launch_bounds__(256) void fm_add_noise_kernel(const float* x, const float* noise, const float* sigmas, const int64_t* index,
                                                           float* out, int B, int ps) {
  long n = (long)B * ps;
  FM_LOOP(i, n) {
    const float s = sigmas[index[i / ps]];
    const float a = s * noise[i];
    const float b = (1.0f - s) * x[i];
    out[i] = a + b;
  }
}
extern "C" int pcm_fm_add_noise(const float* x, const float* noise, const float* sigmas, const int64_t* index, float* out, int B,
                                int per_sample, void* stream) {
  PCM_CHECK(x && noise && sigmas && index && out && B > 0 && per_sample > 0, PCM_EINVAL, "pcm_fm_add_noise: null/empty");
  PCM_LAUNCH(fm_add_noise_kernel, dim3(fm_blocks((long)B * per_sample)), dim3(256), 0, stream, x, noise, sigmas, index, out, B, per_sample);
  return pcm_post_launch("pcm_fm_add_noise");
}

// EulerSolver.euler_style_multiphase_pred (:192-230): end = last phase edge <= index (the reference's mask / flipped argmax);
// sigma = target ? sigmas_prev[index] : sigmas[index];  x = sample + (sigmas_prev[end] - sigma) * model_pred   (float64)
template <bool S64>
__global__ __launch_bounds__(256) void fm_phase_jump_kernel(const void* sample_, const float* pred, const int64_t* index, const float* sigmas,
                                                            const double* sigmas_prev, const int64_t* edges, int n_edges, int target,
                                                            double* out, float* out32, int64_t* end_index, int B, int ps) {
  long n = (long)B * ps;
  FM_LOOP(i, n) {
    const int b = (int)(i / ps);
    const int64_t idx = index[b];
    int64_t end = edges[0];
    for (int e = 1; e < n_edges; e++)
      if (idx >= edges[e]) end = edges[e];
    const double sigma = target ? sigmas_prev[idx] : (double)sigmas[idx];
    const double d = sigmas_prev[end] - sigma;
    const double s = S64 ? ((const double*)sample_)[i] : (double)((const float*)sample_)[i];
    const double m = d * (double)pred[i];
    const double r = s + m;
    out[i] = r;
    if (out32) out32[i] = (float)r;
    if (end_index && i == (long)b * ps) end_index[b] = end;
  }
}
extern "C" int pcm_fm_phase_jump(const void* sample, int sample_f64, const float* model_pred, const int64_t* index, const float* sigmas,
                                 const double* sigmas_prev, const int64_t* edges, int n_edges, int target_mode, double* out,
                                 float* out_f32, int64_t* end_index, int B, int per_sample, void* stream) {
  PCM_CHECK(sample && model_pred && index && sigmas && sigmas_prev && edges && n_edges > 0 && out && B > 0 && per_sample > 0, PCM_EINVAL,
            "pcm_fm_phase_jump: null/empty");
  dim3 grid(fm_blocks((long)B * per_sample)), block(256);
  if (sample_f64) PCM_LAUNCH((fm_phase_jump_kernel<true>), grid, block, 0, stream, sample, model_pred, index, sigmas, sigmas_prev, edges, n_edges, target_mode, out, out_f32, end_index, B, per_sample);
  else PCM_LAUNCH((fm_phase_jump_kernel<false>), grid, block, 0, stream, sample, model_pred, index, sigmas, sigmas_prev, edges, n_edges, target_mode, out, out_f32, end_index, B, per_sample);
  return pcm_post_launch("pcm_fm_phase_jump");
}

// :1334-1357: teacher = cond + w * (cond - uncond) (w = 3, float32);  EulerSolver.euler_step (:184-190):
// x_prev = sample + (sigmas_prev[index] - sigmas[index]) * teacher   (float64)
__global__ __launch_bounds__(256) void fm_cfg_euler_kernel(const float* cond, const float* uncond, const float* sample, const int64_t* index,
                                                           float w, const float* sigmas, const double* sigmas_prev, double* xp, float* xp32,
                                                           int B, int ps) {
  long n = (long)B * ps;
  FM_LOOP(i, n) {
    const int64_t idx = index[i / ps];
    const float c = cond[i];
    float t = c;
    if (uncond) { const float diff = c - uncond[i]; const float wd = w * diff; t = c + wd; }
    const double d = sigmas_prev[idx] - (double)sigmas[idx];
    const double m = d * (double)t;
    const double r = (double)sample[i] + m;
    xp[i] = r;
    if (xp32) xp32[i] = (float)r;
  }
}
extern "C" int pcm_fm_cfg_euler_step(const float* cond, const float* uncond, const float* sample, const int64_t* index, float w,
                                     const float* sigmas, const double* sigmas_prev, double* x_prev, float* x_prev_f32, int B,
                                     int per_sample, void* stream) {
  PCM_CHECK(cond && sample && index && sigmas && sigmas_prev && x_prev && B > 0 && per_sample > 0, PCM_EINVAL, "pcm_fm_cfg_euler_step: null/empty");
  PCM_LAUNCH(fm_cfg_euler_kernel, dim3(fm_blocks((long)B * per_sample)), dim3(256), 0, stream, cond, uncond, sample, index, w, sigmas, sigmas_prev,
             x_prev, x_prev_f32, B, per_sample);
  return pcm_post_launch("pcm_fm_cfg_euler_step");
}

// Adversarial trainers (train_pcm_lora_sd3_adv.py:1413-1445): re-noise a phase-edge sample from sigma_end to sigma_adv,
//   x_adv = ((1 - sigma_adv) * x + (sigma_adv - sigma_end) * noise) / (1 - sigma_end),  sigma_* = sigmas_prev[end_index | adv_index]
// in float64 (x = model_pred / target and the reference's randn_like(x) are float64); float32 copy = the ``.float()`` handed to the
// discriminator; ratio[b] = (1 - sigma_adv) / (1 - sigma_end) = d x_adv / d x for the generator step's chain rule.
__global__ __launch_bounds__(256) void fm_noise_travel_kernel(const double* x, const double* noise, const double* sigmas_prev, const int64_t* end_index,
                                                              const int64_t* adv_index, double* out, float* out32, float* ratio, int B, int ps) {
  long n = (long)B * ps;
  FM_LOOP(i, n) {
    const int b = (int)(i / ps);
    const double se = sigmas_prev[end_index[b]], sa = sigmas_prev[adv_index[b]];
    const double oma = 1.0 - sa, ome = 1.0 - se, dse = sa - se;
    const double t1 = oma * x[i];
    const double t2 = dse * noise[i];
    const double r = (t1 + t2) / ome;
    if (out) out[i] = r;
    if (out32) out32[i] = (float)r;
    if (ratio && i == (long)b * ps) ratio[b] = (float)(oma / ome);
  }
}
extern "C" int pcm_fm_noise_travel(const double* x, const double* noise, const double* sigmas_prev, const int64_t* end_index,
                                   const int64_t* adv_index, double* out, float* out_f32, float* ratio, int B, int per_sample, void* stream) {
  PCM_CHECK(x && noise && sigmas_prev && end_index && adv_index && (out || out_f32) && B > 0 && per_sample > 0, PCM_EINVAL, "pcm_fm_noise_travel: null/empty");
  PCM_LAUNCH(fm_noise_travel_kernel, dim3(fm_blocks((long)B * per_sample)), dim3(256), 0, stream, x, noise, sigmas_prev, end_index, adv_index, out,
             out_f32, ratio, B, per_sample);
  return pcm_post_launch("pcm_fm_noise_travel");
}

// Inference: one step of the PCM flow-matching samplers (pcm_fm_deterministic_scheduler.py:225-233 /
// pcm_fm_stochastic_scheduler.py:225-233), float32, optionally preceded by the pipeline's classifier-free guidance combine
// v = v_u + g * (v_c - v_u) (StableDiffusion3Pipeline's denoising loop; v_u == nullptr: no guidance):
//   denoised = x - v * sigma
//   deterministic: x' = x + ((x - denoised) / sigma) * (sigma_next - sigma)        (noise == nullptr)
//   stochastic:    x' = (1 - sigma_next) * denoised + sigma_next * noise
__global__ __launch_bounds__(256) void fm_sampler_kernel(const float* v, const float* vu, float guidance, const float* x, float sigma,
                                                         float sigma_next, const float* noise, float* out, long n) {
  const float dt = sigma_next - sigma, om = 1.0f - sigma_next;
  FM_LOOP(i, n) {
    const float xs = x[i];
    float vv = v[i];
    if (vu) { const float u = vu[i]; const float d = vv - u; const float gd = guidance * d; vv = u + gd; }
    const float vs = vv * sigma;
    const float den = xs - vs;
    if (noise) {
      const float a = om * den;
      const float b = sigma_next * noise[i];
      out[i] = a + b;
    } else {
      const float diff = xs - den;
      const float der = diff / sigma;
      const float m = der * dt;
      out[i] = xs + m;
    }
  }
}
extern "C" int pcm_fm_sampler_step(const float* model_output, const float* model_output_uncond, float guidance, const float* sample, float sigma,
                                   float sigma_next, const float* noise, float* out, long n, void* stream) {
  PCM_CHECK(model_output && sample && out && n > 0 && sigma > 0.f && sigma_next >= 0.f, PCM_EINVAL, "pcm_fm_sampler_step: null/empty/sigma");
  PCM_LAUNCH(fm_sampler_kernel, dim3(fm_blocks(n)), dim3(256), 0, stream, model_output, model_output_uncond, guidance, sample, sigma, sigma_next, noise, out, n);
  return pcm_post_launch("pcm_fm_sampler_step");
}
