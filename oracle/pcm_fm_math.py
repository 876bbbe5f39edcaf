"""Restatement (torch / numpy, CPU) of the reference-OWNED flow-matching phased-consistency math of the SD3 variant
(SURVEY §8f rank 4): the Euler solver of the trainer, the step's element-wise expressions, and the two PCM samplers.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Each function cites the reference file:line it follows (paths relative
to /root/reference/code/text_to_image_sd3/).  Pinned bit-exactly against the reference's own source (oracle/ref_slice.py)
through tests/golden/pcm_fm_golden.safetensors (tests/test_oracle_pinning.py, tests/golden/make_golden_sd3.py).

dtype quirks that are part of the behaviour (and of the golden vectors):
  * ``sigmas`` is float32, ``sigmas_prev`` is float64 (np.asarray over python floats, train_pcm_lora_sd3.py:168-170), so every
    expression that touches sigma_prev -- euler_step, both multiphase jumps, timesteps_prev -- is evaluated in float64;
  * the samplers work in float32 and divide by sigma before multiplying by dt (pcm_fm_deterministic_scheduler.py:228-232).
"""
import numpy as np
import torch


def flow_sigmas(num_train_timesteps=1000, shift=3.0):
    """The table handed to EulerSolver (train_pcm_lora_sd3.py:961-965: ``noise_scheduler.sigmas.numpy()[::-1]``) -- the
    shifted flow-matching sigmas, same expression as pcm_fm_deterministic_scheduler.py:47-52; ascending, float32."""
    t = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
    s = torch.from_numpy(t).to(torch.float32) / num_train_timesteps
    s = shift * s / (1 + (shift - 1) * s)
    return s.numpy()[::-1].copy()


def extract_into_tensor(a, t, x_shape):
    """train_pcm_lora_sd3.py:153-156."""
    return a.gather(-1, t).reshape(t.shape[0], *((1,) * (len(x_shape) - 1)))


def phase_edges(num_euler, multiphase):
    """train_pcm_lora_sd3.py:200-203: floor(linspace(0, num_euler, multiphase, endpoint=False))."""
    return torch.from_numpy(np.floor(np.linspace(0, num_euler, num=multiphase, endpoint=False)).astype(np.int64)).long()


class EulerSolver:
    """train_pcm_lora_sd3.py:158-230."""

    def __init__(self, sigmas, timesteps=1000, euler_timesteps=50):
        step_ratio = timesteps // euler_timesteps                                                   # :160
        et = (np.arange(1, euler_timesteps + 1) * step_ratio).round().astype(np.int64) - 1          # :161-163
        self.euler_timesteps = torch.from_numpy(et).long()
        self.euler_timesteps_prev = torch.from_numpy(np.asarray([0] + et[:-1].tolist())).long()    # :164
        self.sigmas = torch.from_numpy(np.ascontiguousarray(sigmas[et]))                            # :165  float32
        self.sigmas_prev = torch.from_numpy(np.asarray([sigmas[0]] + sigmas[et[:-1]].tolist()))     # :166-168  float64

    def euler_step(self, sample, model_pred, timestep_index):
        """:184-190 -- x_prev = sample + (sigma_prev - sigma) * model_pred  (float64 through sigma_prev)."""
        sigma = extract_into_tensor(self.sigmas, timestep_index, model_pred.shape)
        sigma_prev = extract_into_tensor(self.sigmas_prev, timestep_index, model_pred.shape)
        return sample + (sigma_prev - sigma) * model_pred

    def euler_style_multiphase_pred(self, sample, model_pred, timestep_index, multiphase, is_target=False):
        """:192-230 -- jump to the left edge of the phase that contains timestep_index; the target branch starts from
        sigma_prev[index] (x_prev lives one Euler step earlier)."""
        edges = phase_edges(len(self.euler_timesteps), multiphase)
        # last edge <= index  (:205-210: mask, flipped argmax)
        end = edges[(timestep_index.unsqueeze(1) >= edges.unsqueeze(0)).long().sum(1) - 1]
        sigma = extract_into_tensor(self.sigmas_prev if is_target else self.sigmas, timestep_index, sample.shape)
        sigma_prev = extract_into_tensor(self.sigmas_prev, end, sample.shape)
        return sample + (sigma_prev - sigma) * model_pred, end


def fm_timesteps(solver, index, num_train_timesteps=1000):
    """train_pcm_lora_sd3.py:1291-1300 -- (timesteps float32, timesteps_prev float64) fed to the transformer."""
    s = solver.sigmas[index]
    sp = solver.sigmas_prev[index]
    return s * num_train_timesteps, sp * num_train_timesteps


def fm_add_noise(solver, model_input, noise, index):
    """train_pcm_lora_sd3.py:1301 -- noisy = sigma * noise + (1 - sigma) * x  (float32)."""
    s = extract_into_tensor(solver.sigmas, index, model_input.shape)
    return s * noise + (1.0 - s) * model_input


def fm_cfg(cond, uncond, w=3):
    """train_pcm_lora_sd3.py:1334,:1352-1354 -- teacher = cond + w * (cond - uncond), w fixed to 3."""
    return cond + w * (cond - uncond)


def fm_noise_travel(solver, x, noise, end_index, adv_index):
    """train_pcm_lora_sd3_adv.py:1413-1445 -- re-noise from sigma_prev[end_index] to sigma_prev[adv_index] (float64)."""
    s_end = extract_into_tensor(solver.sigmas_prev, end_index, x.shape)
    s_adv = extract_into_tensor(solver.sigmas_prev, adv_index, x.shape)
    return ((1 - s_adv) * x + (s_adv - s_end) * noise) / (1 - s_end)


def huber_loss(model_pred, target, huber_c=0.001):
    """train_pcm_lora_sd3.py:1374-1379."""
    return torch.mean(torch.sqrt((model_pred.float() - target.float()) ** 2 + huber_c ** 2) - huber_c)


class PCMFMSampler:
    """pcm_fm_deterministic_scheduler.py:35-242 / pcm_fm_stochastic_scheduler.py (same file but for the update at :228-233).
    ``stochastic=False``: x += ((x - denoised) / sigma) * (sigma_next - sigma); ``True``: x = (1 - sigma_next) * denoised +
    sigma_next * noise, with denoised = x - v * sigma."""

    def __init__(self, num_train_timesteps=1000, shift=1.0, pcm_timesteps=50, stochastic=False):
        self.num_train_timesteps, self.pcm_timesteps, self.stochastic = num_train_timesteps, pcm_timesteps, stochastic
        full = flow_sigmas(num_train_timesteps, shift)                                              # ascending (:47-52)
        et = (np.arange(1, pcm_timesteps + 1) * (num_train_timesteps // pcm_timesteps)).round().astype(np.int64) - 1   # :53-55
        self.sigmas = torch.from_numpy(full[et][::-1].copy())                                       # :56-57 descending
        self.step_index = None

    def set_timesteps(self, num_inference_steps):
        """:120-146."""
        idx = torch.from_numpy(np.floor(np.linspace(0, self.pcm_timesteps, num=num_inference_steps, endpoint=False)).astype(np.int64))
        s = self.sigmas[idx]
        self.timesteps = s * self.num_train_timesteps
        self.sigmas_ = torch.cat([s, torch.zeros(1)])
        self.step_index = 0

    def step(self, model_output, sample, noise=None):
        """:172-239; ``noise`` stands for the stochastic variant's randn_like draw."""
        sample = sample.to(torch.float32)
        sigma = self.sigmas_[self.step_index]
        denoised = sample - model_output * sigma
        if self.stochastic:
            sigma_prev = self.sigmas_[self.step_index + 1]
            prev = (1 - sigma_prev) * denoised + sigma_prev * noise
        else:
            derivative = (sample - denoised) / sigma
            prev = sample + derivative * (self.sigmas_[self.step_index + 1] - sigma)
        self.step_index += 1
        return prev.to(model_output.dtype)


def fm_sample(model_fn, prompt_embeds, pooled, uncond_embeds, uncond_pooled, latents, num_inference_steps, guidance_scale, shift=3.0,
              pcm_timesteps=100, stochastic=False, noises=None):
    """StableDiffusion3Pipeline's denoising loop around the PCM sampler (train_pcm_lora_sd3.py:1433-1470): CFG as
    uncond + g * (text - uncond) when guidance_scale > 1.  ``model_fn(x, t, ctx, pooled)`` -> velocity prediction."""
    sm = PCMFMSampler(1000, shift, pcm_timesteps, stochastic=stochastic)
    sm.set_timesteps(num_inference_steps)
    x = latents
    for i, t in enumerate(sm.timesteps):
        tt = t.expand(x.shape[0])
        v = model_fn(x, tt, prompt_embeds, pooled)
        if guidance_scale > 1.0 and uncond_embeds is not None:
            u = model_fn(x, tt, uncond_embeds, uncond_pooled)
            v = u + guidance_scale * (v - u)
        x = sm.step(v, x, None if noises is None else noises[i])
    return x
