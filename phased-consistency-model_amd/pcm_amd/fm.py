"""Flow-matching phased-consistency math of the SD3 variant, host side.

Mirrors the reference's own objects for this path (code/text_to_image_sd3/):
  * ``EulerSolver``                     train_pcm_lora_sd3.py:158-230  (tables + euler_step + euler_style_multiphase_pred)
  * the step's element-wise expressions train_pcm_lora_sd3.py:1285-1372 (index -> sigmas / timesteps, noisy input, fixed-w CFG)
  * ``PCMFMDeterministicScheduler`` / ``PCMFMStochasticScheduler``     pcm_fm_*_scheduler.py:35-242
with the arithmetic in the HIP kernels of csrc/pcm_fm_math.hip (bit-exact against the reference's source on the golden
vectors of tests/golden/pcm_fm_golden.safetensors).  The tables are built once on the host exactly as the reference builds
them -- including ``sigmas_prev`` being float64 -- and live on the device; no per-step numpy / H2D traffic.

The transformer these expressions wrap is pcm_amd/mmdit.py; the step that strings them together is pcm_amd/trainer_sd3.py.
"""
import numpy as np
import torch

from . import capi
from .ops import ptr, _stream


def flow_sigmas(num_train_timesteps=1000, shift=3.0):
    """Ascending float32 sigma table the trainer hands to EulerSolver (train_pcm_lora_sd3.py:961-965); the shifted
    flow-matching schedule, same expression as pcm_fm_deterministic_scheduler.py:47-52."""
    t = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
    s = torch.from_numpy(t).to(torch.float32) / num_train_timesteps
    s = shift * s / (1 + (shift - 1) * s)
    return s.numpy()[::-1].copy()


class EulerSolver:
    """train_pcm_lora_sd3.py:158-230 with device-resident tables and fused kernels."""

    def __init__(self, sigmas, timesteps=1000, euler_timesteps=50, device="cuda"):
        self.device = torch.device(device)
        self.step_ratio = timesteps // euler_timesteps
        et = (np.arange(1, euler_timesteps + 1) * self.step_ratio).round().astype(np.int64) - 1
        self.num_euler = euler_timesteps
        self.euler_timesteps = torch.from_numpy(et).long().to(self.device)
        self.euler_timesteps_prev = torch.from_numpy(np.asarray([0] + et[:-1].tolist())).long().to(self.device)
        self.sigmas = torch.from_numpy(np.ascontiguousarray(sigmas[et])).to(self.device)                       # float32
        self.sigmas_prev = torch.from_numpy(np.asarray([sigmas[0]] + sigmas[et[:-1]].tolist())).to(self.device)  # float64 (:166-168)
        assert self.sigmas.dtype == torch.float32 and self.sigmas_prev.dtype == torch.float64
        self._edges = {}

    def edges(self, multiphase):
        """floor(linspace(0, E, multiphase, endpoint=False)) (:200-203), built once per multiphase and kept on the device."""
        if multiphase not in self._edges:
            e = np.floor(np.linspace(0, self.num_euler, num=multiphase, endpoint=False)).astype(np.int64)
            self._edges[multiphase] = torch.from_numpy(e).long().to(self.device)
        return self._edges[multiphase]

    def timesteps(self, index, num_train_timesteps=1000):
        """(timesteps float32, timesteps_prev float64) of :1291-1300."""
        return self.sigmas[index] * num_train_timesteps, self.sigmas_prev[index] * num_train_timesteps

    def add_noise(self, model_input, noise, index):
        """:1301  noisy = sigma * noise + (1 - sigma) * x"""
        B = model_input.shape[0]
        x, nz = model_input.float().contiguous(), noise.float().contiguous()
        out = torch.empty_like(x)
        capi.lib().call("pcm_fm_add_noise", ptr(x), ptr(nz), ptr(self.sigmas), ptr(index), ptr(out), B, x.numel() // B, _stream())
        return out

    def euler_step(self, sample, cond, index, uncond=None, w=3.0):
        """:1334-1357 -- fixed-w CFG of the teacher outputs + one Euler step; returns (x_prev float64, float32 copy)."""
        B = sample.shape[0]
        s, c = sample.float().contiguous(), cond.float().contiguous()
        u = uncond.float().contiguous() if uncond is not None else None
        xp = torch.empty(s.shape, dtype=torch.float64, device=s.device)
        xp32 = torch.empty_like(s)
        capi.lib().call("pcm_fm_cfg_euler_step", ptr(c), ptr(u), ptr(s), ptr(index), float(w), ptr(self.sigmas), ptr(self.sigmas_prev),
                        ptr(xp), ptr(xp32), B, s.numel() // B, _stream())
        return xp, xp32

    def euler_style_multiphase_pred(self, sample, model_pred, timestep_index, multiphase, is_target=False, with_f32=False):
        """:192-230 -> (x float64, timestep_index_end); ``sample`` float32 (noisy input) or float64 (x_prev).
        ``with_f32``: also return the float32 copy the loss consumes (``.float()``, :1376) -> (x64, end, x32)."""
        B = sample.shape[0]
        s = sample.contiguous()
        assert s.dtype in (torch.float32, torch.float64)
        p = model_pred.float().contiguous()
        e = self.edges(multiphase)
        out = torch.empty(s.shape, dtype=torch.float64, device=s.device)
        end = torch.empty(B, dtype=torch.int64, device=s.device)
        out32 = torch.empty(s.shape, dtype=torch.float32, device=s.device) if with_f32 else None
        capi.lib().call("pcm_fm_phase_jump", ptr(s), 1 if s.dtype == torch.float64 else 0, ptr(p), ptr(timestep_index), ptr(self.sigmas),
                        ptr(self.sigmas_prev), ptr(e), int(e.numel()), 1 if is_target else 0, ptr(out), ptr(out32), ptr(end), B,
                        s.numel() // B, _stream())
        return (out, end, out32) if with_f32 else (out, end)


def _noise_travel(solver, x64, noise64, end_index, adv_index):
    B = x64.shape[0]
    x, nz = x64.double().contiguous(), noise64.double().contiguous()
    out = torch.empty_like(x)
    out32 = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    ratio = torch.empty(B, dtype=torch.float32, device=x.device)
    capi.lib().call("pcm_fm_noise_travel", ptr(x), ptr(nz), ptr(solver.sigmas_prev), ptr(end_index), ptr(adv_index), ptr(out), ptr(out32),
                    ptr(ratio), B, x.numel() // B, _stream())
    return out, out32, ratio


EulerSolver.noise_travel = _noise_travel
EulerSolver.noise_travel.__doc__ = """train_pcm_lora_sd3_adv.py:1436-1445 -> (x_adv float64, float32 copy, ratio[B] = d x_adv / d x)."""


class PCMFMSampler:
    """PCMFMDeterministicScheduler / PCMFMStochasticScheduler (pcm_fm_*_scheduler.py:35-242): same constructor arguments,
    ``set_timesteps`` and ``step``; the scheduler's sigma tables stay on the host (a handful of scalars), the update runs in
    ``pcm_fm_sampler_step``."""

    def __init__(self, num_train_timesteps=1000, shift=1.0, pcm_timesteps=50, stochastic=False):
        self.num_train_timesteps, self.shift, self.pcm_timesteps, self.stochastic = num_train_timesteps, shift, pcm_timesteps, stochastic
        full = flow_sigmas(num_train_timesteps, shift)
        et = (np.arange(1, pcm_timesteps + 1) * (num_train_timesteps // pcm_timesteps)).round().astype(np.int64) - 1
        self.sigmas = torch.from_numpy(full[et][::-1].copy())           # descending, float32 (:53-57)
        self.timesteps = self.sigmas * num_train_timesteps
        self._step_index = None

    def set_timesteps(self, num_inference_steps, device=None):
        idx = torch.from_numpy(np.floor(np.linspace(0, self.pcm_timesteps, num=num_inference_steps, endpoint=False)).astype(np.int64))
        s = self.sigmas[idx]
        self.timesteps = (s * self.num_train_timesteps).to(device) if device is not None else s * self.num_train_timesteps
        self.sigmas_ = torch.cat([s, torch.zeros(1)])
        self.num_inference_steps = num_inference_steps
        self._step_index = 0

    @property
    def step_index(self):
        return self._step_index

    def step(self, model_output, timestep, sample, generator=None, noise=None, model_output_uncond=None, guidance_scale=1.0):
        """one sampler update; ``timestep`` is accepted for interface parity (the scheduler walks its own step index, like the
        reference after the first call).  Stochastic: ``noise`` (or a draw from ``generator``) stands for the reference's
        ``torch.randn_like(denoised)``."""
        if self._step_index is None:
            raise RuntimeError("PCMFMSampler.step: call set_timesteps first")
        if isinstance(timestep, int) or (isinstance(timestep, torch.Tensor) and not timestep.is_floating_point()):
            raise ValueError("PCMFMSampler.step: pass one of scheduler.timesteps, not an integer index")   # :211-223
        x, v = sample.float().contiguous(), model_output.float().contiguous()
        sigma, sigma_next = float(self.sigmas_[self._step_index]), float(self.sigmas_[self._step_index + 1])
        nz = None
        if self.stochastic:
            nz = noise if noise is not None else torch.randn(x.shape, generator=generator, device=x.device, dtype=torch.float32)
            nz = nz.float().contiguous()
        vu = model_output_uncond.float().contiguous() if model_output_uncond is not None else None   # fused CFG combine
        out = torch.empty_like(x)
        capi.lib().call("pcm_fm_sampler_step", ptr(v), ptr(vu), float(guidance_scale), ptr(x), sigma, sigma_next, ptr(nz), ptr(out), x.numel(), _stream())
        self._step_index += 1
        return out.to(model_output.dtype)

    def __len__(self):
        return self.num_train_timesteps
