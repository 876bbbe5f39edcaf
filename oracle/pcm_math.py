"""Restatement (torch, CPU, fp32) of the reference-OWNED phased-consistency math.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Each function cites the reference
file:line it follows (paths relative to /root/reference/code/text_to_image_sd15/).
Pinned bit-exactly against the reference's own source via oracle/ref_slice.py in
tests/test_oracle_pinning.py and against tests/golden/pcm_math_golden.safetensors.
"""
import math

import numpy as np
import torch


def sd15_alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    """scheduling_ddpm_modified.py:205-207,:220-221 — 'scaled_linear' betas, fp32 cumprod."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                           dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def extract_into_tensor(a, t, x_shape):
    """train_pcm_lora_sd15.py:283-286."""
    b = t.shape[0]
    out = a.gather(-1, t)
    return out.reshape(b, *((1,) * (len(x_shape) - 1)))


def append_dims(x, target_dims):
    """train_pcm_lora_sd15.py:240-247."""
    d = target_dims - x.ndim
    if d < 0:
        raise ValueError(f"input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x[(...,) + (None,) * d]


def scalings_for_boundary_conditions_target(index, selected_indices):
    """train_pcm_lora_sd15.py:250-253."""
    c_skip = torch.isin(index, selected_indices).float()
    return c_skip, 1.0 - c_skip


def scalings_for_boundary_conditions_online(index, selected_indices):
    """train_pcm_lora_sd15.py:256-259."""
    return torch.zeros_like(index).float(), torch.ones_like(index).float()


def predicted_origin(model_output, timesteps, sample, prediction_type, alphas, sigmas):
    """train_pcm_lora_sd15.py:268-280."""
    s = extract_into_tensor(sigmas, timesteps, sample.shape)
    a = extract_into_tensor(alphas, timesteps, sample.shape)
    if prediction_type == "epsilon":
        return (sample - s * model_output) / a
    if prediction_type == "v_prediction":
        return a * sample - s * model_output
    raise ValueError(f"Prediction type {prediction_type} currently not supported.")


def phase_edges(num_ddim, multiphase):
    """train_pcm_lora_sd15.py:1157-1163 and :322-328 — floor(linspace(0, N, M, endpoint=False))."""
    return torch.from_numpy(
        np.floor(np.linspace(0, num_ddim, num=multiphase, endpoint=False)).astype(np.int64)).long()


class DDIMSolver:
    """train_pcm_lora_sd15.py:289-341."""

    def __init__(self, alpha_cumprods, timesteps=1000, ddim_timesteps=50):
        # :291-299
        self.step_ratio = timesteps // ddim_timesteps
        t = (np.arange(1, ddim_timesteps + 1) * self.step_ratio).round().astype(np.int64) - 1
        self.ddim_alpha_cumprods = torch.from_numpy(alpha_cumprods[t])
        self.ddim_timesteps_prev = torch.from_numpy(np.asarray([0] + t[:-1].tolist())).long()
        self.ddim_alpha_cumprods_prev = torch.from_numpy(
            np.asarray([alpha_cumprods[0]] + alpha_cumprods[t[:-1]].tolist()))
        self.ddim_timesteps = torch.from_numpy(t).long()

    def ddim_step(self, pred_x0, pred_noise, timestep_index):
        # :313-319
        acp = extract_into_tensor(self.ddim_alpha_cumprods_prev, timestep_index, pred_x0.shape)
        dir_xt = (1.0 - acp).sqrt() * pred_noise
        return acp.sqrt() * pred_x0 + dir_xt

    def ddim_style_multiphase_pred(self, pred_x0, pred_noise, timestep_index, multiphase):
        # :321-341 — jump to the largest phase edge <= index
        edges = phase_edges(len(self.ddim_timesteps), multiphase)
        exp_idx = timestep_index.unsqueeze(1).expand(-1, edges.size(0))
        mask = exp_idx >= edges
        last = mask.flip(dims=[1]).long().argmax(dim=1)
        last = edges.size(0) - 1 - last
        e = edges[last]
        acp = extract_into_tensor(self.ddim_alpha_cumprods_prev, e, pred_x0.shape)
        dir_xt = (1.0 - acp).sqrt() * pred_noise
        return acp.sqrt() * pred_x0 + dir_xt, self.ddim_timesteps_prev[e]


def add_noise(alphas_cumprod, original_samples, noise, timesteps):
    """scheduling_ddpm_modified.py:500-524 — alphas_cumprod cast to the sample dtype first (:510)."""
    acp = alphas_cumprod.to(dtype=original_samples.dtype)
    sa = acp[timesteps] ** 0.5
    sb = (1 - acp[timesteps]) ** 0.5
    while sa.ndim < original_samples.ndim:
        sa = sa.unsqueeze(-1)
        sb = sb.unsqueeze(-1)
    return sa * original_samples + sb * noise


def noise_travel(alphas_cumprod, current_samples, noise, current_timesteps, target_timesteps):
    """scheduling_ddpm_modified.py:526-554 — forward-diffuse from t_cur to t_tgt (no guard)."""
    acp = alphas_cumprod.to(dtype=current_samples.dtype)
    r = acp[target_timesteps].flatten() / acp[current_timesteps].flatten()
    sa = r ** 0.5
    sb = (1 - r) ** 0.5
    while sa.ndim < current_samples.ndim:
        sa = sa.unsqueeze(-1)
        sb = sb.unsqueeze(-1)
    return sa * current_samples + sb * noise


def consistency_loss(model_pred, target, loss_type="huber", huber_c=0.001):
    """train_pcm_lora_sd15.py:1283-1293."""
    if loss_type == "l2":
        return torch.nn.functional.mse_loss(model_pred.float(), target.float(), reduction="mean")
    if loss_type == "huber":
        return torch.mean(torch.sqrt((model_pred.float() - target.float()) ** 2 + huber_c ** 2) - huber_c)
    raise ValueError(loss_type)


@torch.no_grad()
def update_ema(target_params, source_params, rate=0.99):
    """train_pcm_lora_sd15.py:344-355 (defined by the reference, never called)."""
    for targ, src in zip(target_params, source_params):
        targ.detach().mul_(rate).add_(src, alpha=1 - rate)


def hinge_d_loss(fake_outputs, real_outputs, weight=1.0):
    """discriminator_sd15.py:412-425 — mean over heads of mean relu(f+1)+mean relu(1-r)."""
    loss = 0.0
    for f, r in zip(fake_outputs, real_outputs):
        loss = loss + (torch.mean(weight * torch.relu(f.float() + 1))
                       + torch.mean(weight * torch.relu(1 - r.float()))) / len(fake_outputs)
    return loss


def hinge_g_loss(fake_outputs, weight=1.0):
    """discriminator_sd15.py:427-434."""
    loss = 0.0
    for f in fake_outputs:
        loss = loss + torch.mean(weight * torch.relu(1 - f.float())) / len(fake_outputs)
    return loss


def kohya_key(peft_key, prefix="lora_unet"):
    """train_pcm_lora_sd15.py:59-62 key renaming."""
    k = peft_key.replace("base_model.model", prefix).replace("lora_A", "lora_down").replace("lora_B", "lora_up")
    return k.replace(".", "_", k.count(".") - 2)


def discriminator_head(sd, x, prefix=""):
    """discriminator_sd15.py:348-368 — conv3x3→GN(32)→LeakyReLU(0.01), conv3x3→GN→LeakyReLU
    (+ skip), conv1x1→1; ``sd`` uses the nn.Sequential key names (conv1.0/conv1.1/conv2.0/...).
    discriminator_sdxl.py:348-369 is the same head with 1x1 convs ("to save memory"): the kernel size is read off the weight."""
    F = torch.nn.functional

    def blk(name, h):
        w = sd[prefix + name + ".0.weight"]
        h = F.conv2d(h, w, sd[prefix + name + ".0.bias"], padding=w.shape[-1] // 2)
        h = F.group_norm(h, 32, sd[prefix + name + ".1.weight"], sd[prefix + name + ".1.bias"], 1e-5)
        return F.leaky_relu(h, 0.01)

    h = blk("conv1", x)
    h = blk("conv2", h) + h
    return F.conv2d(h, sd[prefix + "conv_out.weight"], sd[prefix + "conv_out.bias"])


# ----------------------------------------------------------------------------------------------
# Inference sampler.  diffusers is not vendored under /root/reference, so this is a restatement of the pinned
# diffusers 0.26.3 DDIMScheduler (environment.yaml:40) for the configuration log_validation builds
# (train_pcm_lora_sd15.py:126-135): trailing spacing, clip_sample=False, set_alpha_to_one=False, eta=0, epsilon prediction.
# PARITY UNPINNED: the reference holds no test vector for it; it is anchored on the shared alphas_cumprod table (pinned) only.
# ----------------------------------------------------------------------------------------------
def ddim_trailing_timesteps(num_inference_steps, num_train_timesteps=1000):
    import numpy as np
    return (np.round(np.arange(num_train_timesteps, 0, -num_train_timesteps / num_inference_steps)) - 1).astype(np.int64).tolist()


def ddim_sampler_step(eps, t, sample, acp, num_inference_steps, num_train_timesteps=1000):
    prev_t = t - num_train_timesteps // num_inference_steps
    a_t = acp[t]
    a_prev = acp[prev_t] if prev_t >= 0 else acp[0]
    x0 = (sample - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
    return a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps


def ddim_sample(unet_fn, prompt_embeds, uncond_embeds, latents, num_inference_steps, guidance_scale, acp):
    """StableDiffusionPipeline denoising loop: unet_fn(x, t[B], ctx) -> eps; CFG when guidance_scale > 1."""
    import torch
    x = latents.clone()
    B = x.shape[0]
    for t in ddim_trailing_timesteps(num_inference_steps):
        tt = torch.full((B,), t, dtype=torch.int64)
        eps = unet_fn(x, tt, prompt_embeds)
        if guidance_scale > 1.0:
            eps_u = unet_fn(x, tt, uncond_embeds)
            eps = eps_u + guidance_scale * (eps - eps_u)
        x = ddim_sampler_step(eps, t, x, acp, num_inference_steps)
    return x
