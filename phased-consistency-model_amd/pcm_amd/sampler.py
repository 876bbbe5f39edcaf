"""Latent-space inference sampler of the PCM-LoRA student.

Mirrors what ``log_validation`` (train_pcm_lora_sd15.py:120-207) runs between the text encoder and the VAE decoder:
``StableDiffusionPipeline.__call__``'s denoising loop with ``DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085,
beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="trailing", clip_sample=False, set_alpha_to_one=False)`` (:126-135)
and classifier-free guidance (``guidance_scale`` 1 and 7.5 at :1345-1365).  Text / VAE encoders are out of scope:
prompt embeddings come in, latents go out.  The reference fuses the LoRA into the weights (``pipeline.fuse_lora()``, :145); here
the low-rank branch stays the second K segment of the same contractions, which is the same function up to bf16 rounding.
"""
import torch

from . import ops
from .trainer import scaled_linear_alphas_cumprod


def trailing_timesteps(num_inference_steps, num_train_timesteps=1000):
    """DDIMScheduler.set_timesteps, timestep_spacing="trailing": round(arange(T, 0, -T/N)) - 1."""
    step = num_train_timesteps / num_inference_steps
    ts, t = [], float(num_train_timesteps)
    while t > 1e-9 and len(ts) < num_inference_steps:
        ts.append(int(round(t)) - 1)
        t -= step
    return ts


class DDIMTrailingSampler:
    """``sample(prompt_embeds, uncond_embeds, num_inference_steps, guidance_scale, latents)`` -> final latents [B,4,H,W] fp32."""

    def __init__(self, unet, num_train_timesteps=1000):
        self.unet = unet
        self.T = num_train_timesteps
        acp = scaled_linear_alphas_cumprod(num_train_timesteps, 0.00085, 0.012)   # fp32 table, the training schedule (:126-130 = :805-807)
        self.acp = [float(a) for a in acp]
        self.final_alpha = self.acp[0]                                  # set_alpha_to_one=False

    @torch.no_grad()
    def sample(self, prompt_embeds, uncond_embeds=None, num_inference_steps=4, guidance_scale=1.0, latents=None, generator=None,
               height=64, width=64, added_cond=None, uncond_added_cond=None):
        """``added_cond`` / ``uncond_added_cond``: SDXL ``added_cond_kwargs`` ({'text_embeds': [B,1280], 'time_ids': [B,6]}) for the
        positive / negative branch (StableDiffusionXLPipeline, the SDXL script's log_validation, train_pcm_lora_sdxl_adv.py:160-222)."""
        B, dev = prompt_embeds.shape[0], prompt_embeds.device
        if latents is None:
            latents = torch.randn(B, 4, height, width, generator=generator, device=dev, dtype=torch.float32)   # init_noise_sigma = 1
        x = latents.to(torch.float32).contiguous()
        cfg = guidance_scale > 1.0 and uncond_embeds is not None
        skip = self.T // num_inference_steps
        for t in trailing_timesteps(num_inference_steps, self.T):
            tt = torch.full((B,), t, dtype=torch.int64, device=dev)
            if cfg:   # [uncond; cond] as one 2B forward (the pipeline concatenates them the same way)
                ac2 = None
                if added_cond is not None:
                    un = uncond_added_cond if uncond_added_cond is not None else added_cond
                    ac2 = {k: torch.cat([un[k], added_cond[k]]) for k in added_cond}
                both = self.unet.forward(torch.cat([x, x]), torch.cat([tt, tt]), torch.cat([uncond_embeds, prompt_embeds]), added_cond=ac2)
                eps_u, eps_c = both[:B].contiguous(), both[B:].contiguous()
            else:
                eps_u, eps_c = None, self.unet.forward(x, tt, prompt_embeds, added_cond=added_cond)
            prev = t - skip
            a_prev = self.acp[prev] if prev >= 0 else self.final_alpha
            x = ops.sampler_ddim_step(eps_c, eps_u, x, self.acp[t], a_prev, guidance_scale)
        return x
