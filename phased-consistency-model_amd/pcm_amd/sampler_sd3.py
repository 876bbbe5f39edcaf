"""Latent-space inference sampler of the SD3 PCM-LoRA student.

What the reference's validation runs between the text encoders and the VAE decoder (train_pcm_lora_sd3.py:1433-1470):
``StableDiffusion3Pipeline``'s denoising loop with ``scheduler=PCMFMDeterministicScheduler(1000, 3.0, 100)`` (:1453; the stochastic
trainer pairs it with PCMFMStochasticScheduler) -- per step: transformer on the latents at ``t`` (classifier-free guidance as one
[negative; positive] 2B pass when guidance_scale > 1), then ``scheduler.step``.  Prompt / pooled embeddings come in, latents go out.
"""
import torch

from .fm import PCMFMSampler


class PCMFMLatentSampler:
    def __init__(self, mmdit, num_train_timesteps=1000, shift=3.0, pcm_timesteps=100, stochastic=False):
        self.mmdit = mmdit
        self.args = (num_train_timesteps, shift, pcm_timesteps, stochastic)

    @torch.no_grad()
    def sample(self, prompt_embeds, pooled, uncond_embeds=None, uncond_pooled=None, num_inference_steps=4, guidance_scale=1.0, latents=None,
               generator=None, height=128, width=128):
        B, dev = prompt_embeds.shape[0], prompt_embeds.device
        sch = PCMFMSampler(*self.args)
        sch.set_timesteps(num_inference_steps, device=dev)
        if latents is None:
            latents = torch.randn(B, self.mmdit.cfg.in_channels, height, width, generator=generator, device=dev, dtype=torch.float32)
        x = latents.to(torch.float32).contiguous()
        cfg = guidance_scale > 1.0 and uncond_embeds is not None
        for t in sch.timesteps:
            tt = t.expand(B).contiguous()
            if cfg:
                both = self.mmdit.forward(torch.cat([x, x]), torch.cat([tt, tt]), torch.cat([uncond_embeds, prompt_embeds]),
                                          torch.cat([uncond_pooled, pooled]))
                x = sch.step(both[B:].contiguous(), t, x, generator=generator, model_output_uncond=both[:B].contiguous(), guidance_scale=guidance_scale)
            else:
                x = sch.step(self.mmdit.forward(x, tt, prompt_embeds, pooled), t, x, generator=generator)
        return x
