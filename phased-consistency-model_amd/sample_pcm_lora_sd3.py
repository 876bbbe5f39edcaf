#!/usr/bin/env python
"""Few-step latent sampling with an SD3 PCM-LoRA student on MI355X: the denoising loop of the reference's validation
(train_pcm_lora_sd3.py:1433-1470: StableDiffusion3Pipeline with PCMFMDeterministicScheduler(1000, 3.0, 100); ``--stochastic`` swaps
in PCMFMStochasticScheduler, pcm_fm_stochastic_scheduler.py).

    python sample_pcm_lora_sd3.py --pretrained_teacher_model $SD3_DIR --lora_dir out/ --prompt_embeds pe.safetensors \\
        --num_inference_steps 4 --guidance_scale 1.2 --output latents.safetensors

``--prompt_embeds``: safetensors with ``prompt_embeds`` [B,154,4096], ``pooled_prompt_embeds`` [B,2048] (+ ``uncond_prompt_embeds``,
``uncond_pooled_prompt_embeds`` for guidance > 1); the three text encoders and the VAE are outside this repo's scope,
``--synthetic_prompts B`` draws random embeddings instead.  ``--lora_dir`` takes what the trainer writes."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    p.add_argument("--pretrained_teacher_model", required=True, help="diffusers SD3 directory, or 'random'")
    p.add_argument("--num_layers", type=int, default=None, help="(random weights) MMDiT depth, default 24")
    p.add_argument("--tiny_model", action="store_true", help="(random weights) a 128-wide MMDiT for smoke tests of the CLI itself")
    p.add_argument("--lora_dir", default=None, help="trainer output directory (reads its pytorch_lora_weights.safetensors)")
    p.add_argument("--lora_file", default=None, help="a LoRA safetensors file (diffusers 'transformer.*' or peft keys); module set and rank are read from it")
    p.add_argument("--lora_scale", type=float, default=1.0, help="sd3_test.py's alpha: LoRA tensors are multiplied by sqrt(alpha)")
    p.add_argument("--lora_alpha", type=float, default=8.0,
                   help="LoRA alpha the file was trained with (scaling = alpha / rank).  8 = this repo's and the reference trainers' raw output "
                        "(lora_alpha=8 at train_pcm_lora_sd3.py:973-978).  Files processed by the reference's convert.py (A/2, B/2, meant to be "
                        "loaded by diffusers with alpha = rank) need --lora_alpha <rank>, e.g. 32")
    p.add_argument("--lora_rank", type=int, default=32)
    p.add_argument("--prompt_embeds", default=None)
    p.add_argument("--synthetic_prompts", type=int, default=0)
    p.add_argument("--num_inference_steps", type=int, default=4)
    p.add_argument("--guidance_scale", type=float, default=1.0)
    p.add_argument("--pcm_timesteps", type=int, default=100)
    p.add_argument("--shift", type=float, default=3.0)
    p.add_argument("--stochastic", action="store_true")
    p.add_argument("--resolution", type=int, default=1024)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--output", default="latents.safetensors")
    return p.parse_args(argv)


def main(args):
    from safetensors.torch import load_file, save_file

    from pcm_amd import capi, checkpoint as ck
    from pcm_amd.mmdit import MMDiT, MMDiTWeights, sd3_lora_state
    from pcm_amd.mmdit_spec import MMDiTConfig, random_state_dict
    from pcm_amd.sampler_sd3 import PCMFMLatentSampler
    import train_pcm_lora_sd3 as tr
    capi.lib()
    dev = tr.pick_device(0)
    cfg = tr.model_config(args)
    sd = random_state_dict(cfg, 0, dev) if args.pretrained_teacher_model == "random" else ck.load_transformer_state_dict(args.pretrained_teacher_model)
    W = MMDiTWeights(cfg, sd, dev, need_bwd=False)
    del sd
    lora_file = args.lora_file or (os.path.join(args.lora_dir, "pytorch_lora_weights.safetensors") if args.lora_dir else None)
    if lora_file:
        lora = ck.sd3_lora_from_file(cfg, lora_file, dev, lora_alpha=args.lora_alpha, scale=args.lora_scale)
    else:
        lora = sd3_lora_state(cfg, args.lora_rank, 8.0, dev, seed=args.seed)          # B = 0: the teacher
    g = torch.Generator(device=dev).manual_seed(args.seed)
    un = unp = None
    if args.prompt_embeds:
        t = load_file(args.prompt_embeds)
        pe, pp = t["prompt_embeds"].to(dev, torch.float32), t["pooled_prompt_embeds"].to(dev, torch.float32)
        if "uncond_prompt_embeds" in t:
            un = t["uncond_prompt_embeds"].to(dev, torch.float32).expand(pe.shape[0], -1, -1).contiguous()
            unp = t["uncond_pooled_prompt_embeds"].to(dev, torch.float32).expand(pe.shape[0], -1).contiguous()
    else:
        B = max(1, args.synthetic_prompts)
        pe, pp = torch.randn(B, 154, cfg.joint_attention_dim, generator=g, device=dev), torch.randn(B, cfg.pooled_projection_dim, generator=g, device=dev)
        un, unp = torch.randn(B, 154, cfg.joint_attention_dim, generator=g, device=dev), torch.randn(B, cfg.pooled_projection_dim, generator=g, device=dev)
    hw = args.resolution // 8
    smp = PCMFMLatentSampler(MMDiT(W, lora), shift=args.shift, pcm_timesteps=args.pcm_timesteps, stochastic=args.stochastic)
    lat = smp.sample(pe, pp, un, unp, args.num_inference_steps, args.guidance_scale, generator=g, height=hw, width=hw)
    save_file({"latents": lat.cpu().contiguous()}, args.output)
    print("wrote %s: latents %s (apply latents / scaling_factor + shift_factor before the VAE decoder)" % (args.output, tuple(lat.shape)))


if __name__ == "__main__":
    main(parse_args())
