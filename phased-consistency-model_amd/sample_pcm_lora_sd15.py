#!/usr/bin/env python
"""Few-step latent sampling with a PCM-LoRA student on MI355X (the denoising loop of ``log_validation``,
train_pcm_lora_sd15.py:120-207: DDIM, trailing spacing, optional classifier-free guidance).

    python sample_pcm_lora_sd15.py --pretrained_teacher_model $MODEL_DIR --lora_dir out/ --prompt_embeds pe.safetensors \\
        --num_inference_steps 4 --guidance_scale 1.0 --output latents.safetensors

``--prompt_embeds``: safetensors with ``prompt_embeds`` [B,77,768] (+ ``uncond_prompt_embeds`` for guidance > 1); text and VAE
encoders are outside this repo's scope, ``--synthetic_prompts B`` draws random embeddings instead.  ``--lora_dir`` takes what the
trainer writes (peft ``adapter_model.safetensors``)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    p.add_argument("--pretrained_teacher_model", required=True, help="diffusers SD1.5 directory, or 'random'")
    p.add_argument("--tiny_model", action="store_true", help="(random weights) a narrow UNet of the same topology for smoke tests of the CLI itself")
    p.add_argument("--lora_dir", default=None, help="trainer output directory (peft adapter_model.safetensors)")
    p.add_argument("--lora_file", default=None, help="a LoRA safetensors file: kohya-ss (lora_unet_*), peft or diffusers (unet.*) keys; rank / alpha read from it")
    p.add_argument("--lora_rank", type=int, default=64)
    p.add_argument("--prompt_embeds", default=None)
    p.add_argument("--synthetic_prompts", type=int, default=0)
    p.add_argument("--num_inference_steps", type=int, default=4)
    p.add_argument("--guidance_scale", type=float, default=1.0)
    p.add_argument("--resolution", type=int, default=512)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--output", default="latents.safetensors")
    return p.parse_args(argv)


def main(args):
    from safetensors.torch import load_file, save_file

    from pcm_amd import capi, checkpoint as ck
    from pcm_amd.model import LoraState, UNet, UNetWeights
    from pcm_amd.sampler import DDIMTrailingSampler
    from pcm_amd.unet_spec import UNetConfig, random_state_dict
    import train_pcm_lora_sd15 as tr
    capi.lib()
    dev = tr.pick_device(0)
    cfg = tr.unet_config(args)
    sd = random_state_dict(cfg, 0, dev) if args.pretrained_teacher_model == "random" else ck.load_unet_state_dict(args.pretrained_teacher_model)
    W = UNetWeights(cfg, sd, dev, need_bwd=False)
    if args.lora_file:
        lora = ck.unet_lora_from_file(cfg, args.lora_file, dev)
    else:
        lora = LoraState(cfg, args.lora_rank, 8.0, dev, seed=args.seed)
        if args.lora_dir:
            ck.load_lora(lora, args.lora_dir)
    g = torch.Generator(device=dev).manual_seed(args.seed)
    if args.prompt_embeds:
        t = load_file(args.prompt_embeds)
        pe = t["prompt_embeds"].to(dev, torch.float32)
        un = t.get("uncond_prompt_embeds")
        un = un.to(dev, torch.float32) if un is not None else None
    else:
        B = max(1, args.synthetic_prompts)
        pe = torch.randn(B, 77, cfg.cross_attention_dim, generator=g, device=dev)
        un = torch.randn(1, 77, cfg.cross_attention_dim, generator=g, device=dev).expand(B, -1, -1).contiguous()
    hw = args.resolution // 8
    lat = DDIMTrailingSampler(UNet(W, lora)).sample(pe, un, args.num_inference_steps, args.guidance_scale, generator=g, height=hw, width=hw)
    save_file({"latents": lat.cpu().contiguous()}, args.output)
    print("wrote %s: latents %s (scale by 1/0.18215 before the VAE decoder)" % (args.output, tuple(lat.shape)))


if __name__ == "__main__":
    main(parse_args())
