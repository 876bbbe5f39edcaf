#!/usr/bin/env python
"""Few-step latent sampling with an SDXL PCM-LoRA student on MI355X: the denoising loop of the SDXL script's ``log_validation``
(code/text_to_image_sdxl/train_pcm_lora_sdxl_adv.py:160-222: StableDiffusionXLPipeline, DDIM with trailing spacing,
clip_sample=False, set_alpha_to_one=False, optional classifier-free guidance) between the text encoders and the VAE decoder.

    python sample_pcm_lora_sdxl.py --pretrained_teacher_model $SDXL_DIR --lora_dir out/ --prompt_embeds pe.safetensors \\
        --num_inference_steps 4 --guidance_scale 1.0 --output latents.safetensors

``--prompt_embeds``: safetensors with ``prompt_embeds`` [B,77,2048], ``pooled_prompt_embeds`` [B,1280] (+ ``uncond_prompt_embeds`` /
``uncond_pooled_prompt_embeds`` for guidance > 1; SDXL-base uses zeros for the empty negative prompt); ``--synthetic_prompts B``
draws random embeddings instead.  Time ids = (original size, crop top-left, target size) = (R, R, 0, 0, R, R)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    p.add_argument("--pretrained_teacher_model", required=True, help="diffusers SDXL directory, or 'random'")
    p.add_argument("--tiny_model", action="store_true", help="(random weights) a narrow UNet of the same topology for smoke tests of the CLI itself")
    p.add_argument("--lora_dir", default=None, help="trainer output directory (peft adapter_model.safetensors)")
    p.add_argument("--lora_file", default=None, help="a LoRA safetensors file: kohya-ss (lora_unet_*), peft or diffusers (unet.*) keys; rank / alpha read from it")
    p.add_argument("--lora_rank", type=int, default=64)
    p.add_argument("--prompt_embeds", default=None)
    p.add_argument("--synthetic_prompts", type=int, default=0)
    p.add_argument("--num_inference_steps", type=int, default=4)
    p.add_argument("--guidance_scale", type=float, default=1.0)
    p.add_argument("--resolution", type=int, default=1024)
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
    cfg = tr.unet_config(args, "sdxl")
    sd = random_state_dict(cfg, 0, dev) if args.pretrained_teacher_model == "random" else ck.load_unet_state_dict(args.pretrained_teacher_model)
    W = UNetWeights(cfg, sd, dev, need_bwd=False)
    del sd
    if args.lora_file:
        lora = ck.unet_lora_from_file(cfg, args.lora_file, dev)
    else:
        lora = LoraState(cfg, args.lora_rank, 8.0, dev, seed=args.seed)
        if args.lora_dir:
            ck.load_lora(lora, args.lora_dir)
    g = torch.Generator(device=dev).manual_seed(args.seed)
    R = args.resolution
    if args.prompt_embeds:
        t = load_file(args.prompt_embeds)
        pe, pp = t["prompt_embeds"].to(dev, torch.float32), t["pooled_prompt_embeds"].to(dev, torch.float32)
        B = pe.shape[0]
        un = t.get("uncond_prompt_embeds")
        un = un.to(dev, torch.float32).expand(B, -1, -1).contiguous() if un is not None else torch.zeros_like(pe)
        unp = t.get("uncond_pooled_prompt_embeds")
        unp = unp.to(dev, torch.float32).expand(B, -1).contiguous() if unp is not None else torch.zeros_like(pp)
    else:
        B = max(1, args.synthetic_prompts)
        pe, pp = torch.randn(B, 77, cfg.cross_attention_dim, generator=g, device=dev), torch.randn(B, cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim, generator=g, device=dev)
        un, unp = torch.zeros_like(pe), torch.zeros_like(pp)           # force_zeros_for_empty_prompt
    tids = torch.tensor([[R, R, 0, 0, R, R]] * B, device=dev)
    hw = R // 8
    lat = DDIMTrailingSampler(UNet(W, lora)).sample(pe, un, args.num_inference_steps, args.guidance_scale, generator=g, height=hw, width=hw,
                                                    added_cond=dict(text_embeds=pp, time_ids=tids), uncond_added_cond=dict(text_embeds=unp, time_ids=tids))
    save_file({"latents": lat.cpu().contiguous()}, args.output)
    print("wrote %s: latents %s (scale by 1/0.13025 before the SDXL VAE decoder)" % (args.output, tuple(lat.shape)))


if __name__ == "__main__":
    main(parse_args())
