#!/usr/bin/env python
"""train_pcm_lora_sd3.py — PCM-LoRA distillation of the SD3 transformer (MMDiT) on MI355X (BASELINE.json configs[4]).

Takes the launch line of code/text_to_image_sd3/train_pcm_lora_sd3.py (run.sh: --lora_rank=32 --learning_rate=5e-6
--adam_weight_decay=1e-3 --num_euler_timesteps=100 --multiphase=N ...): the reference's flags with the reference's defaults;
flags that only drive out-of-scope subsystems (VAE / three text encoders, validation pipeline, hub, prodigy, xformers ...) are
accepted and ignored.  One process per GPU (torch.distributed.run), the LoRA gradients are all-reduced over RCCL/xGMI.

The step is ``pcm_amd.trainer_sd3.SD3Distiller`` (train_pcm_lora_sd3.py:1270-1390): flow-matching noising at a random Euler index,
online MMDiT prediction, frozen teacher with the reference's fixed w = 3 CFG + one Euler step, target prediction with the online LoRA
weights under no-grad, the two multiphase jumps, huber loss (the reference ignores --loss_type here), clip + AdamW.

Data: ``--latents_dir`` safetensors shards with ``latents`` [N,16,h,w] (already scaled by the VAE scaling factor, :1277-1278),
``prompt_embeds`` [N,154,4096], ``pooled_prompt_embeds`` [N,2048] and, in any shard, ``uncond_prompt_embeds`` [154,4096] /
``uncond_pooled_prompt_embeds`` [2048] (the encodings of "", :1260-1262); or ``--synthetic_data``.
``--pretrained_teacher_model`` is a diffusers SD3 directory (transformer/diffusion_pytorch_model*.safetensors) or ``random``.
Final checkpoint: ``pytorch_lora_weights.safetensors`` in StableDiffusion3Pipeline.save_lora_weights layout (:1495-1500).
"""
import argparse
import glob
import json
import logging
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import train_pcm_lora_sd15 as base  # noqa: E402  (lr schedules)

logger = logging.getLogger("pcm_amd")


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="PCM-LoRA distillation (SD3 / MMDiT) on MI355X")
    # ---- reference flags, reference defaults (train_pcm_lora_sd3.py:247-640) ----
    p.add_argument("--pretrained_teacher_model", type=str, default=None, required=True)
    p.add_argument("--revision", type=str, default=None)
    p.add_argument("--variant", type=str, default=None)
    p.add_argument("--num_validation_images", type=int, default=4)
    p.add_argument("--validation_steps", type=int, default=50)
    p.add_argument("--lora_rank", type=int, default=4)
    p.add_argument("--enable_xformers_memory_efficient_attention", action="store_true")
    p.add_argument("--output_dir", type=str, default="sd3-dreambooth")
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--resolution", type=int, default=1024)
    p.add_argument("--center_crop", default=False, action="store_true")
    p.add_argument("--random_flip", action="store_true")
    p.add_argument("--train_batch_size", type=int, default=4)
    p.add_argument("--sample_batch_size", type=int, default=4)
    p.add_argument("--num_train_epochs", type=int, default=1)
    p.add_argument("--max_train_steps", type=int, default=None)
    p.add_argument("--checkpointing_steps", type=int, default=500)
    p.add_argument("--checkpoints_total_limit", type=int, default=None)
    p.add_argument("--resume_from_checkpoint", type=str, default=None)
    p.add_argument("--gradient_accumulation_steps", type=int, default=1)
    p.add_argument("--gradient_checkpointing", action="store_true")
    p.add_argument("--learning_rate", type=float, default=1e-4)
    p.add_argument("--text_encoder_lr", type=float, default=5e-6)
    p.add_argument("--scale_lr", action="store_true", default=False)
    p.add_argument("--lr_scheduler", type=str, default="constant")
    p.add_argument("--lr_warmup_steps", type=int, default=500)
    p.add_argument("--lr_num_cycles", type=int, default=1)
    p.add_argument("--lr_power", type=float, default=1.0)
    p.add_argument("--dataloader_num_workers", type=int, default=0)
    p.add_argument("--weighting_scheme", type=str, default="sigma_sqrt")
    p.add_argument("--logit_mean", type=float, default=0.0)
    p.add_argument("--logit_std", type=float, default=1.0)
    p.add_argument("--mode_scale", type=float, default=1.29)
    p.add_argument("--optimizer", type=str, default="AdamW")
    p.add_argument("--w_min", type=float, default=5.0)
    p.add_argument("--w_max", type=float, default=15.0)
    p.add_argument("--use_8bit_adam", action="store_true")
    p.add_argument("--loss_type", type=str, default="l2")
    p.add_argument("--huber_c", type=float, default=0.001)
    p.add_argument("--adam_beta1", type=float, default=0.9)
    p.add_argument("--adam_beta2", type=float, default=0.999)
    p.add_argument("--prodigy_beta3", type=float, default=None)
    p.add_argument("--prodigy_decouple", type=bool, default=True)
    p.add_argument("--adam_weight_decay", type=float, default=1e-04)
    p.add_argument("--adam_weight_decay_text_encoder", type=float, default=1e-03)
    p.add_argument("--adam_epsilon", type=float, default=1e-08)
    p.add_argument("--prodigy_use_bias_correction", type=bool, default=True)
    p.add_argument("--prodigy_safeguard_warmup", type=bool, default=True)
    p.add_argument("--max_grad_norm", default=1.0, type=float)
    p.add_argument("--push_to_hub", action="store_true")
    p.add_argument("--hub_token", type=str, default=None)
    p.add_argument("--hub_model_id", type=str, default=None)
    p.add_argument("--logging_dir", type=str, default="logs")
    p.add_argument("--allow_tf32", action="store_true")
    p.add_argument("--report_to", type=str, default="tensorboard")
    p.add_argument("--mixed_precision", type=str, default=None, choices=["no", "fp16", "bf16"])
    p.add_argument("--teacher_precision", type=str, default="reference", choices=["reference", "same", "fp16"], help=base.TEACHER_PRECISION_HELP)
    p.add_argument("--prior_generation_precision", type=str, default=None)
    p.add_argument("--local_rank", type=int, default=-1)
    p.add_argument("--num_euler_timesteps", type=int, default=50)
    p.add_argument("--tracker_project_name", type=str, default="text2image-fine-tune")
    p.add_argument("--not_apply_cfg_solver", action="store_true")
    p.add_argument("--multiphase", default=8, type=int)
    # ---- additions of this build ----
    p.add_argument("--latents_dir", type=str, default=None, help="safetensors shards of precomputed latents / text embeddings")
    p.add_argument("--synthetic_data", action="store_true", help="seeded N(0,1) latents / embeddings")
    p.add_argument("--num_layers", type=int, default=None, help="(with --pretrained_teacher_model random) MMDiT depth, default 24")
    p.add_argument("--tiny_model", action="store_true", help="(with random weights) a 128-wide MMDiT for smoke tests of the CLI itself")
    args = p.parse_args(argv)
    env_local_rank = int(os.environ.get("LOCAL_RANK", -1))
    if env_local_rank != -1 and env_local_rank != args.local_rank:
        args.local_rank = env_local_rank
    return args


class SD3Source:
    """Per-rank batches: latents [16,h,w], prompt embeds [154,4096], pooled [2048] (stands in for the dataset + VAE + 3 text encoders)."""

    def __init__(self, args, rank, world, device, cfg):
        self.bs, self.device = args.train_batch_size, device
        self.g = torch.Generator(device=device).manual_seed((args.seed or 0) + rank)
        self.hw = args.resolution // 8
        self.Lc, self.jd, self.pd, self.cin = 154, cfg.joint_attention_dim, cfg.pooled_projection_dim, cfg.in_channels
        self.shards = False
        un, unp = None, None
        if args.latents_dir:
            from safetensors.torch import load_file
            every = sorted(glob.glob(os.path.join(args.latents_dir, "*.safetensors")))
            files = every[rank::world]
            un = base.find_in_shards(every, "uncond_prompt_embeds", device)          # run-global: may sit in a shard of another rank
            unp = base.find_in_shards(every, "uncond_pooled_prompt_embeds", device)
            if not files:
                raise FileNotFoundError(f"no shards for rank {rank} in {args.latents_dir}")
            data = [load_file(f) for f in files]
            self.lat = torch.cat([d["latents"] for d in data]).float().to(device)
            self.pe = torch.cat([d["prompt_embeds"] for d in data]).float().to(device)
            self.pp = torch.cat([d["pooled_prompt_embeds"] for d in data]).float().to(device)
            self.Lc = self.pe.shape[1]
            self.shards = True
        elif not args.synthetic_data:
            raise SystemExit("pcm_amd: give --latents_dir or --synthetic_data (VAE / text encoding is out of scope, see --help)")
        if self.shards and (un is None or unp is None):
            # the w = 3 CFG teacher step needs the text encoders' output for the empty caption (train_pcm_lora_sd3.py:1216-1226); a random
            # stand-in would distill against a meaningless guidance direction without any sign of it
            raise SystemExit("pcm_amd: --latents_dir shards carry no 'uncond_prompt_embeds' / 'uncond_pooled_prompt_embeds' (the encoding of "
                             "the empty prompt): add them to any one shard")
        if un is None:       # --synthetic_data only
            un = torch.randn(self.Lc, self.jd, generator=self.g, device=device)
        if unp is None:
            unp = torch.randn(self.pd, generator=self.g, device=device)
        self.uncond = un.float().to(device).expand(self.bs, self.Lc, self.jd).contiguous()
        self.uncond_pooled = unp.float().to(device).expand(self.bs, self.pd).contiguous()

    def __len__(self):
        return (self.lat.shape[0] // self.bs) if self.shards else 10 ** 9

    def batch(self):
        if self.shards:
            idx = torch.randint(0, self.lat.shape[0], (self.bs,), generator=self.g, device=self.device)
            return self.lat[idx].contiguous(), self.pe[idx].contiguous(), self.pp[idx].contiguous()
        r = lambda *s: torch.randn(*s, generator=self.g, device=self.device)   # noqa: E731
        return r(self.bs, self.cin, self.hw, self.hw), r(self.bs, self.Lc, self.jd), r(self.bs, self.pd)


def apply_scale_lr(args, world):
    """--scale_lr (train_pcm_lora_sd3.py:1061-1067; the SDXL script has the same block at :1166-1172, the SD1.5 script defines the
    flag but never reads it): lr *= gradient_accumulation_steps * train_batch_size * num_processes."""
    if getattr(args, "scale_lr", False):
        args.learning_rate = args.learning_rate * args.gradient_accumulation_steps * args.train_batch_size * world
    return args.learning_rate


def pick_device(local_rank):
    """cuda:<local_rank>; PCM_CLI_DEVICE=cpu (tests: the CLI end to end on the host emulator, which the test installs as the library)."""
    if os.environ.get("PCM_CLI_DEVICE") == "cpu":
        return torch.device("cpu")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    return dev


def model_config(args):
    from pcm_amd.mmdit_spec import MMDiTConfig
    if args.pretrained_teacher_model != "random":
        return MMDiTConfig.sd3_medium()
    if getattr(args, "tiny_model", False):
        return MMDiTConfig(sample_size=16, num_layers=args.num_layers or 2, num_attention_heads=2, joint_attention_dim=96, caption_projection_dim=128,
                           pooled_projection_dim=64, pos_embed_max_size=16)
    return MMDiTConfig(num_layers=args.num_layers) if args.num_layers else MMDiTConfig.sd3_medium()


def main(args):
    from pcm_amd import capi, checkpoint as ck
    from pcm_amd.mmdit import MMDiTWeights, sd3_lora_state
    from pcm_amd.mmdit_spec import MMDiTConfig, random_state_dict
    from pcm_amd.trainer_sd3 import SD3Distiller, SD3StepConfig
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = max(args.local_rank, 0)
    logging.basicConfig(format="%(asctime)s - %(levelname)s - %(name)s - %(message)s", datefmt="%m/%d/%Y %H:%M:%S",
                        level=logging.INFO if rank == 0 else logging.WARNING)
    if args.gradient_accumulation_steps < 1:
        raise SystemExit("pcm_amd: --gradient_accumulation_steps must be >= 1")
    if args.optimizer.lower() != "adamw":
        raise SystemExit("pcm_amd: only --optimizer AdamW (the reference recipes') is implemented")
    apply_scale_lr(args, world)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world)
    device = pick_device(local_rank)
    base.apply_mixed_precision(args)          # fp16 (every recipe of text_to_image_sd3/run.sh): the half build + device-side GradScaler
    capi.lib()
    if args.seed is not None:
        torch.manual_seed(args.seed + rank)
    mcfg = model_config(args)
    if args.pretrained_teacher_model == "random":
        sd = random_state_dict(mcfg, seed=0, device=device)
    else:
        sd = ck.load_transformer_state_dict(args.pretrained_teacher_model)
    W = MMDiTWeights(mcfg, sd, device)
    Wt = base.teacher_weights_for(args, mcfg, sd, device, MMDiTWeights)
    del sd
    lora = sd3_lora_state(mcfg, args.lora_rank, 8.0, device, seed=(args.seed or 0))
    if world > 1:
        torch.distributed.broadcast(lora.params, src=0)
        lora.repack()
    cfg = SD3StepConfig(num_euler_timesteps=args.num_euler_timesteps, multiphase=args.multiphase, huber_c=args.huber_c,
                        learning_rate=args.learning_rate, adam_beta1=args.adam_beta1, adam_beta2=args.adam_beta2,
                        adam_weight_decay=args.adam_weight_decay, adam_epsilon=args.adam_epsilon, max_grad_norm=args.max_grad_norm,
                        lora_rank=args.lora_rank, not_apply_cfg_solver=args.not_apply_cfg_solver)
    D = SD3Distiller(W, lora, cfg, world_size=world, teacher_weights=Wt)
    src = SD3Source(args, rank, world, device, mcfg)
    if args.max_train_steps is None:
        args.max_train_steps = args.num_train_epochs * base.agreed_steps_per_epoch(len(src), world)
    global_step = 0
    if rank == 0:
        os.makedirs(os.path.join(args.output_dir, args.logging_dir), exist_ok=True)
    if args.resume_from_checkpoint:
        path = os.path.basename(args.resume_from_checkpoint) if args.resume_from_checkpoint != "latest" else ck.latest_checkpoint(args.output_dir)
        if path is None:
            logger.info("Checkpoint '%s' does not exist. Starting a new training run.", args.resume_from_checkpoint)
        else:
            logger.info("Resuming from checkpoint %s", path)
            global_step = ck.load_state(D, os.path.join(args.output_dir, path))
    base.reseed_for_resume(src, args, rank, global_step)
    logf = open(os.path.join(args.output_dir, args.logging_dir, f"{args.tracker_project_name}.jsonl"), "a") if rank == 0 else None
    logger.info("***** Running training *****  world=%d  per-GPU batch=%d  total steps=%d", world, args.train_batch_size, args.max_train_steps)
    t_last = time.time()
    ga = args.gradient_accumulation_steps

    def draw_batch():
        """one (micro-)batch in the reference's draw order, drawn one call ahead of its use (its teacher targets are computed beside the previous
        batch's student work, SD3Distiller.step(prefetch=...)); the sequence of draws is unchanged"""
        latents, pe, pp = src.batch()
        noise = torch.randn(latents.shape, generator=src.g, device=device)                                            # :1281
        index = torch.randint(0, args.num_euler_timesteps, (latents.shape[0],), generator=src.g, device=device)       # :1285-1287
        return (latents, pe, pp, src.uncond, src.uncond_pooled, noise, index)

    prefetch_on = os.environ.get("PCM_TEACHER_PREFETCH", "1") != "0"
    left = (args.max_train_steps - global_step) * ga
    cur = None
    if left > 0:
        cur, left = draw_batch(), left - 1
    while global_step < args.max_train_steps:
        lr = base.lr_at(args, base.sched_pos(D, args, global_step))
        for micro in range(ga):                                # accelerator.accumulate(transformer), :1267-1268
            nxt = None
            if left > 0:
                nxt, left = draw_batch(), left - 1
            out = D.step(*cur, lr=lr, accum=(micro, ga), prefetch=nxt if prefetch_on else None)
            cur = nxt
        global_step += 1
        if rank == 0:
            loss = float(out["loss"].item())
            now = time.time()
            rec = {"step": global_step, "loss": loss, "lr": lr, "grad_norm": D.grad_norm(), "sec": now - t_last}
            t_last = now
            logf.write(json.dumps(rec) + "\n")
            logf.flush()
            if global_step % 10 == 0 or global_step == 1:
                logger.info("step %d loss %.6f lr %.3g grad_norm %.4f (%.3f s/step)", global_step, loss, lr, rec["grad_norm"], rec["sec"])
            if global_step % args.checkpointing_steps == 0:                                                          # :1391-1430
                ck.rotate_checkpoints(args.output_dir, args.checkpoints_total_limit)
                save_path = os.path.join(args.output_dir, f"checkpoint-{global_step}")
                ck.save_state(D, save_path, global_step)
                logger.info("Saved state to %s", save_path)
    if world > 1:
        torch.distributed.barrier()
    if rank == 0:
        ck.save_lora_sd3(lora, args.output_dir)                                                                       # :1495-1500
        logf.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(parse_args())
