#!/usr/bin/env python
"""train_pcm_lora_sd3_adv.py — PCM-LoRA distillation of the SD3 transformer with the latent adversarial consistency loss on MI355X
(BASELINE.json configs[4]: the run.sh recipes, e.g. ``--lora_rank=32 --num_euler_timesteps=100 --multiphase=2
--adv_weight=0.1 --adv_lr=1e-5``).

Takes the launch line of code/text_to_image_sd3/train_pcm_lora_sd3_adv.py: the base SD3 flags + ``--adv_weight`` / ``--adv_lr``
(:640-641).  Differences from the base trainer that are reproduced: the 22-entry LoRA target list with peft's default init
(:987-1016, including the three leading-dot entries that can never match), ``--loss_type`` is honoured on generator steps
(:1468-1481), even global steps update the discriminator heads only, the lr schedule advances on generator steps only.
``train_pcm_lora_sd3_adv_stochastic.py`` (same directory) is this script with ``pos_embed.proj`` dropped from the LoRA list
(:1008 of the stochastic script); its validation sampler is ``sample_pcm_lora_sd3.py --stochastic``.
"""
import json
import logging
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import train_pcm_lora_sd3 as sd3  # noqa: E402
import train_pcm_lora_sd15 as base  # noqa: E402

logger = logging.getLogger("pcm_amd")
STOCHASTIC = False      # set by train_pcm_lora_sd3_adv_stochastic.py


def parse_args(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    extra = {"adv_weight": 0.1, "adv_lr": 1e-5}
    rest, i = [], 0
    while i < len(argv):
        a, hit = argv[i], False
        for k in extra:
            if a == "--" + k:
                extra[k] = argv[i + 1]; i += 2; hit = True
                break
            if a.startswith("--" + k + "="):
                extra[k] = a.split("=", 1)[1]; i += 1; hit = True
                break
        if not hit:
            rest.append(a); i += 1
    args = sd3.parse_args(rest)
    args.adv_weight, args.adv_lr = float(extra["adv_weight"]), float(extra["adv_lr"])
    return args


def lora_targets():
    from pcm_amd.mmdit_spec import LORA_TARGETS_SD3_ADV
    return tuple(t for t in LORA_TARGETS_SD3_ADV if not (STOCHASTIC and t == "pos_embed.proj"))


def main(args):
    from pcm_amd import capi, checkpoint as ck
    from pcm_amd.discriminator import Discriminator
    from pcm_amd.mmdit import MMDiTWeights, sd3_lora_state
    from pcm_amd.mmdit_spec import MMDiTConfig, random_state_dict
    from pcm_amd.trainer_sd3 import SD3AdvDistiller, SD3StepConfig
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = max(args.local_rank, 0)
    logging.basicConfig(format="%(asctime)s - %(levelname)s - %(name)s - %(message)s", datefmt="%m/%d/%Y %H:%M:%S",
                        level=logging.INFO if rank == 0 else logging.WARNING)
    if args.gradient_accumulation_steps != 1:
        raise SystemExit("pcm_amd: --gradient_accumulation_steps != 1 is not implemented (reference recipes use 1)")
    if args.optimizer.lower() != "adamw":
        raise SystemExit("pcm_amd: only --optimizer AdamW (the reference recipes') is implemented")
    sd3.apply_scale_lr(args, world)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world)
    device = sd3.pick_device(local_rank)
    base.apply_mixed_precision(args)          # fp16 (every recipe of text_to_image_sd3/run.sh): the half build + device-side GradScaler
    capi.lib()
    if args.seed is not None:
        torch.manual_seed(args.seed + rank)
    mcfg = sd3.model_config(args)
    if args.pretrained_teacher_model == "random":
        sd = random_state_dict(mcfg, seed=0, device=device)
    else:
        sd = ck.load_transformer_state_dict(args.pretrained_teacher_model)
    W = MMDiTWeights(mcfg, sd, device)
    Wt = base.teacher_weights_for(args, mcfg, sd, device, MMDiTWeights)
    del sd
    lora = sd3_lora_state(mcfg, args.lora_rank, 8.0, device, seed=(args.seed or 0), targets=lora_targets(), init="kaiming")
    disc = Discriminator([mcfg.inner_dim] * mcfg.num_layers, num_h_per_head=1, device=device, seed=(args.seed or 0) + 1, ksize=1)   # discriminator_sd3.py:171-190
    if world > 1:
        torch.distributed.broadcast(lora.params, src=0); lora.repack()
        torch.distributed.broadcast(disc.params, src=0); disc.repack()
    cfg = SD3StepConfig(num_euler_timesteps=args.num_euler_timesteps, multiphase=args.multiphase, huber_c=args.huber_c,
                        learning_rate=args.learning_rate, adam_beta1=args.adam_beta1, adam_beta2=args.adam_beta2,
                        adam_weight_decay=args.adam_weight_decay, adam_epsilon=args.adam_epsilon, max_grad_norm=args.max_grad_norm,
                        lora_rank=args.lora_rank, not_apply_cfg_solver=args.not_apply_cfg_solver)
    D = SD3AdvDistiller(W, lora, cfg, disc, adv_weight=args.adv_weight, adv_lr=args.adv_lr, loss_type=args.loss_type, world_size=world, teacher_weights=Wt)
    src = sd3.SD3Source(args, rank, world, device, mcfg)
    if args.max_train_steps is None:
        args.max_train_steps = args.num_train_epochs * base.agreed_steps_per_epoch(len(src), world)
    global_step, gen_steps = 0, 0
    if rank == 0:
        os.makedirs(os.path.join(args.output_dir, args.logging_dir), exist_ok=True)
    if args.resume_from_checkpoint:
        path = os.path.basename(args.resume_from_checkpoint) if args.resume_from_checkpoint != "latest" else ck.latest_checkpoint(args.output_dir)
        if path is not None:
            global_step = ck.load_state(D, os.path.join(args.output_dir, path))     # the heads restart from scratch, as in the reference
            gen_steps = global_step // 2
    base.reseed_for_resume(src, args, rank, global_step)
    logf = open(os.path.join(args.output_dir, args.logging_dir, f"{args.tracker_project_name}.jsonl"), "a") if rank == 0 else None
    logger.info("***** Running training *****  world=%d  per-GPU batch=%d  total steps=%d  LoRA modules=%d", world, args.train_batch_size,
                args.max_train_steps, len(lora.modules))
    while global_step < args.max_train_steps:
        latents, pe, pp = src.batch()
        B = latents.shape[0]
        noise = torch.randn(latents.shape, generator=src.g, device=device)
        nf, nr = (torch.randn(latents.shape, generator=src.g, device=device, dtype=torch.float64) for _ in range(2))      # :1436-1445
        index = torch.randint(0, args.num_euler_timesteps, (B,), generator=src.g, device=device)
        adv_u = torch.rand(B, generator=src.g, device=device)                                                              # :1413-1422
        lr = base.lr_at(args, base.sched_pos(D, args, gen_steps))                                                                                    # lr_scheduler.step() on G steps
        t0 = time.time()
        out = D.step_adv(global_step, latents, pe, pp, src.uncond, src.uncond_pooled, noise, index, nf, nr, adv_u, lr=lr)
        if not out["is_d"]:
            gen_steps += 1
        global_step += 1
        if rank == 0:
            rec = {"step": global_step, "lr": lr, "sec": time.time() - t0}
            if out["is_d"]:
                rec["d_loss"] = float(out["d_loss"].item())
            else:
                rec["loss_cm"], rec["g_loss"] = float(out["loss_cm"].item()), float(out["g_loss"].item())
            logf.write(json.dumps(rec) + "\n"); logf.flush()
            if global_step % 10 == 0 or global_step <= 2:
                logger.info("%s", rec)
            if global_step % args.checkpointing_steps == 0:
                ck.rotate_checkpoints(args.output_dir, args.checkpoints_total_limit)
                ck.save_state(D, os.path.join(args.output_dir, f"checkpoint-{global_step}"), global_step)
    if world > 1:
        torch.distributed.barrier()
    if rank == 0:
        ck.save_lora_sd3(lora, args.output_dir)
        logf.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(parse_args())
