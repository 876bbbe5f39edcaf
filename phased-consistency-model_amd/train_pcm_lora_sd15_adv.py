#!/usr/bin/env python
"""train_pcm_lora_sd15_adv.py — PCM-LoRA distillation with the latent adversarial consistency loss.

Drop-in for code/text_to_image_sd15/train_pcm_lora_sd15_adv.py: the flags of train_pcm_lora_sd15.py plus
``--adv_weight`` (0.1) and ``--adv_lr`` (1e-5) (:741-742).  Even global steps update the 36 discriminator heads,
odd steps update the LoRA student with loss_cm + adv_weight * g_loss; the lr schedule advances on generator steps only
(:1430).  The discriminator heads are not checkpointed (the reference's save hook drops them, :962-964)."""
import json
import logging
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import train_pcm_lora_sd15 as base  # noqa: E402

logger = logging.getLogger("pcm_amd")


def parse_args(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    extra, rest, i = {"adv_weight": 0.1, "adv_lr": 1e-5}, [], 0
    while i < len(argv):
        a = argv[i]
        hit = False
        for k in extra:
            if a == "--" + k:
                extra[k] = float(argv[i + 1]); i += 2; hit = True
                break
            if a.startswith("--" + k + "="):
                extra[k] = float(a.split("=", 1)[1]); i += 1; hit = True
                break
        if not hit:
            rest.append(a); i += 1
    args = base.parse_args(rest)
    args.adv_weight, args.adv_lr = extra["adv_weight"], extra["adv_lr"]
    return args


def main(args):
    from pcm_amd import capi, checkpoint as ck
    from pcm_amd.discriminator import Discriminator
    from pcm_amd.model import LoraState, UNetWeights
    from pcm_amd.trainer import AdvDistiller, StepConfig
    from pcm_amd.unet_spec import UNetConfig, random_state_dict
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = max(args.local_rank, 0)
    if args.gradient_accumulation_steps != 1:
        raise SystemExit("pcm_amd: --gradient_accumulation_steps != 1 is not implemented for this trainer (reference recipes use 1)")
    logging.basicConfig(format="%(asctime)s - %(levelname)s - %(name)s - %(message)s", level=logging.INFO if rank == 0 else logging.WARNING)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world)
    device = base.pick_device(local_rank)
    base.apply_mixed_precision(args)
    capi.lib()
    ucfg = base.unet_config(args)
    sd = random_state_dict(ucfg, 0, device) if args.pretrained_teacher_model == "random" else ck.load_unet_state_dict(args.pretrained_teacher_model)
    W = UNetWeights(ucfg, sd, device)
    Wt = base.teacher_weights_for(args, ucfg, sd, device)
    del sd
    lora = LoraState(ucfg, args.lora_rank, 8.0, device, seed=(args.seed or 0))
    b = tuple(ucfg.block_out_channels)                            # feature taps: every down-block output, mid, every up-block output
    dims = b + (b[-1],) + b[::-1] if getattr(args, "tiny_model", False) else None          # (SD1.5: discriminator_sd15.py:377)
    disc = Discriminator(**(dict(adapter_channel_dims=dims, num_h_per_head=1) if dims else {}), device=device, seed=(args.seed or 0) + 1)
    if world > 1:
        torch.distributed.broadcast(lora.params, src=0); lora.repack()
        torch.distributed.broadcast(disc.params, src=0); disc.repack()
    cfg = StepConfig(num_ddim_timesteps=args.num_ddim_timesteps, multiphase=args.multiphase, w_min=args.w_min, w_max=args.w_max,
                     loss_type=args.loss_type, huber_c=args.huber_c, learning_rate=args.learning_rate, adam_beta1=args.adam_beta1,
                     adam_beta2=args.adam_beta2, adam_weight_decay=args.adam_weight_decay, adam_epsilon=args.adam_epsilon,
                     max_grad_norm=args.max_grad_norm, lora_rank=args.lora_rank, not_apply_cfg_solver=args.not_apply_cfg_solver)
    D = AdvDistiller(W, lora, cfg, disc, adv_weight=args.adv_weight, adv_lr=args.adv_lr, world_size=world, teacher_weights=Wt)
    src = base.LatentSource(args, rank, world, device)
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
    logf = open(os.path.join(args.output_dir, args.logging_dir, f"{args.tracker_project_name}.jsonl"), "a") if rank == 0 else None
    cpu_gen = base.reseed_for_resume(src, args, rank, global_step)

    def draw_batch():
        """one batch in the reference's draw order, drawn one step ahead of its use (its ODE-solver teacher pass runs beside the previous step's
        work, AdvDistiller.step_adv(prefetch=...)); the sequence of draws is unchanged"""
        latents, pe = src.batch()
        B = latents.shape[0]
        rn = lambda: torch.randn(latents.shape, generator=src.g, device=device)
        index = torch.randint(0, args.num_ddim_timesteps, (B,), generator=src.g, device=device)
        w = ((args.w_max - args.w_min) * torch.rand((B,), generator=cpu_gen) + args.w_min).to(device)
        adv_u = torch.rand(B, generator=src.g, device=device)
        noise, noise_fake, noise_real = rn(), rn(), rn()
        return (latents, pe, src.uncond, noise, index, w, noise_fake, noise_real, adv_u)

    prefetch_on = os.environ.get("PCM_TEACHER_PREFETCH", "1") != "0"
    left = args.max_train_steps - global_step
    cur = None
    if left > 0:
        cur, left = draw_batch(), left - 1
    while global_step < args.max_train_steps:
        nxt = None
        if left > 0:
            nxt, left = draw_batch(), left - 1
        lr = base.lr_at(args, base.sched_step(base.sched_pos(D, args, gen_steps), world))
        t0 = time.time()
        out = D.step_adv(global_step, *cur, lr=lr, prefetch=nxt[:6] if (prefetch_on and nxt is not None) else None)
        cur = nxt
        if not out["is_d"]:
            gen_steps += 1
        global_step += 1
        if rank == 0:
            rec = {"step": global_step, "lr": lr, "sec": time.time() - t0}
            if out["is_d"]:
                rec["d_loss"] = float(out["d_loss"].item())                               # :1497-1509
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
        ck.save_lora(lora, args.output_dir)
        logf.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(parse_args())
