#!/usr/bin/env python
"""train_pcm_lora_sdxl_adv.py — PCM-LoRA distillation of the SDXL UNet on MI355X (BASELINE.json configs[3]).

Takes the launch line of code/text_to_image_sdxl/train_pcm_lora_sdxl_adv.py (its 55 flags: the SD1.5 set + --adv_weight, --adv_lr,
--train_shards_path_or_url, --use_fix_crop_and_size; reference defaults resolution 1024, w_min 3, multiphase 4).  The consistency
distillation step (same math as SD1.5 with ``added_cond_kwargs``: pooled text embeds + 6 time ids, zero unconditional embeds,
:1113-1131, :1216-1221, :1300-1460) runs on the generalised UNet.  ``--adv_weight`` > 0 (reference default 0.1) adds the SDXL latent
discriminator (discriminator_sdxl.py: teacher features after the three down blocks and the mid block, one 1x1-conv head each) with the
even/odd D/G alternation of :1483-1529; ``--adv_weight 0`` is pure phased-consistency distillation.

Data: ``--latents_dir`` shards with ``latents`` [N,4,128,128], ``prompt_embeds`` [N,77,2048], ``pooled_prompt_embeds`` [N,1280]
(VAE / the two CLIP encoders are upstream of the path), or ``--synthetic_data``."""
import glob
import json
import logging
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import train_pcm_lora_sd15 as base  # noqa: E402

logger = logging.getLogger("pcm_amd")
SDXL_DEFAULTS = dict(resolution=1024, dataloader_num_workers=0, w_min=3.0, multiphase=4)


def parse_args(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    extra = {"adv_weight": 0.1, "adv_lr": 1e-5, "train_shards_path_or_url": None, "use_fix_crop_and_size": False}
    rest, i = [], 0
    while i < len(argv):
        a, hit = argv[i], False
        for k in extra:
            if a == "--" + k:
                if k == "use_fix_crop_and_size":
                    extra[k] = True; i += 1
                else:
                    extra[k] = argv[i + 1]; i += 2
                hit = True
                break
            if a.startswith("--" + k + "="):
                extra[k] = a.split("=", 1)[1]; i += 1; hit = True
                break
        if not hit:
            rest.append(a); i += 1
    given = {a.split("=", 1)[0].lstrip("-") for a in rest if a.startswith("--")}
    args = base.parse_args(rest)
    for k, v in SDXL_DEFAULTS.items():
        if k not in given:
            setattr(args, k, v)
    args.adv_weight, args.adv_lr = float(extra["adv_weight"]), float(extra["adv_lr"])
    args.train_shards_path_or_url, args.use_fix_crop_and_size = extra["train_shards_path_or_url"], bool(extra["use_fix_crop_and_size"])
    return args


class SdxlSource:
    """Per-rank batches: latents, prompt embeds [77, 2048], pooled embeds [1280]; time ids = (original size, crop top-left, target size)."""

    def __init__(self, args, rank, world, device):
        self.bs, self.device, self.hw = args.train_batch_size, device, args.resolution // 8
        if float(getattr(args, "proportion_empty_prompts", 0) or 0) > 0:
            raise SystemExit("pcm_amd: --proportion_empty_prompts needs the two text encoders' output for the empty caption, which is upstream "
                             "of this path: drop the captions when the embedding shards are written")
        self.g = torch.Generator(device=device).manual_seed((args.seed or 0) + rank)
        self.data = None
        if args.latents_dir:
            from safetensors.torch import load_file
            files = sorted(glob.glob(os.path.join(args.latents_dir, "*.safetensors")))[rank::world]
            if not files:
                raise FileNotFoundError(f"no shards for rank {rank} in {args.latents_dir}")
            d = [load_file(f) for f in files]
            self.data = {k: torch.cat([x[k] for x in d]).float().to(device) for k in ("latents", "prompt_embeds", "pooled_prompt_embeds")}
        elif not args.synthetic_data:
            raise SystemExit("pcm_amd: give --latents_dir or --synthetic_data (VAE / text encoders are out of scope)")
        r = args.resolution
        self.time_ids = torch.tensor([[r, r, 0, 0, r, r]] * self.bs, device=device)                       # :1115-1122
        pe_shape, pp_dim = ((77, 2048), 1280) if not self.data else (tuple(self.data["prompt_embeds"].shape[1:]), self.data["pooled_prompt_embeds"].shape[1])
        self.uncond = torch.zeros(self.bs, *pe_shape, device=device)                                       # zero uncond embeds, :1216-1221
        self.uncond_pooled = torch.zeros(self.bs, pp_dim, device=device)

    def __len__(self):
        return (self.data["latents"].shape[0] // self.bs) if self.data else 10 ** 9

    def batch(self):
        if self.data:
            idx = torch.randint(0, self.data["latents"].shape[0], (self.bs,), generator=self.g, device=self.device)
            return tuple(self.data[k][idx].contiguous() for k in ("latents", "prompt_embeds", "pooled_prompt_embeds"))
        rn = lambda *s: torch.randn(*s, generator=self.g, device=self.device)
        return rn(self.bs, 4, self.hw, self.hw), rn(self.bs, 77, 2048), rn(self.bs, 1280)


def main(args):
    from pcm_amd import capi, checkpoint as ck
    from pcm_amd.model import LoraState, UNetWeights
    from pcm_amd.discriminator import ADAPTER_DIMS_SDXL, Discriminator
    from pcm_amd.trainer import AdvDistiller, Distiller, StepConfig
    from pcm_amd.unet_spec import UNetConfig, random_state_dict
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = max(args.local_rank, 0)
    if args.gradient_accumulation_steps != 1:
        raise SystemExit("pcm_amd: --gradient_accumulation_steps != 1 is not implemented for this trainer (reference recipes use 1)")
    if args.scale_lr:            # train_pcm_lora_sdxl_adv.py:1166-1172 (unlike the SD1.5 script, which never reads the flag)
        args.learning_rate = args.learning_rate * args.gradient_accumulation_steps * args.train_batch_size * world
    logging.basicConfig(format="%(asctime)s - %(levelname)s - %(name)s - %(message)s", level=logging.INFO if rank == 0 else logging.WARNING)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world)
    device = base.pick_device(local_rank)
    base.apply_mixed_precision(args)
    capi.lib()
    ucfg = base.unet_config(args, "sdxl")
    sd = random_state_dict(ucfg, 0, device) if args.pretrained_teacher_model == "random" else ck.load_unet_state_dict(args.pretrained_teacher_model)
    W = UNetWeights(ucfg, sd, device)
    Wt = base.teacher_weights_for(args, ucfg, sd, device)
    del sd
    lora = LoraState(ucfg, args.lora_rank, 8.0, device, seed=(args.seed or 0))
    if world > 1:
        torch.distributed.broadcast(lora.params, src=0); lora.repack()
    cfg = StepConfig(num_ddim_timesteps=args.num_ddim_timesteps, multiphase=args.multiphase, w_min=args.w_min, w_max=args.w_max,
                     loss_type=args.loss_type, huber_c=args.huber_c, learning_rate=args.learning_rate, adam_beta1=args.adam_beta1,
                     adam_beta2=args.adam_beta2, adam_weight_decay=args.adam_weight_decay, adam_epsilon=args.adam_epsilon,
                     max_grad_norm=args.max_grad_norm, lora_rank=args.lora_rank, not_apply_cfg_solver=args.not_apply_cfg_solver)
    adv = args.adv_weight != 0
    if adv:
        b = ucfg.block_out_channels
        dims = ADAPTER_DIMS_SDXL if not getattr(args, "tiny_model", False) else tuple(b) + (b[-1],)            # down-block outputs + mid
        disc = Discriminator(dims, num_h_per_head=1, device=device, seed=(args.seed or 0) + 1, ksize=1, taps="down_mid")
        if world > 1:
            torch.distributed.broadcast(disc.params, src=0); disc.repack()
        D = AdvDistiller(W, lora, cfg, disc, adv_weight=args.adv_weight, adv_lr=args.adv_lr, world_size=world, teacher_weights=Wt)
    else:
        D = Distiller(W, lora, cfg, world_size=world, teacher_weights=Wt)
    src = SdxlSource(args, rank, world, device)
    if args.max_train_steps is None:
        args.max_train_steps = args.num_train_epochs * base.agreed_steps_per_epoch(len(src), world)
    if rank == 0:
        os.makedirs(os.path.join(args.output_dir, args.logging_dir), exist_ok=True)
    global_step = gen_steps = 0
    if args.resume_from_checkpoint:
        path = os.path.basename(args.resume_from_checkpoint) if args.resume_from_checkpoint != "latest" else ck.latest_checkpoint(args.output_dir)
        if path is not None:
            global_step = ck.load_state(D, os.path.join(args.output_dir, path))
            gen_steps = global_step // 2 if adv else global_step       # the lr schedule counts generator steps (every second step)
    logf = open(os.path.join(args.output_dir, args.logging_dir, f"{args.tracker_project_name}.jsonl"), "a") if rank == 0 else None
    cpu_gen = base.reseed_for_resume(src, args, rank, global_step)
    uac = dict(text_embeds=src.uncond_pooled, time_ids=src.time_ids)
    while global_step < args.max_train_steps:
        latents, pe, pooled = src.batch()
        B = latents.shape[0]
        noise = torch.randn(latents.shape, generator=src.g, device=device)
        index = torch.randint(0, args.num_ddim_timesteps, (B,), generator=src.g, device=device)
        w = ((args.w_max - args.w_min) * torch.rand((B,), generator=cpu_gen) + args.w_min).to(device)
        ac = dict(text_embeds=pooled, time_ids=src.time_ids)
        t0 = time.time()
        if adv:
            lr = base.lr_at(args, base.sched_step(base.sched_pos(D, args, gen_steps), world))    # the lr schedule advances on generator steps only (:1527)
            rn = lambda: torch.randn(latents.shape, generator=src.g, device=device)
            out = D.step_adv(global_step, latents, pe, src.uncond, noise, index, w, rn(), rn(), torch.rand(B, generator=src.g, device=device),
                             lr=lr, added_cond=ac, uncond_added_cond=uac)
            gen_steps += 0 if out["is_d"] else 1
        else:
            lr = base.lr_at(args, base.sched_step(base.sched_pos(D, args, global_step), world))
            out = D.step(latents, pe, src.uncond, noise, index, w, lr=lr, added_cond=ac, uncond_added_cond=uac)
        global_step += 1
        if rank == 0:
            if adv:
                rec = {"step": global_step, "lr": lr, "sec": time.time() - t0}
                if out["is_d"]:
                    rec["d_loss"] = float(out["d_loss"].item())
                else:
                    rec["loss_cm"], rec["g_loss"] = float(out["loss_cm"].item()), float(out["g_loss"].item())
            else:
                rec = {"step": global_step, "loss": float(out["loss"].item()), "lr": lr, "grad_norm": D.grad_norm(), "sec": time.time() - t0}
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
