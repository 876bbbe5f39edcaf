#!/usr/bin/env python
"""train_pcm_lora_sd15.py — MI355X-native PCM-LoRA distillation of SD1.5.

Drop-in for the reference CLI (code/text_to_image_sd15/train_pcm_lora_sd15.py:381-735, launched by
train_pcm_lora_sd15.sh): same flag names and defaults, argparse prefix abbreviations still parse
(`--tracker_project_nam=` in the reference .sh).  One process per GPU:
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_pcm_lora_sd15.py ...
replaces `accelerate launch`; gradients of the 67 M LoRA parameters are all-reduced over RCCL/xGMI.

Scope notes: the VAE / CLIP encoders are upstream of the hot path and not part of this
build: batches come from ``--latents_dir`` (safetensors shards with ``latents`` [N,4,64,64] already
scaled by 0.18215 and ``prompt_embeds`` [N,77,768]; ``uncond_prompt_embeds`` [77,768] in any shard)
or from ``--synthetic_data`` (seeded N(0,1), as the benchmark uses).  ``--pretrained_teacher_model``
is a diffusers directory (unet/diffusion_pytorch_model.safetensors) or the literal ``random``.
Flags of the reference that only drive out-of-scope subsystems are accepted and ignored (listed
in IGNORED below) so the reference's launch lines keep working.
"""
import argparse
import glob
import json
import logging
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

logger = logging.getLogger("pcm_amd")
IGNORED = ["pretrained_vae_model_name_or_path", "teacher_revision", "revision", "cache_dir", "center_crop", "random_flip",
           "dataloader_num_workers", "scale_lr", "use_8bit_adam",
           "allow_tf32", "cast_teacher_unet", "enable_xformers_memory_efficient_attention", "gradient_checkpointing",
           "push_to_hub", "hub_token", "hub_model_id", "validation_steps"]


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="PCM-LoRA distillation (SD1.5) on MI355X")
    # ---- reference flags, reference defaults (train_pcm_lora_sd15.py:381-735) ----
    p.add_argument("--pretrained_teacher_model", type=str, default=None, required=True)
    p.add_argument("--pretrained_vae_model_name_or_path", type=str, default=None)
    p.add_argument("--teacher_revision", type=str, default=None)
    p.add_argument("--revision", type=str, default=None)
    p.add_argument("--output_dir", type=str, default="lcm-xl-distilled")
    p.add_argument("--cache_dir", type=str, default=None)
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--logging_dir", type=str, default="logs")
    p.add_argument("--report_to", type=str, default="tensorboard")
    p.add_argument("--checkpointing_steps", type=int, default=500)
    p.add_argument("--checkpoints_total_limit", type=int, default=None)
    p.add_argument("--resume_from_checkpoint", type=str, default=None)
    p.add_argument("--resolution", type=int, default=512)
    p.add_argument("--center_crop", default=False, action="store_true")
    p.add_argument("--random_flip", action="store_true")
    p.add_argument("--dataloader_num_workers", type=int, default=8)
    p.add_argument("--train_batch_size", type=int, default=16)
    p.add_argument("--num_train_epochs", type=int, default=100)
    p.add_argument("--max_train_steps", type=int, default=None)
    p.add_argument("--max_train_samples", type=int, default=None)
    p.add_argument("--learning_rate", type=float, default=1e-4)
    p.add_argument("--scale_lr", action="store_true", default=False)
    p.add_argument("--lr_scheduler", type=str, default="constant")
    p.add_argument("--lr_warmup_steps", type=int, default=500)
    p.add_argument("--gradient_accumulation_steps", type=int, default=1)
    p.add_argument("--use_8bit_adam", action="store_true")
    p.add_argument("--adam_beta1", type=float, default=0.9)
    p.add_argument("--adam_beta2", type=float, default=0.999)
    p.add_argument("--adam_weight_decay", type=float, default=1e-2)
    p.add_argument("--adam_epsilon", type=float, default=1e-08)
    p.add_argument("--max_grad_norm", default=1.0, type=float)
    p.add_argument("--proportion_empty_prompts", type=float, default=0)
    p.add_argument("--w_min", type=float, default=5.0)
    p.add_argument("--w_max", type=float, default=15.0)
    p.add_argument("--num_ddim_timesteps", type=int, default=50)
    p.add_argument("--loss_type", type=str, default="l2", choices=["l2", "huber"])
    p.add_argument("--huber_c", type=float, default=0.001)
    p.add_argument("--lora_rank", type=int, default=64)
    p.add_argument("--mixed_precision", type=str, default=None, choices=["no", "fp16", "bf16"])
    p.add_argument("--allow_tf32", action="store_true")
    p.add_argument("--cast_teacher_unet", action="store_true")
    p.add_argument("--teacher_precision", type=str, default="reference", choices=["reference", "same", "fp16"], help=TEACHER_PRECISION_HELP)
    p.add_argument("--enable_xformers_memory_efficient_attention", action="store_true")
    p.add_argument("--gradient_checkpointing", action="store_true")
    p.add_argument("--local_rank", type=int, default=-1)
    p.add_argument("--validation_steps", type=int, default=200)
    p.add_argument("--push_to_hub", action="store_true")
    p.add_argument("--hub_token", type=str, default=None)
    p.add_argument("--hub_model_id", type=str, default=None)
    p.add_argument("--tracker_project_name", type=str, default="text2image-fine-tune")
    p.add_argument("--not_apply_cfg_solver", action="store_true")
    p.add_argument("--multiphase", default=8, type=int)
    # ---- additions of this build ----
    p.add_argument("--latents_dir", type=str, default=None, help="safetensors shards of precomputed latents / prompt embeds")
    p.add_argument("--synthetic_data", action="store_true", help="seeded N(0,1) latents / prompt embeds")
    p.add_argument("--ema_rate", type=float, default=None, help="enable the reference's (dead) update_ema on a shadow copy")
    p.add_argument("--tiny_model", action="store_true", help="(with random weights) a narrow UNet of the same topology for smoke tests of the CLI itself")
    args = p.parse_args(argv)
    env_local_rank = int(os.environ.get("LOCAL_RANK", -1))      # :730-732
    if env_local_rank != -1 and env_local_rank != args.local_rank:
        args.local_rank = env_local_rank
    return args


class LatentSource:
    """Per-rank batch provider (stands in for CustomImageDataset + VAE + CLIP, :75-117,:1118-1136)."""

    def __init__(self, args, rank, world, device):
        self.bs, self.device = args.train_batch_size, device
        self.p_empty = float(getattr(args, "proportion_empty_prompts", 0) or 0)
        if not 0.0 <= self.p_empty <= 1.0:
            raise ValueError("`--proportion_empty_prompts` must be in the range [0, 1].")        # :733-734
        self.hw = args.resolution // 8                      # VAE downsampling factor: synthetic latents follow --resolution
        self.g = torch.Generator(device=device).manual_seed((args.seed or 0) + rank)
        self.shards, self.uncond = None, None
        if args.latents_dir:
            from safetensors.torch import load_file
            every = sorted(glob.glob(os.path.join(args.latents_dir, "*.safetensors")))
            files = every[rank::world]
            self.uncond = find_in_shards(every, "uncond_prompt_embeds", device)      # one global tensor: "in any shard", whichever rank reads it
            if not files:
                raise FileNotFoundError(f"no shards for rank {rank} in {args.latents_dir}")
            data = [load_file(f) for f in files]
            self.lat = torch.cat([d["latents"] for d in data]).float().to(device)
            self.pe = torch.cat([d["prompt_embeds"] for d in data]).float().to(device)   # resident in HBM (288 GB)
            if getattr(args, "max_train_samples", None):            # debugging knob of the reference's parser: truncate the training set
                n = max(1, args.max_train_samples // world)
                self.lat, self.pe = self.lat[:n], self.pe[:n]
            self.shards = True
        elif not args.synthetic_data:
            raise SystemExit("pcm_amd: give --latents_dir or --synthetic_data (VAE/CLIP encoding is out of scope, see --help)")
        if self.uncond is None and self.shards:
            # the CFG teacher step needs the CLIP encoding of the empty caption (train_pcm_lora_sd15.py:1053-1059); zeros or noise in its
            # place would distill against a meaningless guidance direction without any sign of it
            raise SystemExit("pcm_amd: --latents_dir shards carry no 'uncond_prompt_embeds' [77,768] (the encoding of the empty prompt): "
                             "add it to any one shard")
        if self.uncond is None:       # --synthetic_data only
            self.uncond = torch.randn(77, 768, generator=self.g, device=device)
        self.uncond = self.uncond.expand(self.bs, *self.uncond.shape[-2:]).contiguous()

    def __len__(self):
        return (self.lat.shape[0] // self.bs) if self.shards else 10 ** 9

    def batch(self):
        if self.shards:
            idx = torch.randint(0, self.lat.shape[0], (self.bs,), generator=self.g, device=self.device)
            lat, pe = self.lat[idx].contiguous(), self.pe[idx].contiguous()
        else:
            lat = torch.randn(self.bs, 4, self.hw, self.hw, generator=self.g, device=self.device)
            pe = torch.randn(self.bs, 77, 768, generator=self.g, device=self.device)
        if self.p_empty > 0:        # caption dropout (:740-746): the caption becomes "" -> its encoding is the unconditional embedding
            drop = torch.rand(self.bs, generator=self.g, device=self.device) < self.p_empty
            pe = torch.where(drop[:, None, None], self.uncond, pe)
        return lat, pe


def find_in_shards(files, key, device):
    """A run-global tensor (the unconditional embedding) may sit in ANY shard, not necessarily in one this rank trains on."""
    from safetensors import safe_open
    for f in files:
        with safe_open(f, "pt") as sf:
            if key in sf.keys():
                return sf.get_tensor(key).float().to(device)
    return None


def agreed_steps_per_epoch(n_local, world):
    """Ranks read disjoint shard subsets ([rank::world]) that need not hold the same number of samples; every rank must run the SAME number
    of steps or the gradient all-reduce of the longer ranks never completes: take the minimum over the ranks (one tiny all-reduce)."""
    if world <= 1 or not torch.distributed.is_initialized():
        return n_local
    backend = torch.distributed.get_backend()
    t = torch.tensor([n_local], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
    return int(t.item())


def pick_device(local_rank):
    """cuda:<local_rank>; PCM_CLI_DEVICE=cpu (tests: the CLI end to end on the host emulator, which the test installs as the library)."""
    if os.environ.get("PCM_CLI_DEVICE") == "cpu":
        return torch.device("cpu")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    return dev


def unet_config(args, kind="sd15"):
    """the trainer's UNet: SD1.5 / SDXL, or with --tiny_model (random weights only) a narrow config of the same topology."""
    from pcm_amd.unet_spec import UNetConfig
    tiny = getattr(args, "tiny_model", False) and args.pretrained_teacher_model == "random"
    if kind == "sdxl":
        if tiny:
            return UNetConfig(block_out_channels=(64, 128), cross_attention_dim=64, heads=(1, 2), down_attn=(False, True),
                              transformer_depth=(1, 2), use_linear_projection=True, addition_time_embed_dim=32,
                              projection_class_embeddings_input_dim=64 + 6 * 32, layers_per_block=1)
        return UNetConfig.sdxl()
    if tiny:
        return UNetConfig(block_out_channels=(64, 128), layers_per_block=1, cross_attention_dim=64, heads=2)
    return UNetConfig.sd15()


def sched_step(step, world):
    """Scheduler position after ``step`` optimizer steps.  The SD1.5 / SDXL scripts build the schedule with the raw
    ``lr_warmup_steps`` / ``max_train_steps`` (:1026-1031) and hand it to ``accelerator.prepare``: accelerate's AcceleratedScheduler
    (0.27.2, split_batches=False) then advances it ``num_processes`` times per optimizer step, so every non-constant schedule runs
    ``world`` times faster on ``world`` GPUs.  (The SD3 scripts multiply both counts by num_processes, which cancels this.)"""
    return step * world


def reseed_for_resume(src, args, rank, global_step):
    """A resumed run must not replay the noise / timestep / data draws of step 0: the stream generators are re-seeded with an offset
    derived from the restored step count (the reference resumes accelerate's saved RNG state; here the state is a function of the step).
    Returns the CPU generator for the guidance-scale draws."""
    off = 7919 * int(global_step)
    if off:
        src.g.manual_seed((args.seed or 0) + rank + off)
    return torch.Generator().manual_seed((args.seed or 0) + rank + off)


def sched_pos(D, args, host_steps):
    """optimizer steps the lr scheduler has seen.  accelerate skips ``lr_scheduler.step()`` when GradScaler skipped the optimizer step
    (fp16 overflow): under --mixed_precision=fp16 with a non-constant schedule the position is the trainer's device-side count of
    APPLIED steps (one host sync per step, like GradScaler's own found_inf read); otherwise the host counter (no step is ever skipped)."""
    if getattr(args, "mixed_precision", None) == "fp16" and args.lr_scheduler != "constant":
        return D.applied_steps()
    return host_steps


def lr_at(args, step):
    """get_scheduler(args.lr_scheduler, ...) (:1026-1031): 'constant' ignores warmup (App. A.6)."""
    if args.lr_scheduler == "constant":
        return args.learning_rate
    if args.lr_scheduler == "constant_with_warmup":
        # diffusers get_constant_schedule_with_warmup's lambda: step / max(1, warmup) while step < warmup, else 1 (so warmup 0 -> 1.0 at step 0)
        return args.learning_rate * (1.0 if step >= args.lr_warmup_steps else step / max(1.0, args.lr_warmup_steps))
    if args.lr_scheduler == "linear":
        w = args.lr_warmup_steps
        if step < w:
            return args.learning_rate * step / max(1, w)
        return args.learning_rate * max(0.0, (args.max_train_steps - step) / max(1, args.max_train_steps - w))
    if args.lr_scheduler == "cosine":
        w = args.lr_warmup_steps
        if step < w:
            return args.learning_rate * step / max(1, w)
        prog = (step - w) / max(1, args.max_train_steps - w)
        return args.learning_rate * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))
    if args.lr_scheduler == "cosine_with_restarts":      # diffusers get_cosine_with_hard_restarts_schedule_with_warmup (--lr_num_cycles)
        w = args.lr_warmup_steps
        if step < w:
            return args.learning_rate * step / max(1, w)
        prog = (step - w) / max(1, args.max_train_steps - w)
        if prog >= 1.0:
            return 0.0
        return args.learning_rate * max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((getattr(args, "lr_num_cycles", 1) * prog) % 1.0))))
    if args.lr_scheduler == "polynomial":                 # diffusers get_polynomial_decay_schedule_with_warmup (--lr_power, lr_end 1e-7)
        w, lr_end, power = args.lr_warmup_steps, 1e-7, getattr(args, "lr_power", 1.0)
        if step < w:
            return args.learning_rate * step / max(1, w)
        if step > args.max_train_steps:
            return lr_end
        pct = 1 - (step - w) / max(1, args.max_train_steps - w)
        return (args.learning_rate - lr_end) * pct ** power + lr_end
    raise ValueError(f"unsupported --lr_scheduler {args.lr_scheduler}")


def apply_mixed_precision(args):
    """--mixed_precision (handed to accelerate at train_pcm_lora_sd15.py:1034).  fp16: accelerate's fp16 autocast + GradScaler (:1296-1299)
    = the IEEE-half build of the kernel library with the GradScaler state on the device (pcm_amd/precision.py, trainer.Distiller /
    AdvDistiller).  "bf16" / None / "no": the bfloat16 build (there is no fp32-storage build; "no" is computed in bf16).  Call before any
    weights are packed.  Shared by the SD1.5, SD1.5-adversarial and SDXL-adversarial scripts."""
    if getattr(args, "mixed_precision", None) == "fp16":
        from pcm_amd import precision
        precision.set_precision("fp16")
        logger.info("--mixed_precision=fp16: half build of the kernel library (lib/libpcm_hip_f16.so), dynamic loss scaling on the device")


TEACHER_PRECISION_HELP = ("format of the ODE-solver teacher pass.  The reference runs it under torch.autocast('cuda') with no dtype "
                          "(train_pcm_lora_sd15.py:1217-1218), i.e. in IEEE half whatever --mixed_precision says.  'reference' (default, round 6): "
                          "follow that -- IEEE half next to a bfloat16 student (a second, half packing of the frozen weights: +1.7 GB at SD1.5 "
                          "size; the step costs ~1.5 %% more), one format under --mixed_precision=fp16; 'fp16': the same, named explicitly; "
                          "'same': one format for every pass, one weight packing (what bench.py measures: BASELINE.json's configs say bf16)")


def teacher_weights_for(args, ucfg, sd, device, weights_cls=None):
    """--teacher_precision fp16 under a bfloat16 student: the frozen weights packed a second time, in IEEE half, for the ODE-solver teacher pass
    (trainer.Distiller ``teacher_weights``); None when every pass runs in the one format of the process.  No backward operands: the pass has none.
    ``weights_cls``: UNetWeights (default) or MMDiTWeights."""
    from pcm_amd import precision
    if weights_cls is None:
        from pcm_amd.model import UNetWeights as weights_cls
    mode = getattr(args, "teacher_precision", "reference")
    if mode == "reference":      # the reference's dtype-less autocast: IEEE half for the teacher pass in every run
        mode = "fp16"
    if mode != "fp16" or precision.precision() == "fp16":
        return None
    with precision.format_scope("fp16"):
        Wt = weights_cls(ucfg, sd, device, need_bwd=False)
    logger.info("--teacher_precision=%s: ODE-solver teacher pass in IEEE half (lib/libpcm_hip_f16.so) next to the bfloat16 student, as the "
                "reference's torch.autocast('cuda') without a dtype" % getattr(args, "teacher_precision", "reference"))
    return Wt


def main(args):
    from pcm_amd import capi, checkpoint as ck
    from pcm_amd.model import LoraState, UNetWeights
    from pcm_amd.trainer import Distiller, StepConfig
    from pcm_amd.unet_spec import UNetConfig, random_state_dict
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = max(args.local_rank, 0)
    logging.basicConfig(format="%(asctime)s - %(levelname)s - %(name)s - %(message)s", datefmt="%m/%d/%Y %H:%M:%S",
                        level=logging.INFO if rank == 0 else logging.WARNING)
    apply_mixed_precision(args)
    if args.gradient_accumulation_steps < 1:
        raise SystemExit("pcm_amd: --gradient_accumulation_steps must be >= 1")
    ignored = [k for k in IGNORED if getattr(args, k) not in (None, False, 0, 8, 200)]
    if ignored:
        logger.info("flags accepted for CLI compatibility and ignored: %s", ", ".join(ignored))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world)
    device = pick_device(local_rank)
    capi.lib()
    if args.seed is not None:
        torch.manual_seed(args.seed + rank)                       # set_seed(seed + process_index), :795-797
    ucfg = unet_config(args)
    if args.pretrained_teacher_model == "random":
        sd = random_state_dict(ucfg, seed=0, device=device)
    else:
        sd = ck.load_unet_state_dict(args.pretrained_teacher_model)
    W = UNetWeights(ucfg, sd, device)
    Wt = teacher_weights_for(args, ucfg, sd, device)
    del sd
    lora = LoraState(ucfg, args.lora_rank, 8.0, device, seed=(args.seed or 0))
    if world > 1:
        torch.distributed.broadcast(lora.params, src=0)           # identical adapters on every rank (DDP init)
        lora.repack()
    cfg = StepConfig(num_ddim_timesteps=args.num_ddim_timesteps, multiphase=args.multiphase, w_min=args.w_min, w_max=args.w_max,
                     loss_type=args.loss_type, huber_c=args.huber_c, learning_rate=args.learning_rate, adam_beta1=args.adam_beta1,
                     adam_beta2=args.adam_beta2, adam_weight_decay=args.adam_weight_decay, adam_epsilon=args.adam_epsilon,
                     max_grad_norm=args.max_grad_norm, lora_rank=args.lora_rank, not_apply_cfg_solver=args.not_apply_cfg_solver,
                     ema_rate=args.ema_rate)
    D = Distiller(W, lora, cfg, world_size=world, teacher_weights=Wt)
    src = LatentSource(args, rank, world, device)
    steps_per_epoch = agreed_steps_per_epoch(len(src), world)
    if args.max_train_steps is None:
        args.max_train_steps = args.num_train_epochs * steps_per_epoch
    global_step = 0
    if rank == 0:
        os.makedirs(args.output_dir, exist_ok=True)
        os.makedirs(os.path.join(args.output_dir, args.logging_dir), exist_ok=True)
    if args.resume_from_checkpoint:                                # :1081-1105
        path = os.path.basename(args.resume_from_checkpoint) if args.resume_from_checkpoint != "latest" else ck.latest_checkpoint(args.output_dir)
        if path is None:
            logger.info("Checkpoint '%s' does not exist. Starting a new training run.", args.resume_from_checkpoint)
        else:
            logger.info("Resuming from checkpoint %s", path)
            global_step = ck.load_state(D, os.path.join(args.output_dir, path))
    logf = open(os.path.join(args.output_dir, args.logging_dir, f"{args.tracker_project_name}.jsonl"), "a") if rank == 0 else None
    logger.info("***** Running training *****  world=%d  per-GPU batch=%d  total steps=%d", world, args.train_batch_size, args.max_train_steps)
    cpu_gen = reseed_for_resume(src, args, rank, global_step)
    t_last = time.time()
    ga = args.gradient_accumulation_steps

    def draw_batch():
        """one (micro-)batch in the reference's draw order; drawn ONE call ahead of its use so that its teacher targets can be computed beside the
        previous batch's student work (Distiller.step(prefetch=...)): the sequence of draws -- and with it every batch -- is unchanged"""
        latents, pe = src.batch()
        noise = torch.randn(latents.shape, generator=src.g, device=device)                                      # :1139
        index = torch.randint(0, args.num_ddim_timesteps, (latents.shape[0],), generator=src.g, device=device)  # :1147
        w = ((args.w_max - args.w_min) * torch.rand((latents.shape[0],), generator=cpu_gen) + args.w_min).to(device)  # :1183 CPU RNG
        return (latents, pe, src.uncond, noise, index, w)

    prefetch_on = os.environ.get("PCM_TEACHER_PREFETCH", "1") != "0"
    left = (args.max_train_steps - global_step) * ga           # (micro-)batches still to draw
    cur = None
    if left > 0:
        cur, left = draw_batch(), left - 1
    while global_step < args.max_train_steps:
        lr = lr_at(args, sched_step(sched_pos(D, args, global_step), world))
        for micro in range(ga):                                # accelerator.accumulate(unet), :1120: one optimizer step per ga batches
            nxt = None
            if left > 0:
                nxt, left = draw_batch(), left - 1
            out = D.step(*cur, lr=lr, accum=(micro, ga), prefetch=nxt if prefetch_on else None)
            cur = nxt
        global_step += 1
        if rank == 0:
            loss = float(out["loss"].item())                       # the reference's only per-step host sync (:1367)
            now = time.time()
            rec = {"step": global_step, "loss": loss, "lr": lr, "grad_norm": D.grad_norm(), "sec": now - t_last}
            t_last = now
            logf.write(json.dumps(rec) + "\n")
            logf.flush()
            if global_step % 10 == 0 or global_step == 1:
                logger.info("step %d loss %.6f lr %.3g grad_norm %.4f (%.3f s/step)", global_step, loss, lr, rec["grad_norm"], rec["sec"])
            if global_step % args.checkpointing_steps == 0:        # :1309-1343
                ck.rotate_checkpoints(args.output_dir, args.checkpoints_total_limit)
                save_path = os.path.join(args.output_dir, f"checkpoint-{global_step}")
                ck.save_state(D, save_path, global_step)
                logger.info("Saved state to %s", save_path)
    if world > 1:
        torch.distributed.barrier()                                # accelerator.wait_for_everyone(), :1375
    if rank == 0:
        ck.save_lora(lora, args.output_dir)                        # :1376-1382
        logf.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(parse_args())
