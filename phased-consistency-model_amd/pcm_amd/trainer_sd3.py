"""The phased-consistency distillation step of the SD3 variant (reference: code/text_to_image_sd3/train_pcm_lora_sd3.py:1270-1390)
as host code over the HIP kernels: flow-matching noising, online MMDiT forward (grad), frozen teacher
cond (+ uncond) with the fixed w = 3 CFG and one Euler step, target forward with the online LoRA weights under no-grad, the two
multiphase jumps (float64 through sigma_prev, like the reference), huber loss, LoRA-only backward, clip + AdamW.

Shares the optimizer / gradient-exchange half with the SD1.5 Distiller (same flat fp32 LoRA buffer, one all-reduce).
"""
import contextlib
import os

import torch

from . import fm, ops, precision
from .mmdit import MMDiT, MMDiTWeights
from .model import LoraState
from .trainer import Distiller


class SD3StepConfig:
    """Hyper-parameters with the reference's argparse names (train_pcm_lora_sd3.py:270-700; run.sh: lora_rank 32, multiphase per
    recipe, learning_rate 5e-6).  ``w`` is hard-coded to 3 in the reference (:1334); ``shift`` is the SD3 scheduler's 3.0."""

    def __init__(self, num_euler_timesteps=50, multiphase=4, w=3.0, huber_c=0.001, learning_rate=5e-6, adam_beta1=0.9, adam_beta2=0.999,
                 adam_weight_decay=1e-2, adam_epsilon=1e-8, max_grad_norm=1.0, lora_rank=32, lora_alpha=8.0, not_apply_cfg_solver=False,
                 num_train_timesteps=1000, shift=3.0, ema_rate=None):
        self.__dict__.update({k: v for k, v in locals().items() if k != "self"})


class SD3Distiller(Distiller):
    """Owns the frozen MMDiT weights, the LoRA state and the optimizer state of one rank."""

    def __init__(self, weights: MMDiTWeights, lora: LoraState, cfg: SD3StepConfig, world_size=1, process_group=None, teacher_weights=None):
        """``teacher_weights``: the frozen weights packed in the OTHER 16-bit format for the ODE-solver teacher pass, which the reference runs
        under a dtype-less ``torch.autocast("cuda")`` = IEEE half (train_pcm_lora_sd3.py:1334); see trainer.Distiller."""
        # (Distiller.__init__ builds the UNet runners and DDIM tables; this variant has its own, the optimizer half is inherited)
        self.W, self.lora, self.cfg = weights, lora, cfg
        self.device = lora.device
        self.solver = fm.EulerSolver(fm.flow_sigmas(cfg.num_train_timesteps, cfg.shift), cfg.num_train_timesteps, cfg.num_euler_timesteps, self.device)
        self.student = MMDiT(weights, lora)
        self.teacher = MMDiT(weights, None)
        self.teacher_ode, self._ode_scope = self.teacher, contextlib.nullcontext
        if teacher_weights is not None and teacher_weights.format != precision.precision():
            fmt = teacher_weights.format
            self.teacher_ode, self._ode_scope = MMDiT(teacher_weights, None), (lambda: precision.format_scope(fmt))
        self.world_size, self.pg = world_size, process_group
        self.step_count = 0
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.lr_dev = torch.full((1,), float(cfg.learning_rate), dtype=torch.float32, device=self.device)
        self._graph = None
        self._init_loss_scaler()          # half build (--mixed_precision=fp16, every recipe of text_to_image_sd3/run.sh): device-side GradScaler state
        self.ema = lora.params.clone() if cfg.ema_rate is not None else None

    @property
    def online_target_mode(self):
        """how the online and the target forward of a step are issued: "fused" | "side" | "serial" (PCM_SD3_ONLINE_TARGET; measured in
        profiles/r06_n_*, r06_p_*)"""
        return getattr(self, "_online_target_mode", None) or os.environ.get("PCM_SD3_ONLINE_TARGET", "side")

    def _target_side(self):
        """the HIP stream the no-grad target pass is issued on beside the online pass in mode "side" (None on the host emulator: one launch chain)"""
        if self.device.type != "cuda":
            return None
        if getattr(self, "_tgt_side", None) is None:
            self._tgt_side = torch.cuda.Stream()
        return self._tgt_side

    # ---- the part of the step that reads nothing trainable (see trainer.Distiller.teacher_targets: same role, same cross-step prefetch) ----
    TARGET_KEYS = ("timesteps", "timesteps_prev", "noisy", "cond", "uncond", "x_prev64", "x_prev32")
    N_BATCH_KEY = 7

    def teacher_targets(self, model_input, prompt_embeds, pooled_prompt_embeds, uncond_prompt_embeds, uncond_pooled_prompt_embeds, noise, index):
        """train_pcm_lora_sd3.py:1291-1301 + :1332-1357: timesteps, flow-matching noising, the frozen teacher's cond (+ uncond) pass as one 2B
        pass, fixed-w CFG + one Euler step."""
        cfg, S = self.cfg, self.solver
        B = model_input.shape[0]
        timesteps, timesteps_prev = S.timesteps(index, cfg.num_train_timesteps)                               # :1291-1300
        noisy = S.add_noise(model_input, noise, index)                                                        # :1301
        with self._ode_scope():
            if cfg.not_apply_cfg_solver:
                cond = uncond = self.teacher_ode.forward(noisy, timesteps, prompt_embeds, pooled_prompt_embeds)
            else:
                both = self.teacher_ode.forward(torch.cat([noisy, noisy]), torch.cat([timesteps, timesteps]),
                                                torch.cat([prompt_embeds, uncond_prompt_embeds]), torch.cat([pooled_prompt_embeds, uncond_pooled_prompt_embeds]))
                cond, uncond = both[:B], both[B:]
        if cfg.not_apply_cfg_solver:
            x_prev64, x_prev32 = S.euler_step(noisy, cond, index, None)
        else:
            x_prev64, x_prev32 = S.euler_step(noisy, cond, index, uncond, cfg.w)
        return dict(timesteps=timesteps, timesteps_prev=timesteps_prev, noisy=noisy, cond=cond, uncond=uncond, x_prev64=x_prev64, x_prev32=x_prev32)

    def forward_backward(self, model_input, prompt_embeds, pooled_prompt_embeds, uncond_prompt_embeds, uncond_pooled_prompt_embeds, noise,
                         index, backward=True, grad_scale=1.0, zero_grad=True, targets=None):
        cfg, S = self.cfg, self.solver
        if targets is None:
            targets = self.teacher_targets(model_input, prompt_embeds, pooled_prompt_embeds, uncond_prompt_embeds, uncond_pooled_prompt_embeds, noise, index)
        timesteps, timesteps_prev, noisy, cond, uncond, x_prev64, x_prev32 = (targets[k] for k in self.TARGET_KEYS)
        # online prediction (grad, :1304-1315) and target prediction (the online weights, LoRA included, under no-grad at (x_prev, t_prev),
        # :1360-1370) with their jumps to the phase edge.  The two passes share nothing but the weights, and at this trainer's batch sizes (2 per
        # GPU: 160 tiles of a 256-CU chip per contraction) neither fills the chip:
        #   "side" (default): two B-sample passes issued on two HIP streams (one launch chain on the host emulator);
        #   "fused": ONE 2B-sample launch schedule, back-propagated through the online half of its tape (as trainer.Distiller does);
        #   "serial": one launch chain.   Measured on one MI355X, bs 2 (profiles/r06_p_*): serial 142.5, fused 136.1-136.8, side 134.8-134.9 ms per step.
        def target_pass():
            tp = self.student.forward(x_prev32, timesteps_prev.float(), prompt_embeds, pooled_prompt_embeds)
            return (tp,) + tuple(S.euler_style_multiphase_pred(x_prev64, tp, index, cfg.multiphase, True, with_f32=True))
        mode = self.online_target_mode
        side = self._target_side() if mode == "side" else None
        if mode == "fused":
            B = model_input.shape[0]
            both, tape2 = self.student.forward(torch.cat([noisy, x_prev32]), torch.cat([timesteps.float(), timesteps_prev.float()]),
                                               torch.cat([prompt_embeds, prompt_embeds]), torch.cat([pooled_prompt_embeds, pooled_prompt_embeds]), save=True)
            pred, target_pred, tape = both[:B], both[B:], self.student.tape_first_half(tape2)
            del tape2
        else:
            if side is not None:
                cur_s = torch.cuda.current_stream()
                side.wait_stream(cur_s)
                with torch.cuda.stream(side):
                    target_pred, target64, _, target32 = target_pass()
            pred, tape = self.student.forward(noisy, timesteps, prompt_embeds, pooled_prompt_embeds, save=True)
        model_pred64, end_index, model_pred32 = S.euler_style_multiphase_pred(noisy, pred, index, cfg.multiphase, with_f32=True)
        if mode == "fused":
            target64, _, target32 = S.euler_style_multiphase_pred(x_prev64, target_pred, index, cfg.multiphase, True, with_f32=True)
        elif side is not None:
            cur_s.wait_stream(side)
        else:
            target_pred, target64, _, target32 = target_pass()
        # d model_pred / d pred = sigma_prev[end] - sigma[index]  (per sample)
        coef = (S.sigmas_prev[end_index] - S.sigmas[index].double()).float().contiguous()
        loss, d_pred = ops.consistency_loss(model_pred32, target32, coef, True, cfg.huber_c, grad_scale=grad_scale)   # :1374-1379
        if self.loss_scale_dev is not None:
            ops.scale_by_dev(d_pred, self.loss_scale_dev)
        out = dict(loss=loss, noisy_model_input=noisy, model_output=pred, model_pred=model_pred64, cond_teacher_output=cond,
                   uncond_teacher_output=uncond, x_prev=x_prev64, target_pred=target_pred, target=target64, timesteps=timesteps,
                   timesteps_prev=timesteps_prev, end_index=end_index)
        if not backward:
            out["tape"], out["d_pred"] = tape, d_pred
            return out
        if zero_grad:
            self.lora.zero_grad()
        self.student.backward(d_pred, tape)                                                                    # :1381
        return out

    def step(self, model_input, prompt_embeds, pooled_prompt_embeds, uncond_prompt_embeds, uncond_pooled_prompt_embeds, noise, index,
             lr=None, update=True, accum=None, prefetch=None):
        """One distillation step on this rank's batch; all inputs device tensors (latents/noise [B,16,H,W] fp32, prompt embeds
        [B,Lc,4096], pooled [B,2048], index [B] int64).  Returns device tensors (no host sync).  ``accum=(i, k)``: micro-batch i of k
        (``--gradient_accumulation_steps``, ``accelerator.accumulate``, :1267-1268).  ``prefetch``: the NEXT call's seven batch tensors (its
        teacher targets are computed on a side stream beside this call's student work, trainer.Distiller.step)."""
        i, k = accum if accum is not None else (0, 1)
        targets = self._take_prefetched((model_input, prompt_embeds, pooled_prompt_embeds, uncond_prompt_embeds, uncond_pooled_prompt_embeds, noise, index))
        if prefetch is not None:
            self._prefetch(prefetch)
        out = self.forward_backward(model_input, prompt_embeds, pooled_prompt_embeds, uncond_prompt_embeds, uncond_pooled_prompt_embeds,
                                    noise, index, backward=update, grad_scale=1.0 / k, zero_grad=(i == 0), targets=targets)
        if not update or i < k - 1:
            return out
        if lr is not None:
            self.lr_dev.fill_(float(lr))
        self.optimizer_step()
        out["grad_sumsq"] = self.lora.gradsq
        return out

    # ---- hipGraph replay (same scheme as Distiller.capture: forward+backward and the optimizer as two graphs around the eager
    # gradient all-reduce; ``pipeline``: the next batch's teacher targets as a forked branch of the captured step) ----
    _STATIC_ORDER = ("model_input", "prompt_embeds", "pooled_prompt_embeds", "uncond_prompt_embeds", "uncond_pooled_prompt_embeds", "noise", "index")

    def _fill_static(self, st, *batch):
        for k, v in zip(self._STATIC_ORDER, batch):
            st[k].copy_(v)

    def capture(self, B, H=128, W=128, ctx_len=154, pipeline=False):
        dev, mc = self.device, self.W.cfg
        f32 = dict(dtype=torch.float32, device=dev)
        self._static = dict(model_input=torch.zeros(B, mc.in_channels, H, W, **f32), prompt_embeds=torch.zeros(B, ctx_len, mc.joint_attention_dim, **f32),
                            pooled_prompt_embeds=torch.zeros(B, mc.pooled_projection_dim, **f32),
                            uncond_prompt_embeds=torch.zeros(B, ctx_len, mc.joint_attention_dim, **f32),
                            uncond_pooled_prompt_embeds=torch.zeros(B, mc.pooled_projection_dim, **f32),
                            noise=torch.zeros(B, mc.in_channels, H, W, **f32), index=torch.zeros(B, dtype=torch.int64, device=dev))
        lo = self.lora
        state = [lo.params, lo.exp_avg, lo.exp_avg_sq, self.step_dev, self.lr_dev]
        if self.loss_scale_dev is not None:
            state += [self.loss_scale_dev, self.loss_good_dev]
        saved = [t.clone() for t in state]
        if self.ema is not None:
            saved.append(self.ema.clone())
        count = self.step_count
        self._pipeline, self._pipe_key = bool(pipeline), None
        if pipeline:
            self._static_next = {k: v.clone() for k, v in self._static.items()}
            self._pipe_side = torch.cuda.Stream()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up: lazy init, allocator pools
            if pipeline:
                tg = self.teacher_targets(**self._static_next)
                self._tg_next = {k: v.clone() for k, v in tg.items()}
                self._tg_cur = {k: v.clone() for k, v in tg.items()}
            self.forward_backward(**self._static)
            self._optimizer_apply()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

        def body():
            if not pipeline:
                return self.forward_backward(**self._static)
            for k in self.TARGET_KEYS:                     # the targets a previous replay (or the eager prologue) left for THIS batch
                self._tg_cur[k].copy_(self._tg_next[k])
            self._pipe_side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._pipe_side):
                tgn = self.teacher_targets(**self._static_next)
                for k in self.TARGET_KEYS:
                    self._tg_next[k].copy_(tgn[k])
            out_ = self.forward_backward(**self._static, targets=self._tg_cur)
            torch.cuda.current_stream().wait_stream(self._pipe_side)
            return out_
        self._g_fb, self._g_opt = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_fb, capture_error_mode="thread_local"):
            self._static_out = body()
        with torch.cuda.graph(self._g_opt, pool=self._g_fb.pool(), capture_error_mode="thread_local"):
            self._optimizer_apply()
        for dst, src in zip(state, saved):
            dst.copy_(src)
        if self.ema is not None:
            self.ema.copy_(saved[-1])
        self.step_count = count
        lo.repack()
        self._static_out["grad_sumsq"] = lo.gradsq
        self._graph = True

    def step_graphed(self, model_input, prompt_embeds, pooled_prompt_embeds, uncond_prompt_embeds, uncond_pooled_prompt_embeds, noise, index, lr=None,
                     prefetch=None):
        """Same as step() through the captured graphs; returned tensors are the graph's static outputs.  ``prefetch`` (capture(pipeline=True)):
        the NEXT call's seven batch tensors."""
        batch = (model_input, prompt_embeds, pooled_prompt_embeds, uncond_prompt_embeds, uncond_pooled_prompt_embeds, noise, index)
        self._fill_static(self._static, *batch)
        if getattr(self, "_pipeline", False):
            self._pipe_feed(batch, None, None, prefetch)
        if lr is not None:
            self.lr_dev.fill_(float(lr))
        self._g_fb.replay()
        self.all_reduce_grads()
        self.step_count += 1
        self._g_opt.replay()
        return self._static_out


class SD3AdvDistiller(SD3Distiller):
    """PCM-LoRA + latent adversarial consistency for SD3 (reference: train_pcm_lora_sd3_adv.py:1330-1520, discriminator_sd3.py).
    Even ``global_step``: discriminator (heads) update only; odd: student update with loss_cm + adv_weight * g_loss.
    The discriminator = the frozen teacher transformer's per-block image-stream states + one 1x1-conv head per block
    (``pcm_amd.discriminator.Discriminator([D] * num_layers, num_h_per_head=1, ksize=1)``)."""

    def __init__(self, weights, lora, cfg, discriminator, adv_weight=0.1, adv_lr=1e-5, loss_type="huber", world_size=1, process_group=None,
                 teacher_weights=None):
        super().__init__(weights, lora, cfg, world_size, process_group, teacher_weights=teacher_weights)
        self.disc, self.adv_weight, self.adv_lr, self.loss_type = discriminator, adv_weight, adv_lr, loss_type
        self.adv_lr_dev = torch.full((1,), float(adv_lr), dtype=torch.float32, device=self.device)
        assert discriminator.head_num == weights.cfg.num_layers and discriminator.ksize == 1 and discriminator.nh == 1

    def _feats(self, runner_out, H, W):
        return [(f, H // 2, W // 2) for f in runner_out]

    def step_adv(self, global_step, model_input, prompt_embeds, pooled_prompt_embeds, uncond_prompt_embeds, uncond_pooled_prompt_embeds, noise,
                 index, noise_fake, noise_real, adv_u, lr=None):
        """adv_u [B] in [0,1): adv_index = end_index + floor(adv_u * (E // multiphase)) (the reference's per-sample randint, :1413-1422);
        noise_fake / noise_real float64 [B,16,H,W] (its two randn_like draws, :1436-1445)."""
        cfg, S, disc = self.cfg, self.solver, self.disc
        B, _, H, Wd = model_input.shape
        is_d = (global_step % 2 == 0)
        timesteps, timesteps_prev = S.timesteps(index, cfg.num_train_timesteps)
        noisy = S.add_noise(model_input, noise, index)
        with self._ode_scope():     # the ODE-solver teacher pass only (sd3_adv.py:1408); the discriminator's feature passes stay in the build format
            if cfg.not_apply_cfg_solver:
                cond = self.teacher_ode.forward(noisy, timesteps, prompt_embeds, pooled_prompt_embeds)
            else:
                both = self.teacher_ode.forward(torch.cat([noisy, noisy]), torch.cat([timesteps, timesteps]),
                                                torch.cat([prompt_embeds, uncond_prompt_embeds]), torch.cat([pooled_prompt_embeds, uncond_pooled_prompt_embeds]))
        if cfg.not_apply_cfg_solver:
            x_prev64, x_prev32 = S.euler_step(noisy, cond, index, None)
        else:
            x_prev64, x_prev32 = S.euler_step(noisy, both[:B], index, both[B:], cfg.w)
        if is_d:        # the student forward is not back-propagated on discriminator steps: no tape
            pred, tape = self.student.forward(noisy, timesteps, prompt_embeds, pooled_prompt_embeds), None
        else:
            pred, tape = self.student.forward(noisy, timesteps, prompt_embeds, pooled_prompt_embeds, save=True)
        model_pred64, end_index, model_pred32 = S.euler_style_multiphase_pred(noisy, pred, index, cfg.multiphase, with_f32=True)
        target_pred = self.student.forward(x_prev32, timesteps_prev.float(), prompt_embeds, pooled_prompt_embeds)
        target64, _, target32 = S.euler_style_multiphase_pred(x_prev64, target_pred, index, cfg.multiphase, True, with_f32=True)
        span = cfg.num_euler_timesteps // cfg.multiphase
        adv_index = end_index + torch.clamp((adv_u * span).long(), max=span - 1)                           # :1413-1422
        t_adv = (S.sigmas_prev[adv_index] * cfg.num_train_timesteps).float()                               # :1430-1435
        _, fake32, ratio = S.noise_travel(model_pred64, noise_fake, end_index, adv_index)                  # :1441-1445
        out = dict(model_pred=model_pred64, target=target64, end_index=end_index, adv_index=adv_index, fake_adv=fake32, is_d=is_d)
        if is_d:
            _, real32, _ = S.noise_travel(target64, noise_real, end_index, adv_index)                      # :1436-1440
            feats = self.teacher.forward(torch.cat([fake32, real32]), torch.cat([t_adv, t_adv]), torch.cat([prompt_embeds, prompt_embeds]),
                                         torch.cat([pooled_prompt_embeds, pooled_prompt_embeds]), features=True)
            logits, dtape = disc.forward(self._feats(feats, H, Wd), save=True)
            disc.grads.zero_()
            out["d_loss"] = disc.d_loss_backward(logits, dtape, B, loss_scale_dev=self.loss_scale_dev)     # :1446-1466
            out["real_adv"] = real32
            self._disc_optimizer_step()
            return out
        feats, utape = self.teacher.forward(fake32, t_adv, prompt_embeds, pooled_prompt_embeds, features=True, save=True)
        logits, dtape = disc.forward(self._feats(feats, H, Wd), save=True)
        g_loss, d_feats = disc.g_loss_backward(logits, dtape, grad_scale=self.adv_weight, loss_scale_dev=self.loss_scale_dev)   # :1492-1500
        d_fake = self.teacher.backward(None, utape, d_feats=d_feats, need_input_grad=True)
        coef = (S.sigmas_prev[end_index] - S.sigmas[index].double()).float().contiguous()                  # d model_pred / d pred
        loss_cm, d_pred = ops.consistency_loss(model_pred32, target32, coef, self.loss_type == "huber", cfg.huber_c)   # :1468-1481
        if self.loss_scale_dev is not None:
            ops.scale_by_dev(d_pred, self.loss_scale_dev)
        ops.scale_add_rows(d_pred, d_fake, ratio, coef)        # d fake_adv / d model_pred = ratio; d model_pred / d pred = coef
        out.update(loss_cm=loss_cm, g_loss=g_loss, d_fake_adv=d_fake, d_pred=d_pred)
        self.lora.zero_grad()
        self.student.backward(d_pred, tape)
        if lr is not None:
            self.lr_dev.fill_(float(lr))
        self.optimizer_step()
        out["grad_sumsq"] = self.lora.gradsq
        return out

    def _disc_optimizer_step(self):
        """adv_optimizer (train_pcm_lora_sd3_adv.py:1150-1156): AdamW(lr=adv_lr, betas=(0, 0.999)) over the heads, global-norm clip (:1459-1462)."""
        cfg, d = self.cfg, self.disc
        if self.world_size > 1:
            torch.distributed.all_reduce(d.grads, op=torch.distributed.ReduceOp.SUM, group=self.pg)
        self._disc_adamw(d, self.adv_lr, self.adv_lr_dev)
