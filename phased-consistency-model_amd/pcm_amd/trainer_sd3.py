"""The phased-consistency distillation step of the SD3 variant (reference: code/text_to_image_sd3/train_pcm_lora_sd3.py:1270-1390)
as host code over the HIP kernels: flow-matching noising, online MMDiT forward (grad), frozen teacher
cond (+ uncond) with the fixed w = 3 CFG and one Euler step, target forward with the online LoRA weights under no-grad, the two
multiphase jumps (float64 through sigma_prev, like the reference), huber loss, LoRA-only backward, clip + AdamW.

Shares the optimizer / gradient-exchange half with the SD1.5 Distiller (same flat fp32 LoRA buffer, one all-reduce).
"""
import torch

from . import fm, ops
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

    def __init__(self, weights: MMDiTWeights, lora: LoraState, cfg: SD3StepConfig, world_size=1, process_group=None):
        # (Distiller.__init__ builds the UNet runners and DDIM tables; this variant has its own, the optimizer half is inherited)
        self.W, self.lora, self.cfg = weights, lora, cfg
        self.device = lora.device
        self.solver = fm.EulerSolver(fm.flow_sigmas(cfg.num_train_timesteps, cfg.shift), cfg.num_train_timesteps, cfg.num_euler_timesteps, self.device)
        self.student = MMDiT(weights, lora)
        self.teacher = MMDiT(weights, None)
        self.world_size, self.pg = world_size, process_group
        self.step_count = 0
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.lr_dev = torch.full((1,), float(cfg.learning_rate), dtype=torch.float32, device=self.device)
        self._graph = None
        self.ema = lora.params.clone() if cfg.ema_rate is not None else None

    def forward_backward(self, model_input, prompt_embeds, pooled_prompt_embeds, uncond_prompt_embeds, uncond_pooled_prompt_embeds, noise,
                         index, backward=True):
        cfg, S = self.cfg, self.solver
        B = model_input.shape[0]
        timesteps, timesteps_prev = S.timesteps(index, cfg.num_train_timesteps)                               # :1291-1300
        noisy = S.add_noise(model_input, noise, index)                                                        # :1301
        # frozen teacher, cond (+ uncond) as one 2B pass, fixed-w CFG + Euler step ----------------------------- :1332-1357
        if cfg.not_apply_cfg_solver:
            cond = self.teacher.forward(noisy, timesteps, prompt_embeds, pooled_prompt_embeds)
            x_prev64, x_prev32 = S.euler_step(noisy, cond, index, None)
            uncond = cond
        else:
            both = self.teacher.forward(torch.cat([noisy, noisy]), torch.cat([timesteps, timesteps]),
                                        torch.cat([prompt_embeds, uncond_prompt_embeds]), torch.cat([pooled_prompt_embeds, uncond_pooled_prompt_embeds]))
            cond, uncond = both[:B], both[B:]
            x_prev64, x_prev32 = S.euler_step(noisy, cond, index, uncond, cfg.w)
        # online prediction (grad) and its jump to the phase edge ------------------------------------------------ :1304-1315
        pred, tape = self.student.forward(noisy, timesteps, prompt_embeds, pooled_prompt_embeds, save=True)
        model_pred64, end_index, model_pred32 = S.euler_style_multiphase_pred(noisy, pred, index, cfg.multiphase, with_f32=True)
        # target: the online weights (LoRA included) under no-grad at (x_prev, t_prev) ------------------------------ :1360-1370
        target_pred = self.student.forward(x_prev32, timesteps_prev.float(), prompt_embeds, pooled_prompt_embeds)
        target64, _, target32 = S.euler_style_multiphase_pred(x_prev64, target_pred, index, cfg.multiphase, True, with_f32=True)
        # d model_pred / d pred = sigma_prev[end] - sigma[index]  (per sample)
        coef = (S.sigmas_prev[end_index] - S.sigmas[index].double()).float().contiguous()
        loss, d_pred = ops.consistency_loss(model_pred32, target32, coef, True, cfg.huber_c)                      # :1374-1379
        out = dict(loss=loss, noisy_model_input=noisy, model_output=pred, model_pred=model_pred64, cond_teacher_output=cond,
                   uncond_teacher_output=uncond, x_prev=x_prev64, target_pred=target_pred, target=target64, timesteps=timesteps,
                   timesteps_prev=timesteps_prev, end_index=end_index)
        if not backward:
            out["tape"], out["d_pred"] = tape, d_pred
            return out
        self.lora.zero_grad()
        self.student.backward(d_pred, tape)                                                                    # :1381
        return out

    def step(self, model_input, prompt_embeds, pooled_prompt_embeds, uncond_prompt_embeds, uncond_pooled_prompt_embeds, noise, index,
             lr=None, update=True):
        """One distillation step on this rank's batch; all inputs device tensors (latents/noise [B,16,H,W] fp32, prompt embeds
        [B,Lc,4096], pooled [B,2048], index [B] int64).  Returns device tensors (no host sync)."""
        out = self.forward_backward(model_input, prompt_embeds, pooled_prompt_embeds, uncond_prompt_embeds, uncond_pooled_prompt_embeds,
                                    noise, index, backward=update)
        if not update:
            return out
        if lr is not None:
            self.lr_dev.fill_(float(lr))
        self.optimizer_step()
        out["grad_sumsq"] = self.lora.gradsq
        return out

    def capture(self, *a, **k):
        raise NotImplementedError("SD3Distiller: hipGraph capture is not wired for this variant yet (eager launches)")
