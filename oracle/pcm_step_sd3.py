"""One PCM-LoRA distillation step of the SD3 variant, restated in plain torch (fp32 / fp64 exactly where the reference is) with autograd.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows code/text_to_image_sd3/train_pcm_lora_sd3.py:1270-1390 line by line;
the transformer is oracle/mmdit_sd3.py (parity unpinned for the block internals, see its header), the PCM math is
oracle/pcm_fm_math.py (pinned bit-exactly against the reference's own source).
"""
import torch

from . import mmdit_sd3 as M
from . import pcm_fm_math as FM


def distill_step_sd3(cfg, sd, lora, model_input, prompt_embeds, pooled, uncond_prompt_embeds, uncond_pooled, noise, index, multiphase=4,
                     w=3, huber_c=0.001, lora_alpha=8.0, num_euler_timesteps=50, shift=3.0, not_apply_cfg_solver=False):
    """``lora``: {module path: (A [r, K] requires_grad, B [N, r] requires_grad)}.  Returns dict(loss, model_pred, target, x_prev, ...);
    call ``out['loss'].backward()`` for the LoRA gradients."""
    solver = FM.EulerSolver(FM.flow_sigmas(1000, shift), 1000, num_euler_timesteps)
    timesteps, timesteps_prev = FM.fm_timesteps(solver, index)                                               # :1291-1300
    noisy = FM.fm_add_noise(solver, model_input, noise, index)                                               # :1301
    pred = M.mmdit_forward(cfg, sd, noisy, timesteps, prompt_embeds, pooled, lora, lora_alpha)                # :1304-1310
    model_pred, end_index = solver.euler_style_multiphase_pred(noisy, pred, index, multiphase)               # :1313-1315
    with torch.no_grad():
        cond = M.mmdit_forward(cfg, sd, noisy, timesteps, prompt_embeds, pooled)                             # :1336-1341 (teacher: no LoRA)
        uncond = cond if not_apply_cfg_solver else M.mmdit_forward(cfg, sd, noisy, timesteps, uncond_prompt_embeds, uncond_pooled)
        teacher = cond if not_apply_cfg_solver else FM.fm_cfg(cond, uncond, w)                               # :1352-1354
        # (with --not_apply_cfg_solver the reference still evaluates cond + w*(cond - cond) == cond)
        x_prev = solver.euler_step(noisy, teacher, index)                                                    # :1355-1357
        target_pred = M.mmdit_forward(cfg, sd, x_prev.float(), timesteps_prev, prompt_embeds, pooled, lora, lora_alpha)   # :1361-1366
        target, _ = solver.euler_style_multiphase_pred(x_prev, target_pred, index, multiphase, True)         # :1368-1370
    loss = FM.huber_loss(model_pred, target, huber_c)                                                         # :1374-1379
    return dict(loss=loss, noisy_model_input=noisy, model_output=pred, model_pred=model_pred, cond_teacher_output=cond,
                uncond_teacher_output=uncond, x_prev=x_prev, target_pred=target_pred, target=target, end_index=end_index)
