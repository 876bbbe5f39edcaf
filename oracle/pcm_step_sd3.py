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


def discriminator_forward_sd3(cfg, sd, disc_sd, sample, timestep, prompt_embeds, pooled):
    """Discriminator._forward of discriminator_sd3.py:197-214: the frozen teacher transformer's per-block image-stream states
    (``modified_forward``, :36-137) -> one DiscriminatorHead per block (:140-167: tokens [B, L, C] viewed as a C-channel map, 1x1 convs)."""
    from . import pcm_math as PM
    _, feats = M.mmdit_forward(cfg, sd, sample, timestep, prompt_embeds, pooled, return_features=True)
    B, _, H, W = sample.shape
    outs = []
    for k, f in enumerate(feats):
        x = f.permute(0, 2, 1).reshape(B, f.shape[2], H // cfg.patch_size, W // cfg.patch_size)     # reference hard-codes 64 x 64 (:163)
        outs.append(PM.discriminator_head(disc_sd, x, prefix=f"heads.{k}.0."))
    return outs


def distill_step_sd3_adv(cfg, sd, lora, disc_sd, model_input, prompt_embeds, pooled, uncond_prompt_embeds, uncond_pooled, noise, index,
                         adv_index_offset, noise_fake, noise_real, global_step, multiphase=4, w=3, huber_c=0.001, loss_type="huber",
                         adv_weight=0.1, lora_alpha=8.0, num_euler_timesteps=50, shift=3.0):
    """train_pcm_lora_sd3_adv.py:1330-1520.  ``adv_index_offset`` [B] stands for the per-sample ``torch.randint(end, end + E // multiphase)``
    draw (:1413-1422), ``noise_fake`` / ``noise_real`` (float64) for the two ``randn_like`` draws (:1436-1445).
    Even global_step -> dict(d_loss, head_grads); odd -> dict(loss, loss_cm, g_loss) with ``loss`` ready for ``.backward()``."""
    from . import pcm_math as PM
    base = distill_step_sd3(cfg, sd, lora, model_input, prompt_embeds, pooled, uncond_prompt_embeds, uncond_pooled, noise, index,
                            multiphase=multiphase, w=w, huber_c=huber_c, lora_alpha=lora_alpha, num_euler_timesteps=num_euler_timesteps, shift=shift)
    solver = FM.EulerSolver(FM.flow_sigmas(1000, shift), 1000, num_euler_timesteps)
    model_pred, target, end_index = base["model_pred"], base["target"], base["end_index"]
    adv_index = end_index + adv_index_offset
    timesteps_adv = solver.sigmas_prev[adv_index] * 1000                                                  # :1430-1435
    real_adv = FM.fm_noise_travel(solver, target, noise_real, end_index, adv_index)                       # :1436-1440
    fake_adv = FM.fm_noise_travel(solver, model_pred, noise_fake, end_index, adv_index)                   # :1441-1445
    res = dict(base, adv_index=adv_index, timesteps_adv=timesteps_adv, fake_adv=fake_adv, real_adv=real_adv)
    if global_step % 2 == 0:                                                                              # :1446-1466
        dsd = {k: v.detach().clone().requires_grad_(True) for k, v in disc_sd.items()}
        fake_o = discriminator_forward_sd3(cfg, sd, dsd, fake_adv.detach().float(), timesteps_adv, prompt_embeds, pooled)
        real_o = discriminator_forward_sd3(cfg, sd, dsd, real_adv.detach().float(), timesteps_adv, prompt_embeds, pooled)
        loss = PM.hinge_d_loss(fake_o, real_o, 1.0)
        names = list(dsd)
        grads = torch.autograd.grad(loss, [dsd[n] for n in names], allow_unused=True)
        res.update(d_loss=loss.detach(), head_grads=dict(zip(names, grads)))
        return res
    if loss_type == "l2":                                                                                 # :1468-1481
        loss_cm = torch.nn.functional.mse_loss(model_pred.float(), target.float(), reduction="mean")
    else:
        loss_cm = FM.huber_loss(model_pred, target, huber_c)
    fake_o = discriminator_forward_sd3(cfg, sd, disc_sd, fake_adv.float(), timesteps_adv, prompt_embeds, pooled)
    g_loss = adv_weight * PM.hinge_g_loss(fake_o, 1.0)                                                    # :1492-1500
    res.update(loss_cm=loss_cm, g_loss=g_loss, loss=loss_cm + g_loss)
    return res
