"""Adversarial (BASELINE configs[2]) step parity case shared by the emulator and the GPU test files: the reference's configuration shape --
FOUR heads per tapped feature (discriminator_sd15.py:371-393), a batch of 2, NON-ZERO learning rates -- so that what is compared
includes the head / LoRA *updates* (clip + AdamW) and the sum of the four heads' gradients into one feature.
Reference: train_pcm_lora_sd15_adv.py:1375-1431 (D step :1375-1397, G step :1399-1431), optimizer_discriminator :1026-1032."""
import copy
import math

import torch


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def cos(a, b):
    a, b = a.double().cpu().reshape(-1), b.double().cpu().reshape(-1)
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-300))


def head_flat(disc, which, cpu=True):
    """this build's head tensors (parameters ``p`` or gradients ``g``) in the reference's names / layouts, as {name: tensor}
    (``cpu=False``: clones on the heads' own device -- the full-size comparison sketches 664 M elements there)"""
    out, cnt = {}, {}
    for k, hd in disc.heads:
        h = cnt.get(k, 0)
        cnt[k] = h + 1
        for n, t in (hd.p if which == "p" else hd.g).items():
            v = t.detach()
            if n in ("conv1.0.weight", "conv2.0.weight"):
                v = v.permute(0, 3, 1, 2) if disc.ksize == 3 else v.view(hd.C, hd.C, 1, 1)
            elif n == "conv_out.weight":
                v = v.view(1, hd.C, 1, 1)
            out[f"heads.{k}.{h}.{n}"] = v.cpu().clone() if cpu else v.contiguous().clone()
    return out


def _setup(dev, kw, dims, B, hw, ctx_len, ctx_dim, nh, lr, index, seed):
    """seeded CPU weights / LoRA / heads / inputs: identical on the fixture-writing host and on the GPU box"""
    from oracle import pcm_step as OS
    from oracle import unet_sd15 as O
    from pcm_amd.discriminator import Discriminator
    from pcm_amd.model import LoraState
    from pcm_amd.trainer import StepConfig
    from pcm_amd.unet_spec import UNetConfig
    oc, pc = O.UNetConfig(**kw), UNetConfig(**kw)
    lora = LoraState(pc, 64, 8.0, dev, seed=1, b_std=0.02)
    disc = Discriminator(dims, num_h_per_head=nh, device=dev, seed=2)
    assert len(disc.heads) == len(dims) * nh
    ocfg = OS.StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0, lr=lr, adam_weight_decay=1e-2)
    cfg = StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0, learning_rate=lr, adam_weight_decay=1e-2)
    inp = OS.draw_inputs(B, ocfg, seed=seed, latent_hw=hw, ctx_len=ctx_len, ctx_dim=ctx_dim)
    if index is not None:
        inp["index"] = torch.tensor(index)
    g = torch.Generator().manual_seed(9)
    inp["noise_fake"], inp["noise_real"] = torch.randn(B, 4, hw, hw, generator=g), torch.randn(B, 4, hw, hw, generator=g)
    inp["adv_u"] = torch.rand(B, generator=g)
    return oc, pc, lora, disc, ocfg, cfg, inp


def ref_adv_c3(kw, dims, B, hw, ctx_len, ctx_dim, global_step, nh=4, lr=5e-6, adv_lr=1e-5, adv_weight=0.1, index=None, seed=11):
    """ORACLE side (CPU only): one D step (even) or G step (odd) incl. the oracle's clip + AdamW on its own gradients.  Big vectors are
    count-sketched (tests/golden_fixture.py); the dictionary is what tests/golden/make_golden_step.py commits for the full-size case."""
    from golden_fixture import sketch, sketch_cat
    from oracle import pcm_step as OS
    from oracle import unet_sd15 as O
    from step_golden_cases import cpu_capi, olora_of
    with cpu_capi():
        oc, pc, lora, disc, ocfg, cfg, inp = _setup("cpu", kw, dims, B, hw, ctx_len, ctx_dim, nh, lr, index, seed)
    sd = O.init_state_dict(oc, 0)
    olora = olora_of(lora)
    dsd = {k: v.cpu() for k, v in disc.state_dict().items()}
    del disc
    ref = OS.distill_step_adv(oc, sd, olora, dsd, inp, ocfg, global_step, adv_weight=adv_weight)
    out = dict(adv_timesteps=ref["adv_timesteps"], fake_adv=ref["fake_adv"])
    if global_step % 2 == 0:
        names = list(ref["head_grads"])
        out["d_loss"] = float(ref["d_loss"])
        out["sk_head_grad"] = sketch_cat([ref["head_grads"][n] for n in names])
        for k in range(len(dims)):
            out["sk_head_grad_tap%d" % k] = sketch_cat([ref["head_grads"][n] for n in names if n.startswith("heads.%d." % k)])
        # the oracle's update: optimizer_discriminator = AdamW(lr=adv_lr, betas=(0, 0.999)) after the global-norm clip (:1026-1032, :1392-1396)
        grads = [ref["head_grads"][n].clone() for n in names]
        out["head_grad_norm"] = float(OS.clip_grad_norm_(grads, ocfg.max_grad_norm))
        params = [dsd[n].clone() for n in names]
        dcfg = copy.copy(ocfg)
        dcfg.lr, dcfg.adam_beta1 = adv_lr, 0.0
        OS.adamw_step(params, grads, {}, 1, dcfg)
        out["sk_head_update"] = sketch_cat([p_ - dsd[n] for p_, n in zip(params, names)])
        out["sk_head_param_after"] = sketch_cat(params)
    else:
        out["loss_cm"], out["g_loss"] = float(ref["loss_cm"]), float(ref["g_loss"])
        out["sk_lora_grad"] = sketch_cat(ref["lora_grads"])
        grads = [g_.clone() for g_ in ref["lora_grads"]]
        out["lora_grad_norm"] = float(OS.clip_grad_norm_(grads, ocfg.max_grad_norm))
        params = [t.clone() for ab in olora.values() for t in ab]
        p0 = torch.cat([t.reshape(-1) for t in params])
        OS.adamw_step(params, grads, {}, 1, ocfg)
        p1 = torch.cat([t.reshape(-1) for t in params])
        out["sk_lora_update"], out["sk_lora_param_after"] = sketch(p1 - p0), sketch(p1)
    return out


def case_adv_c3(dev, kw, dims, B, hw, ctx_len, ctx_dim, global_step, nh=4, lr=5e-6, adv_lr=1e-5, adv_weight=0.1, index=None, seed=11, golden_name=None):
    """One D step (even) or G step (odd) with real learning rates: the HIP path on ``dev`` against the oracle -- evaluated live
    (``golden_name`` None: the emulator's narrow case) or loaded from the committed fixture.  Returns the report; the caller asserts."""
    from golden_fixture import golden, sk_cos, sk_rel, sketch, sketch_cat
    from oracle import unet_sd15 as O
    from pcm_amd.model import UNetWeights
    from pcm_amd.trainer import AdvDistiller
    ref = golden(golden_name, lambda: ref_adv_c3(kw, dims, B, hw, ctx_len, ctx_dim, global_step, nh, lr, adv_lr, adv_weight, index, seed))
    oc, pc, lora, disc, ocfg, cfg, inp = _setup(dev, kw, dims, B, hw, ctx_len, ctx_dim, nh, lr, index, seed)
    W = UNetWeights(pc, O.init_state_dict(oc, 0), dev)
    D = AdvDistiller(W, lora, cfg, disc, adv_weight=adv_weight, adv_lr=adv_lr)
    gsc = float(D.loss_scale_dev.item()) if D.loss_scale_dev is not None else 1.0      # half build: the gradient buffers hold S * grad
    lora_p0, head_p0 = lora.params.clone(), head_flat(disc, "p", cpu=False)
    p0 = torch.cat([torch.cat([lora.A_peft(m).detach().cpu().reshape(-1), m.B.detach().cpu().reshape(-1)]) for m in lora.modules.values()])
    names = list(head_p0)
    dvc = {k: v.to(dev) for k, v in inp.items()}
    out = D.step_adv(global_step, dvc["latents"], dvc["prompt_embeds"], dvc["uncond_prompt_embeds"], dvc["noise"], dvc["index"], dvc["w"],
                     dvc["noise_fake"], dvc["noise_real"], dvc["adv_u"])
    assert torch.equal(out["adv_timesteps"].cpu(), ref["adv_timesteps"])
    rep = {"fake_adv": rel(out["fake_adv"], ref["fake_adv"]), "heads": len(disc.heads), "B": B}
    if global_step % 2 == 0:
        rep["d_loss"], rep["d_loss_oracle"] = float(out["d_loss"]), float(ref["d_loss"])
        rep["d_loss_rel"] = abs(rep["d_loss"] - rep["d_loss_oracle"]) / abs(rep["d_loss_oracle"])
        mine_g = head_flat(disc, "g", cpu=False)
        mg = sketch_cat([mine_g[n] for n in names]) / gsc
        rep["head_grad_rel"], rep["head_grad_cos"] = sk_rel(mg, ref["sk_head_grad"]), sk_cos(mg, ref["sk_head_grad"])
        # per tapped feature (the 4 heads of one feature share its bucket)
        rep["head_grad_cos_per_tap"] = [sk_cos(sketch_cat([mine_g[n] for n in names if n.startswith("heads.%d." % k)]), ref["sk_head_grad_tap%d" % k])
                                        for k in range(len(dims))]
        del mine_g
        mine_p = head_flat(disc, "p", cpu=False)
        up_m = sketch_cat([mine_p[n] - head_p0[n] for n in names])
        gn = float(ref["head_grad_norm"])
        rep["head_grad_norm_rel"] = abs(math.sqrt(float(disc.gradsq.item())) / gsc - gn) / gn
        rep["head_update_cos"] = sk_cos(up_m, ref["sk_head_update"])
        rep["head_update_norm_ratio"] = float(up_m.norm() / ref["sk_head_update"].double().norm())
        rep["head_param_rel_after"] = sk_rel(sketch_cat([mine_p[n] for n in names]), ref["sk_head_param_after"])
        rep["lora_untouched"] = bool(torch.equal(lora.params, lora_p0))
    else:
        for k in ("loss_cm", "g_loss"):
            rep[k], rep[k + "_oracle"] = float(out[k]), float(ref[k])
            rep[k + "_rel"] = abs(rep[k] - rep[k + "_oracle"]) / abs(rep[k + "_oracle"])
        mg = sketch_cat([t for m in lora.modules.values() for t in (lora.gA_peft(m), m.gB)]) / gsc
        rep["lora_grad_rel"], rep["lora_grad_cos"] = sk_rel(mg, ref["sk_lora_grad"]), sk_cos(mg, ref["sk_lora_grad"])
        mine1 = torch.cat([torch.cat([lora.A_peft(m).detach().cpu().reshape(-1), m.B.detach().cpu().reshape(-1)]) for m in lora.modules.values()])
        gn = float(ref["lora_grad_norm"])
        rep["lora_grad_norm_rel"] = abs(math.sqrt(float(lora.gradsq.item())) / gsc - gn) / gn
        up_m = sketch(mine1 - p0)
        rep["lora_update_cos"] = sk_cos(up_m, ref["sk_lora_update"])
        rep["lora_update_norm_ratio"] = float(up_m.norm() / ref["sk_lora_update"].double().norm())
        rep["lora_param_rel_after"] = sk_rel(sketch(mine1), ref["sk_lora_param_after"])
        rep["heads_untouched"] = all(torch.equal(v, head_p0[n]) for n, v in head_flat(disc, "p", cpu=False).items())
    print({k: (("%.4g" % v) if isinstance(v, float) else v) for k, v in rep.items()})
    return rep
