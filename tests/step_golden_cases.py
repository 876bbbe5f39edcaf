"""TEST INFRASTRUCTURE: the ORACLE side of the full-size parity tests, as ``ref_*`` builders that run on the CPU only (minutes each) and return
what the tests compare -- small tensors as they are, 67 M / 664 M-element vectors as count-sketches (tests/golden_fixture.py).
tests/golden/make_golden_step.py evaluates every builder in the build container and commits the results; the ``-m gpu`` tests obtain the
same dictionaries through ``golden_fixture.golden(name, builder)``, i.e. from the committed file, and only run the HIP path on the GPU box.

Everything random here comes from seeded CPU generators (weights: oracle/unet_sd15.init_state_dict or *_spec.random_state_dict on "cpu";
LoRA: LoraState's CPU generator; inputs: oracle/pcm_step.draw_inputs), so the GPU test reconstructs identical operands."""
import contextlib
import copy
import math
import time

import torch

from golden_fixture import golden, sk_cos, sk_rel, sketch, sketch_cat

SD15_KW = dict(block_out_channels=(320, 640, 1280, 1280), cross_attention_dim=768, heads=8, norm_num_groups=32)
KEYS7 = ("noise_pred", "cond_teacher_output", "uncond_teacher_output", "x_prev", "target_noise_pred", "model_pred", "target")
KEYS6 = ("noise_pred", "cond_teacher_output", "target_noise_pred", "x_prev", "model_pred", "target")
TS = ("start_timesteps", "timesteps", "end_timesteps")


@contextlib.contextmanager
def cpu_capi():
    """LoraState / Discriminator objects on "cpu" pack their operands through the C ABI: the host-emulation build of the same sources"""
    from emu_lib import emu_lib
    from pcm_amd import capi
    prev = capi._LIB
    capi.set_lib(emu_lib())
    try:
        yield
    finally:
        capi.set_lib(prev)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def lora_flat(lora, which):
    """this build's flat LoRA buffers in the oracle's (peft) order and layout, on the CPU: which = "p" (parameters) or "g" (gradients)"""
    out = []
    for m in lora.modules.values():
        a, b = (m.A, m.B) if which == "p" else (m.gA, m.gB)
        out += [lora.to_peft(m, a).detach().cpu().reshape(-1), b.detach().cpu().reshape(-1)]
    return torch.cat(out)


def olora_of(lora):
    return {p: (lora.A_peft(m).detach().cpu().clone(), m.B.detach().cpu().clone()) for p, m in lora.modules.items()}


def _sd15():
    from oracle import unet_sd15 as O
    oc = O.UNetConfig.sd15()
    return oc, O.init_state_dict(oc, 0)


def _cpu_lora(pc, seed, b_std):
    from pcm_amd.model import LoraState
    with cpu_capi():
        return LoraState(pc, 64, 8.0, "cpu", seed=seed, b_std=b_std)


def step_cfgs(multiphase, lr=5e-6, wd=1e-3):
    from oracle import pcm_step as OS
    from pcm_amd.trainer import StepConfig
    return (OS.StepConfig(multiphase=multiphase, loss_type="huber", lr=lr, adam_weight_decay=wd, w_min=4.0, w_max=5.0),
            StepConfig(multiphase=multiphase, loss_type="huber", learning_rate=lr, adam_weight_decay=wd, w_min=4.0, w_max=5.0))


# ---------------------------------------------------------------------------------------------------------------------------------
# SD1.5, bs 2, 2 phases: one whole step of the fp32 oracle (tests/test_gpu_step.py)
# ---------------------------------------------------------------------------------------------------------------------------------
def step_name(b_std):
    return "sd15_m2_bs2_bstd%g" % b_std


def step_inputs(B=2, multiphase=2, seed=453645634, index=(13, 37)):
    from oracle import pcm_step as OS
    ocfg, _ = step_cfgs(multiphase)
    inp = OS.draw_inputs(B, ocfg, seed=seed)
    if index is not None:
        inp["index"] = torch.tensor(list(index))
    return inp


def ref_sd15_step(b_std):
    """fp32 oracle step (train_pcm_lora_sd15.py:1139-1301): forward tensors, loss, un-clipped LoRA gradients, AdamW update (sketches)"""
    from oracle import pcm_step as OS
    from pcm_amd.unet_spec import UNetConfig
    oc, sd = _sd15()
    ocfg, _ = step_cfgs(2)
    inp = step_inputs()
    lora = _cpu_lora(UNetConfig.sd15(), 1, b_std)
    olora = olora_of(lora)
    p_before = lora_flat(lora, "p")
    t0 = time.time()
    ref = OS.distill_step(oc, sd, olora, inp, ocfg, {}, 1)
    print("oracle step %.1f s" % (time.time() - t0))
    out = {k: ref[k] for k in TS + ("noisy_model_input",) + KEYS7}
    out["loss"], out["grad_norm"] = float(ref["loss"]), float(ref["grad_norm"])
    coef = min(1.0, 1.0 / (out["grad_norm"] + 1e-6))
    out["sk_grad"] = sketch_cat(ref["grads"]) / coef                       # the oracle's gradients are post-clip: undo
    p_after = torch.cat([t.reshape(-1) for ab in olora.values() for t in ab])
    out["sk_param_before"], out["sk_param_after"], out["sk_update"] = sketch(p_before), sketch(p_after), sketch(p_after - p_before)
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# SD1.5, bs 2, 2 phases: rounding-point-matched oracle in fp32 and fp64 arithmetic (tests/test_gpu_rounding_matched.py)
# ---------------------------------------------------------------------------------------------------------------------------------
def ref_sd15_matched():
    from oracle import pcm_step as OS
    from pcm_amd.unet_spec import UNetConfig
    oc, sd = _sd15()
    ocfg, _ = step_cfgs(2)
    inp = step_inputs()
    olora = olora_of(_cpu_lora(UNetConfig.sd15(), 1, 0.02))
    out = {}
    with torch.no_grad():
        for tag, kw in (("m32", dict(storage="bf16")), ("m64", dict(storage="bf16", compute=torch.float64))):
            t0 = time.time()
            r = OS.distill_step_forward(oc, sd, olora, inp, ocfg, **kw)
            print("matched oracle", tag, "%.1f s" % (time.time() - t0))
            for k in KEYS6:
                out[tag + "." + k] = r[k].float() if tag == "m32" else r[k].double()
            out[tag + ".loss"] = float(r["loss"])
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# BASELINE configs[1] AS BENCHMARKED: SD1.5, 4 phases, bs 16 (train_pcm_lora_sd15.sh:12-17; sd15.py:1157-1174)
# ---------------------------------------------------------------------------------------------------------------------------------
C2_B, C2_PHASES, C2_SEED = 16, 4, 453645634


def c2_inputs():
    return step_inputs(B=C2_B, multiphase=C2_PHASES, seed=C2_SEED, index=None)


def ref_c2():
    """fp32 oracle (with LoRA gradients, accumulated over 8 micro-batches of 2: the loss is a batch mean, sd15.py:1288-1293, and the
    oracle's explicit softmax would need > 60 GB of autograd state at bs 16) and the rounding-point-matched oracle (forward)"""
    from oracle import pcm_step as OS
    from pcm_amd.unet_spec import UNetConfig
    oc, sd = _sd15()
    ocfg, _ = step_cfgs(C2_PHASES)
    inp = c2_inputs()
    lora = _cpu_lora(UNetConfig.sd15(), 1, 0.02)
    olora = olora_of(lora)
    out = {k: [] for k in TS + ("noisy_model_input",) + KEYS7}
    mout = {k: [] for k in KEYS6}
    losses, mlosses, gacc, gacc_m = [], [], None, None
    for i in range(0, C2_B, 2):
        sub = {k: v[i:i + 2] for k, v in inp.items()}
        t0 = time.time()
        leaves, lrg = [], {}
        for k, (a, b) in olora.items():
            a, b = a.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
            lrg[k] = (a, b)
            leaves += [a, b]
        r = OS.distill_step_forward(oc, sd, lrg, sub, ocfg)
        grads = torch.autograd.grad(r["loss"], leaves, allow_unused=True)
        flat = torch.cat([(torch.zeros_like(l) if g is None else g).reshape(-1) for g, l in zip(grads, leaves)])
        gacc = flat if gacc is None else gacc + flat
        for k in out:
            out[k].append(r[k].detach())
        losses.append(float(r["loss"]))
        leaves_m, lrg_m = [], {}
        for k, (a, b) in olora.items():
            a, b = a.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
            lrg_m[k] = (a, b)
            leaves_m += [a, b]
        m = OS.distill_step_forward(oc, sd, lrg_m, sub, ocfg, storage="bf16")       # the matched oracle rounds cotangents of bf16-stored tensors too
        gm = torch.autograd.grad(m["loss"], leaves_m, allow_unused=True)
        flat_m = torch.cat([(torch.zeros_like(l) if g is None else g).reshape(-1) for g, l in zip(gm, leaves_m)])
        gacc_m = flat_m if gacc_m is None else gacc_m + flat_m
        for k in KEYS6:
            mout[k].append(m[k].detach().float())
        mlosses.append(float(m["loss"]))
        print("c2 micro-batch %d: %.1f s" % (i // 2, time.time() - t0), flush=True)
    res = {k: torch.cat(v) for k, v in out.items()}
    res["noisy_model_input"] = res["noisy_model_input"][:4].clone()      # (bit-level check of add_noise: four samples are enough; fixture size)
    res.update({"m32." + k: torch.cat(v) for k, v in mout.items()})
    n = C2_B // 2
    g = gacc / n
    res["loss"], res["m32.loss"] = sum(losses) / n, sum(mlosses) / n
    res["grad_norm"] = float(g.double().norm())
    res["sk_grad"] = sketch(g)
    gm_ = gacc_m / n
    res["sk_grad_m32"] = sketch(gm_)
    res["grad_m32_vs_fp32"] = float((gm_.double() - g.double()).norm() / g.double().norm())
    res["sk_param_before"] = sketch(lora_flat(lora, "p"))
    return res


# ---------------------------------------------------------------------------------------------------------------------------------
# 20-step loss curve at the real SD1.5 size (sd15.py:1283-1301): bs 2, the recipe's learning rate, B = 0 init as the reference starts
# ---------------------------------------------------------------------------------------------------------------------------------
CURVE_STEPS, CURVE_FLOOR_STEPS = 20, (1, 10, 20)


def curve_inputs(step):
    from oracle import pcm_step as OS
    ocfg, _ = step_cfgs(2)
    return OS.draw_inputs(2, ocfg, seed=2000 + step)


def ref_curve20():
    """per step: fp32 oracle step (its own AdamW trajectory), the matched oracle's and the reference-style bf16-autocast loss on the same
    parameters; the matched oracle's fp64-arithmetic floor on 3 of the 20 steps"""
    from oracle import pcm_step as OS
    from pcm_amd.unet_spec import UNetConfig
    oc, sd = _sd15()
    ocfg, _ = step_cfgs(2)
    lora = _cpu_lora(UNetConfig.sd15(), 1, 0.0)
    olora = olora_of(lora)
    state = {}
    cols = dict(fp32=[], matched=[], bf16_autocast=[], floor=[], timesteps=[], end_timesteps=[])
    for step in range(1, CURVE_STEPS + 1):
        t0 = time.time()
        inp = curve_inputs(step)
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            l16 = float(OS.distill_step_forward(oc, sd, olora, inp, ocfg)["loss"])
        with torch.no_grad():
            lm = float(OS.distill_step_forward(oc, sd, olora, inp, ocfg, storage="bf16")["loss"])
            lm64 = float(OS.distill_step_forward(oc, sd, olora, inp, ocfg, storage="bf16", compute=torch.float64)["loss"]) if step in CURVE_FLOOR_STEPS else lm
        ref = OS.distill_step(oc, sd, olora, inp, ocfg, state, step)
        cols["fp32"].append(float(ref["loss"])); cols["matched"].append(lm); cols["bf16_autocast"].append(l16)
        cols["floor"].append(abs(lm64 - lm) / lm)
        cols["timesteps"].append(ref["timesteps"].tolist()); cols["end_timesteps"].append(ref["end_timesteps"].tolist())
        print("curve step %2d: fp32 %.6f matched %.6f bf16 %.6f  (%.1f s)" % (step, cols["fp32"][-1], lm, l16, time.time() - t0), flush=True)
    p_after = torch.cat([t.reshape(-1) for ab in olora.values() for t in ab])
    cols["sk_param_after"] = sketch(p_after)
    cols["sk_update"] = sketch(p_after - lora_flat(lora, "p"))
    return cols


# ---------------------------------------------------------------------------------------------------------------------------------
# SDXL / SD3-medium at their real sizes: ONE sample, teacher and LoRA-student forward (tests/test_gpu_zy_*, test_gpu_zz_*)
# ---------------------------------------------------------------------------------------------------------------------------------
def sdxl_one_sample_inputs():
    g = torch.Generator().manual_seed(7)
    x, t, ctx = torch.randn(1, 4, 128, 128, generator=g), torch.tensor([759]), torch.randn(1, 77, 2048, generator=g)
    ac = dict(text_embeds=torch.randn(1, 1280, generator=g), time_ids=torch.tensor([[1024, 1024, 0, 0, 1024, 1024]]))
    return x, t, ctx, ac


def ref_sdxl_one_sample():
    from oracle import unet_sd15 as O
    from pcm_amd.unet_spec import UNetConfig, random_state_dict
    cfg = UNetConfig.sdxl()
    sd = random_state_dict(cfg, 0, "cpu")
    x, t, ctx, ac = sdxl_one_sample_inputs()
    olora = olora_of(_cpu_lora(cfg, 3, 0.02))
    oc = O.UNetConfig.sdxl()
    t0 = time.time()
    with torch.no_grad():
        ref_t = O.unet_forward(oc, sd, x, t, ctx, added_cond=ac)
        ref_s = O.unet_forward(oc, sd, x, t, ctx, olora, 8.0, added_cond=ac)
    return dict(teacher=ref_t, student=ref_s, oracle_seconds=time.time() - t0)


def sd3_one_sample_inputs():
    g = torch.Generator().manual_seed(7)
    x, t = torch.randn(1, 16, 128, 128, generator=g), torch.tensor([640.5])
    ctx, pooled = torch.randn(1, 154, 4096, generator=g), torch.randn(1, 2048, generator=g)
    return x, t, ctx, pooled


def ref_sd3_one_sample():
    from oracle import mmdit_sd3 as O
    from pcm_amd.mmdit import sd3_lora_state
    from pcm_amd.mmdit_spec import MMDiTConfig, random_state_dict
    cfg = MMDiTConfig.sd3_medium()
    sd = random_state_dict(cfg, 0, "cpu")
    x, t, ctx, pooled = sd3_one_sample_inputs()
    with cpu_capi():
        lora = sd3_lora_state(cfg, 32, 8.0, "cpu", seed=3, b_std=0.05)
    olora = {p: (m.A[:32].detach().cpu().clone(), m.B[:, :32].detach().cpu().clone()) for p, m in lora.modules.items()}
    oc = O.MMDiTConfig.sd3_medium()
    t0 = time.time()
    with torch.no_grad():
        ref_t = O.mmdit_forward(oc, sd, x, t, ctx, pooled)
        ref_s = O.mmdit_forward(oc, sd, x, t, ctx, pooled, olora, 8.0)
    return dict(teacher=ref_t, student=ref_s, oracle_seconds=time.time() - t0)


# ---------------------------------------------------------------------------------------------------------------------------------
# SDXL / SD3-medium at their real sizes: ONE WHOLE distillation step of one sample -- forward tensors, loss, LoRA gradients, AdamW
# update (train_pcm_lora_sdxl_adv.py:1358-1480 consistency branch; train_pcm_lora_sd3.py:1270-1390)
# ---------------------------------------------------------------------------------------------------------------------------------
SDXL_STEP_KEYS = ("noise_pred", "cond_teacher_output", "uncond_teacher_output", "x_prev", "target_noise_pred", "model_pred", "target")
SD3_STEP_KEYS = ("model_output", "cond_teacher_output", "uncond_teacher_output", "x_prev", "target_pred", "model_pred", "target")


def sdxl_step_cfgs():
    """the SDXL recipe's step: 4 phases over 40 DDIM steps, w in [6, 7], huber, lr 2e-6 (train_pcm_lora_sdxl_adv.sh)"""
    from oracle import pcm_step as OS
    from pcm_amd.trainer import StepConfig
    kw = dict(multiphase=4, loss_type="huber", w_min=6.0, w_max=7.0, num_ddim_timesteps=40, adam_weight_decay=0.0)
    return OS.StepConfig(lr=2e-6, **kw), StepConfig(learning_rate=2e-6, **kw)


def sdxl_step_inputs():
    from oracle import pcm_step as OS
    ocfg, _ = sdxl_step_cfgs()
    inp = OS.draw_inputs(1, ocfg, seed=31, latent_hw=128, ctx_len=77, ctx_dim=2048)
    inp["index"] = torch.tensor([23])
    g = torch.Generator().manual_seed(32)
    tids = torch.tensor([[1024, 1024, 0, 0, 1024, 1024]])
    inp["added_cond"] = dict(text_embeds=torch.randn(1, 1280, generator=g), time_ids=tids)
    inp["uncond_added_cond"] = dict(text_embeds=torch.zeros(1, 1280), time_ids=tids)          # sdxl_adv.py:1216-1221
    return inp


def ref_sdxl_step_fullsize():
    from oracle import pcm_step as OS
    from oracle import unet_sd15 as O
    from pcm_amd.unet_spec import UNetConfig, random_state_dict
    cfg = UNetConfig.sdxl()
    sd = random_state_dict(cfg, 0, "cpu")
    ocfg, _ = sdxl_step_cfgs()
    inp = sdxl_step_inputs()
    lora = _cpu_lora(cfg, 3, 0.02)
    olora = olora_of(lora)
    p_before = lora_flat(lora, "p")
    t0 = time.time()
    ref = OS.distill_step(O.UNetConfig.sdxl(), sd, olora, inp, ocfg, {}, 1)
    secs = time.time() - t0
    print("SDXL full-size oracle step %.1f s" % secs, flush=True)
    out = {k: ref[k] for k in TS + ("noisy_model_input",) + SDXL_STEP_KEYS}
    out["loss"], out["grad_norm"], out["oracle_seconds"] = float(ref["loss"]), float(ref["grad_norm"]), secs
    coef = min(1.0, 1.0 / (out["grad_norm"] + 1e-6))
    out["sk_grad"] = sketch_cat(ref["grads"]) / coef                       # the oracle's gradients are post-clip: undo
    p_after = torch.cat([t.reshape(-1) for ab in olora.values() for t in ab])
    out["sk_param_before"], out["sk_param_after"], out["sk_update"] = sketch(p_before), sketch(p_after), sketch(p_after - p_before)
    return out


def sd3_step_inputs():
    g = torch.Generator().manual_seed(41)
    x0, noise = torch.randn(1, 16, 128, 128, generator=g), torch.randn(1, 16, 128, 128, generator=g)
    pe, upe = torch.randn(1, 154, 4096, generator=g), torch.randn(1, 154, 4096, generator=g)
    pp, upp = torch.randn(1, 2048, generator=g), torch.randn(1, 2048, generator=g)
    return x0, pe, pp, upe, upp, noise, torch.tensor([29])


def sd3_lora_flat(lora, which, rank=32):
    """flat LoRA parameters / gradients of the MMDiT in the oracle's order (A [r, K], B [N, r] of the live ranks), on the CPU"""
    out = []
    for m in lora.modules.values():
        a, b = (m.A, m.B) if which == "p" else (m.gA, m.gB)
        out += [a[:rank].detach().cpu().reshape(-1), b[:, :rank].detach().cpu().reshape(-1)]
    return torch.cat(out)


def ref_sd3_step_fullsize():
    """SD3-medium, 2-step deterministic recipe (BASELINE configs[4]: multiphase 2), one sample: forward tensors, loss, LoRA gradients and the
    AdamW update (torch.optim.AdamW semantics: oracle/pcm_step.adamw_step, lr 5e-6, weight decay 1e-2, clip 1.0)"""
    from oracle import mmdit_sd3 as O
    from oracle import pcm_step as OSD
    from oracle import pcm_step_sd3 as OS
    from pcm_amd.mmdit import sd3_lora_state
    from pcm_amd.mmdit_spec import MMDiTConfig, random_state_dict
    cfg = MMDiTConfig.sd3_medium()
    sd = random_state_dict(cfg, 0, "cpu")
    with cpu_capi():
        lora = sd3_lora_state(cfg, 32, 8.0, "cpu", seed=3, b_std=0.05)
    olora = {p: (m.A[:32].detach().cpu().clone().requires_grad_(True), m.B[:, :32].detach().cpu().clone().requires_grad_(True))
             for p, m in lora.modules.items()}
    p_before = sd3_lora_flat(lora, "p")
    a = sd3_step_inputs()
    t0 = time.time()
    ref = OS.distill_step_sd3(O.MMDiTConfig.sd3_medium(), sd, olora, *a, multiphase=2)
    ref["loss"].backward()
    secs = time.time() - t0
    print("SD3-medium full-size oracle step %.1f s" % secs, flush=True)
    leaves = [t for ab in olora.values() for t in ab]
    grads = [torch.zeros_like(l) if l.grad is None else l.grad.detach().clone() for l in leaves]
    out = {k: ref[k].detach() for k in ("noisy_model_input",) + SD3_STEP_KEYS}
    out["end_index"] = ref["end_index"]
    out["loss"], out["oracle_seconds"] = float(ref["loss"].detach()), secs
    out["sk_grad"] = sketch_cat(grads)
    out["grad_norm"] = float(torch.cat([g.reshape(-1) for g in grads]).double().norm())
    scfg = OSD.StepConfig(lr=5e-6, adam_weight_decay=1e-2)
    OSD.clip_grad_norm_(grads, scfg.max_grad_norm)
    params = [l.detach() for l in leaves]
    with torch.no_grad():
        OSD.adamw_step(params, grads, {}, 1, scfg)
    p_after = torch.cat([p.reshape(-1) for p in params])
    out["sk_param_before"], out["sk_param_after"], out["sk_update"] = sketch(p_before), sketch(p_after), sketch(p_after - p_before)
    return out
