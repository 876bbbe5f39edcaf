"""Host-emulation end-to-end check of the explicit UNet forward / LoRA-only backward schedule
(pcm_amd.model) against the oracle's autograd on a tiny config with the SD1.5 topology."""
import pytest
import torch

from emu_lib import emu_lib
from pcm_amd import capi


@pytest.fixture(autouse=True)
def _use_emu():
    capi.set_lib(emu_lib())
    yield
    capi.set_lib(None)


def tiny_cfgs():
    from oracle.unet_sd15 import UNetConfig as OC
    from pcm_amd.unet_spec import UNetConfig as PC
    # three levels keep every SD1.5 block type (2 x CrossAttnDownBlock2D + DownBlock2D, mid, UpBlock2D + 2 x CrossAttnUpBlock2D, down / up
    # samplers, 2 resnets per block) at a third less emulator time than four; the GPU suite runs the 4-level and the full-size configs
    kw = dict(block_out_channels=(64, 128, 128), cross_attention_dim=64, heads=2, norm_num_groups=32)
    return OC(**kw), PC(**kw)


def test_spec_matches_oracle():
    from oracle import unet_sd15 as O
    from pcm_amd import unet_spec as P
    for oc, pc in [(O.UNetConfig.sd15(), P.UNetConfig.sd15()), tiny_cfgs()]:
        assert O.param_spec(oc) == P.param_spec(pc)
        assert O.lora_target_modules(oc) == P.lora_target_modules(pc)
    oc, pc = tiny_cfgs()
    a, b = O.init_state_dict(oc, 3), P.random_state_dict(pc, 3)
    assert all(torch.equal(a[k], b[k]) for k in a)


@pytest.mark.slow
def test_unet_forward_backward_vs_oracle():
    from oracle import unet_sd15 as O
    from pcm_amd.model import LoraState, UNet, UNetWeights
    oc, pc = tiny_cfgs()
    sd = O.init_state_dict(oc, 0)
    B, Hh = 2, 8
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, 4, Hh, Hh, generator=g)
    t = torch.tensor([19, 759])
    ctx = torch.randn(B, 7, 64, generator=g)
    d_eps = torch.randn(B, 4, Hh, Hh, generator=g)
    W = UNetWeights(pc, sd, "cpu")
    lora = LoraState(pc, 64, 8.0, "cpu", seed=1, b_std=0.05)
    # oracle with the same LoRA values and bf16-rounded operands where the kernels round them
    olora = {p: (lora.A_peft(m).clone().requires_grad_(True), m.B.clone().requires_grad_(True)) for p, m in lora.modules.items()}
    ref_t = O.unet_forward(oc, sd, x, t, ctx)
    ref_s = O.unet_forward(oc, sd, x, t, ctx, olora, 8.0)
    teacher = UNet(W, None)
    out_t = teacher.forward(x, t, ctx)
    student = UNet(W, lora)
    out_s, tape = student.forward(x, t, ctx, save=True)
    # the self-attention q/k/v LoRA projections ran as the fused (concatenated / block-diagonal operand) schedule
    assert lora.qkv and all(sv["blk0"]["sa1"].get("fused") for kind, _, sv in tape if kind == "transformer")
    # ... and the feed-forward of the widest level (M = 128 rows here) took the fused-GEGLU projection that keeps the pre-activation
    from pcm_amd import model as Mdl
    assert not Mdl.FUSE_GEGLU_GRAD or any(sv["blk0"]["pre"] is not None and sv["blk0"]["hg"] is None for kind, _, sv in tape if kind == "transformer")
    scale = ref_t.abs().max().item()
    err_t = (out_t - ref_t).abs().max().item()
    err_s = (out_s - ref_s.detach()).abs().max().item()
    print("fwd err teacher %.3e student %.3e (scale %.3e), lora effect %.3e" % (err_t, err_s, scale, (ref_s - ref_t).abs().max().item()))
    assert err_t < 0.03 * scale and err_s < 0.03 * scale
    assert (ref_s - ref_t).abs().max().item() > 5 * err_s, "LoRA branch not exercised"
    (ref_s * d_eps).sum().backward()
    lora.zero_grad()
    student.backward(d_eps, tape)
    num = den = 0.0
    worst = 0.0
    for p, m in lora.modules.items():
        for got, ref in ((lora.gA_peft(m), olora[p][0].grad), (m.gB, olora[p][1].grad)):
            ref = ref.view_as(got)
            num += float(((got - ref) ** 2).sum())
            den += float((ref ** 2).sum())
            rel = float((got - ref).norm() / (ref.norm() + 1e-12))
            worst = max(worst, rel)
            assert rel < 0.15, (p, rel)
    print("grad rel err: global %.3e worst module %.3e" % ((num / den) ** 0.5, worst))
    assert (num / den) ** 0.5 < 0.05


def test_fused_online_target_pass_keeps_half_of_the_geglu_pre_activation():
    """forward(save=True, save_half=True) on [online; target] + tape_first_half == forward(save=True) on the online half alone:
    same eps for both halves, same LoRA gradients (the fused-GEGLU projection stores its pre-activation for the first half only)."""
    from oracle import unet_sd15 as O
    from pcm_amd import model as Mdl
    from pcm_amd.model import LoraState, UNet, UNetWeights
    oc, pc = tiny_cfgs()
    sd = O.init_state_dict(oc, 0)
    B, Hh = 2, 8
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2 * B, 4, Hh, Hh, generator=g)
    t = torch.tensor([19, 759, 300, 40])
    ctx = torch.randn(2 * B, 7, 64, generator=g)
    d_eps = torch.randn(B, 4, Hh, Hh, generator=g)
    W = UNetWeights(pc, sd, "cpu")
    lora = LoraState(pc, 64, 8.0, "cpu", seed=1, b_std=0.05)
    student = UNet(W, lora)
    out2, tape2 = student.forward(x, t, ctx, save=True, save_half=True)
    pres = [sv["blk0"]["pre"] for kind, _, sv in tape2 if kind == "transformer" and sv["blk0"]["pre"] is not None]
    assert pres and all(isinstance(p_, Mdl.HalfSaved) and p_.t.shape[0] * 2 == 2 * B * Hh * Hh for p_ in pres)
    lora.zero_grad()
    student.backward(d_eps, student.tape_first_half(tape2))
    g_half = lora.grads.clone()
    # (a) the online half run on its own: identical eps; gradients agree to the bf16 noise of the different GEMM plans (M halves)
    out1, tape1 = student.forward(x[:B], t[:B], ctx[:B], save=True)
    assert torch.equal(out1, out2[:B])
    lora.zero_grad()
    student.backward(d_eps, tape1)
    rel = float((g_half - lora.grads).norm() / lora.grads.norm())
    assert rel < 2e-2, rel
    # (b) the same fused batch with GEGLU as a separate pass (the pre-activation saved for all rows): same plans elsewhere
    Mdl.FUSE_GEGLU_GRAD = False
    try:
        out3, tape3 = student.forward(x, t, ctx, save=True, save_half=True)
        assert all(sv["blk0"]["pre"] is None and sv["blk0"]["hg"] is not None for kind, _, sv in tape3 if kind == "transformer")
        lora.zero_grad()
        student.backward(d_eps, student.tape_first_half(tape3))
    finally:
        Mdl.FUSE_GEGLU_GRAD = True
    scale = float(out3.abs().max())
    assert float((out3 - out2).abs().max()) < 0.02 * scale
    # two bf16 evaluations of the same gradient (each ~2.2 % from the fp32 oracle on this narrow config, see the test above)
    rel = float((g_half - lora.grads).norm() / lora.grads.norm())
    cos = float((g_half * lora.grads).sum() / (g_half.norm() * lora.grads.norm()))
    assert rel < 4e-2 and cos > 0.999, (rel, cos)


def test_sdxl_topology_forward_backward_vs_oracle():
    """the SDXL UNet wiring (no attention at level 0, transformer depth > 1, per-level head counts with 64-wide heads,
    Linear proj_in/proj_out, text_time added conditioning) on a narrow config; plus the full-size parameter count."""
    import math
    from oracle import unet_sd15 as O
    from pcm_amd.model import LoraState, UNet, UNetWeights
    from pcm_amd.unet_spec import UNetConfig, param_spec
    assert sum(math.prod(s) for _, s in param_spec(UNetConfig.sdxl())) == 2567463684
    assert [k for k, _ in param_spec(UNetConfig.sdxl())] == [k for k, _ in O.param_spec(O.UNetConfig.sdxl())]
    kw = dict(block_out_channels=(64, 128, 128), cross_attention_dim=64, heads=(1, 2, 2), down_attn=(False, True, True),
              transformer_depth=(1, 2, 2), use_linear_projection=True, addition_time_embed_dim=32, projection_class_embeddings_input_dim=64 + 6 * 32,
              layers_per_block=1)
    oc, pc = O.UNetConfig(**kw), UNetConfig(**kw)
    sd = O.init_state_dict(oc, 0)
    B, Hh = 2, 8
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, 4, Hh, Hh, generator=g)
    t = torch.tensor([19, 759])
    ctx = torch.randn(B, 7, 64, generator=g)
    added = dict(text_embeds=torch.randn(B, 64, generator=g), time_ids=torch.tensor([[1024, 1024, 0, 0, 1024, 1024], [512, 768, 16, 32, 1024, 1024]]))
    d_eps = torch.randn(B, 4, Hh, Hh, generator=g)
    W = UNetWeights(pc, sd, "cpu")
    lora = LoraState(pc, 64, 8.0, "cpu", seed=1, b_std=0.05)
    olora = {p: (lora.A_peft(m).clone().requires_grad_(True), m.B.clone().requires_grad_(True)) for p, m in lora.modules.items()}
    assert set(olora) == {p for p, _ in O.lora_target_modules(oc)}
    ref_t = O.unet_forward(oc, sd, x, t, ctx, added_cond=added)
    ref_s = O.unet_forward(oc, sd, x, t, ctx, olora, 8.0, added_cond=added)
    out_t = UNet(W, None).forward(x, t, ctx, added_cond=added)
    student = UNet(W, lora)
    out_s, tape = student.forward(x, t, ctx, save=True, added_cond=added)
    scale = ref_t.abs().max().item()
    err_t, err_s = (out_t - ref_t).abs().max().item(), (out_s - ref_s.detach()).abs().max().item()
    print("sdxl-topology fwd err teacher %.3e student %.3e (scale %.3e)" % (err_t, err_s, scale))
    assert err_t < 0.03 * scale and err_s < 0.03 * scale
    # the added conditioning must matter (otherwise the test would not see a wiring error)
    other = dict(added, time_ids=added["time_ids"] * 0 + 7)
    assert (O.unet_forward(oc, sd, x, t, ctx, added_cond=other) - ref_t).abs().max().item() > 10 * err_t
    (ref_s * d_eps).sum().backward()
    lora.zero_grad()
    student.backward(d_eps, tape)
    num = den = 0.0
    for p, m in lora.modules.items():
        for got, ref in ((lora.gA_peft(m), olora[p][0].grad), (m.gB, olora[p][1].grad)):
            ref = ref.view_as(got)
            num += float(((got - ref) ** 2).sum()); den += float((ref ** 2).sum())
    print("sdxl-topology LoRA grad rel err %.3e" % (num / den) ** 0.5)
    assert (num / den) ** 0.5 < 0.08


def test_sdxl_topology_distillation_step_vs_oracle():
    """One PCM distillation step with SDXL-style added conditioning (cond + uncond pooled embeds) through the Distiller."""
    from oracle import pcm_step as OS
    from oracle import unet_sd15 as O
    from pcm_amd.model import LoraState, UNetWeights
    from pcm_amd.trainer import Distiller, StepConfig
    from pcm_amd.unet_spec import UNetConfig
    kw = dict(block_out_channels=(64, 128), cross_attention_dim=64, heads=(1, 2), down_attn=(False, True), transformer_depth=(1, 2),
              use_linear_projection=True, addition_time_embed_dim=32, projection_class_embeddings_input_dim=64 + 6 * 32, layers_per_block=1)
    oc, pc = O.UNetConfig(**kw), UNetConfig(**kw)
    sd = O.init_state_dict(oc, 0)
    W = UNetWeights(pc, sd, "cpu")
    lora = LoraState(pc, 64, 8.0, "cpu", seed=1, b_std=0.05)
    olora = {p: (lora.A_peft(m).clone().requires_grad_(True), m.B.clone().requires_grad_(True)) for p, m in lora.modules.items()}
    ocfg = OS.StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0)
    B = 2
    inp = OS.draw_inputs(B, ocfg, seed=7, latent_hw=8, ctx_len=7, ctx_dim=64)
    g = torch.Generator().manual_seed(3)
    tids = torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * B)
    inp["added_cond"] = dict(text_embeds=torch.randn(B, 64, generator=g), time_ids=tids)
    inp["uncond_added_cond"] = dict(text_embeds=torch.zeros(B, 64), time_ids=tids)       # zero uncond pooled embeds (sdxl_adv.py:1216-1221)
    ref = OS.distill_step_forward(oc, sd, olora, inp, ocfg)
    cfg = StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0)
    D = Distiller(W, lora, cfg)
    out = D.forward_backward(inp["latents"], inp["prompt_embeds"], inp["uncond_prompt_embeds"], inp["noise"], inp["index"], inp["w"],
                             backward=False, added_cond=inp["added_cond"], uncond_added_cond=inp["uncond_added_cond"])
    for k in ("noise_pred", "uncond_teacher_output", "x_prev", "target"):
        r = ref[k].detach().float()
        rel = float((out[k].float() - r).norm() / r.norm())
        print(k, "%.3e" % rel)
        assert rel < 3e-2, (k, rel)
    assert abs(float(out["loss"]) - float(ref["loss"].detach())) < 5e-2 * abs(float(ref["loss"].detach()))


@pytest.mark.slow
def test_unet_non_square_ragged_resolution_vs_oracle():
    """latents that are neither square nor a power of two (12 x 20 -> 6 x 10 -> 3 x 5 at the lower levels, odd row lengths for the
    conv tap walkers, attention lengths 240 / 60 that are not multiples of the 64-key tile), batch 1."""
    from oracle import unet_sd15 as O
    from pcm_amd.model import LoraState, UNet, UNetWeights
    oc, pc = tiny_cfgs()
    sd = O.init_state_dict(oc, 0)
    W = UNetWeights(pc, sd, "cpu")
    lora = LoraState(pc, 64, 8.0, "cpu", seed=1, b_std=0.05)
    olora = {p: (lora.A_peft(m).clone(), m.B.clone()) for p, m in lora.modules.items()}
    g = torch.Generator().manual_seed(3)
    for B, (H, Wd) in ((1, (12, 20)),):
        x = torch.randn(B, 4, H, Wd, generator=g)
        t = torch.randint(0, 1000, (B,), generator=g)
        ctx = torch.randn(B, 9, 64, generator=g)
        with torch.no_grad():
            ref = O.unet_forward(oc, sd, x, t, ctx, olora, 8.0)
        out = UNet(W, lora).forward(x, t, ctx)
        err = (out - ref).abs().max().item()
        assert out.shape == ref.shape and err < 0.03 * ref.abs().max().item(), (B, H, Wd, err)


def test_unet_forward_same_through_both_attention_forward_kernels():
    """the whole student forward (self-attention at head dims 32 / 64, text cross-attention with a two-tile ragged key stream) through the
    software-pipelined attention forward equals the forward through the first kernel: the two kernels differ in schedule only"""
    from oracle import unet_sd15 as O
    from pcm_amd.model import LoraState, UNet, UNetWeights
    oc, pc = tiny_cfgs()
    sd = O.init_state_dict(oc, 0)
    g = torch.Generator().manual_seed(11)
    x, t, ctx = torch.randn(2, 4, 16, 16, generator=g), torch.tensor([19, 759]), torch.randn(2, 77, 64, generator=g)
    W = UNetWeights(pc, sd, "cpu")
    lora = LoraState(pc, 64, 8.0, "cpu", seed=1, b_std=0.05)
    dll = capi.lib().dll
    outs = []
    try:
        for var in (0, 1):
            dll.pcm_debug_attn_fwd_variant(var)
            outs.append(UNet(W, lora).forward(x, t, ctx))
    finally:
        dll.pcm_debug_attn_fwd_variant(-1)
    rel = float((outs[1] - outs[0]).norm() / outs[0].norm())
    assert rel < 1e-4, rel


def test_teacher_shared_prefix_equals_plain_2b_pass():
    """UNet.forward(dup_halves=True): the teacher's [cond; uncond] batch shares sample and timestep between its halves
    (train_pcm_lora_sd15.py:1217-1252); conv_in, the first resnet and the first self-attention are computed once and duplicated where the
    first cross-attention makes the halves differ -- per sample the same operations as the plain 2B pass.  Bitwise equal when the prefix's
    half-size GEMMs take the same tile / split-K plan (measured at B = 16); otherwise the two passes are two bf16 evaluations of one function and
    sit as far from each other as each sits from the fp32 oracle."""
    from oracle import unet_sd15 as O
    from pcm_amd.model import UNet, UNetWeights
    oc, pc = tiny_cfgs()
    sd = O.init_state_dict(oc, 0)
    W = UNetWeights(pc, sd, "cpu")
    g = torch.Generator().manual_seed(5)
    rel = lambda a, b: float((a - b).norm() / b.norm())   # noqa: E731
    for B in (2, 8):         # GroupNorm statistics: partials form / atomics arena (2B >= 16)
        x, t = torch.randn(B, 4, 16, 16, generator=g), torch.randint(0, 1000, (B,), generator=g)
        c, u = torch.randn(B, 9, 64, generator=g), torch.randn(B, 9, 64, generator=g)
        T = UNet(W, None)
        args = (torch.cat([x, x]), torch.cat([t, t]), torch.cat([c, u]))
        a, b = T.forward(*args), T.forward(*args, dup_halves=True)
        with torch.no_grad():
            ref = O.unet_forward(oc, sd, *args)
        ea, eb = rel(a, ref), rel(b, ref)
        assert eb <= 1.2 * ea + 1e-3 and rel(a, b) <= 1.5 * max(ea, eb), (B, ea, eb, rel(a, b))
        assert rel(a[:B], a[B:]) > 0.1, B        # the text conditioning is live


def test_deterministic_step_equals_the_atomic_step_to_rounding():
    """ops.set_deterministic(True) (slab / partial reductions, include/pcm_hip.h abi 4) through a whole Distiller step on the tiny SD1.5
    topology: same loss, same LoRA gradients and gradient norm to summation rounding as the atomic forms, and bitwise the same twice.
    (The GPU run of the same property at the real size is tests/test_gpu_step.py::test_deterministic_step_is_bitwise_reproducible.)"""
    from oracle import pcm_step as OS
    from oracle import unet_sd15 as O
    from pcm_amd import ops
    from pcm_amd.model import LoraState, UNetWeights
    from pcm_amd.trainer import Distiller, StepConfig
    oc, pc = tiny_cfgs()
    sd = O.init_state_dict(oc, 0)
    W = UNetWeights(pc, sd, "cpu")
    ocfg = OS.StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0)
    inp = OS.draw_inputs(2, ocfg, seed=7, latent_hw=8, ctx_len=7, ctx_dim=64)
    cfg = StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0)

    from pcm_amd import model as M_

    def one(det, gn_fuse=False):
        lora = LoraState(pc, 64, 8.0, "cpu", seed=1, b_std=0.05)
        D = Distiller(W, lora, cfg)
        ops.set_deterministic(det)
        keep, M_.FUSE_GN_STATS = M_.FUSE_GN_STATS, gn_fuse
        try:
            out = D.step(inp["latents"], inp["prompt_embeds"], inp["uncond_prompt_embeds"], inp["noise"], inp["index"], inp["w"])
        finally:
            ops.set_deterministic(False)
            M_.FUSE_GN_STATS = keep
        return float(out["loss"]), lora.grads.clone(), float(lora.gradsq), lora.params.clone()

    # the atomic forms of the SAME reductions (GroupNorm statistics by their own pass: PCM_GN_FUSE=0) against the reproducible forms
    l0, g0, q0, p0 = one(False, gn_fuse=False)
    l1, g1, q1, p1 = one(True)
    l2, g2, q2, p2 = one(True)
    assert l1 == l2 and q1 == q2 and torch.equal(g1, g2) and torch.equal(p1, p2)
    assert abs(l1 - l0) <= 1e-6 * abs(l0) and abs(q1 - q0) <= 1e-4 * q0
    assert float((g1 - g0).norm() / g0.norm()) < 1e-4
    # the opt-in path (PCM_GN_FUSE=1) takes the GroupNorm statistics from the producing contraction's epilogue (round 6): other fp32 partial sums
    # (closer to the exact sums than the statistics pass), so isolated 16-bit roundings of the normalised activations flip -- the step agrees
    # at the storage format's own noise level (on this narrow net: a few percent of the near-cancelling gradient), not to summation rounding
    l3, g3, q3, p3 = one(False, gn_fuse=True)
    assert abs(l3 - l0) <= 5e-3 * abs(l0) and float((g3 - g0).norm() / g0.norm()) < 0.15, (l3, l0, float((g3 - g0).norm() / g0.norm()))


def test_fp16_teacher_next_to_a_bf16_student_is_exactly_the_two_pure_builds():
    """Distiller(teacher_weights = the frozen weights packed in IEEE half): the reference's ODE-solver teacher pass runs under
    torch.autocast("cuda") with no dtype, i.e. in half, even when the student trains in bfloat16 (train_pcm_lora_sd15.py:1217-1218).  The
    split must be clean: the teacher outputs (and x_prev, a function of them and of fp32 inputs only) are BITWISE those of an all-half
    process, the student's noise prediction is BITWISE that of an all-bfloat16 process, and nothing of the process's format leaks out
    of the scope.  (GPU run at the real size: tests/test_gpu_step.py::test_fp16_teacher_bf16_student_split.)"""
    from oracle import pcm_step as OS
    from oracle import unet_sd15 as O
    from pcm_amd import ops, precision
    from pcm_amd.model import LoraState, UNetWeights
    from pcm_amd.trainer import Distiller, StepConfig
    oc, pc = tiny_cfgs()
    sd = O.init_state_dict(oc, 0)
    ocfg = OS.StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0)
    inp = OS.draw_inputs(2, ocfg, seed=7, latent_hw=8, ctx_len=7, ctx_dim=64)
    cfg = StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0)
    args = (inp["latents"], inp["prompt_embeds"], inp["uncond_prompt_embeds"], inp["noise"], inp["index"], inp["w"])
    keys = ("cond_teacher_output", "uncond_teacher_output", "x_prev", "noise_pred", "target_noise_pred", "loss")

    def run(teacher_weights=None):
        W = UNetWeights(pc, sd, "cpu")
        lora = LoraState(pc, 64, 8.0, "cpu", seed=1, b_std=0.05)
        out = Distiller(W, lora, cfg, teacher_weights=teacher_weights).forward_backward(*args)
        return {k: out[k].clone() for k in keys}, lora.grads.clone()

    precision.set_precision("bf16", lib=emu_lib("bf16"))
    precision.register_lib("fp16", emu_lib("f16"))
    try:
        pure_b, g_b = run()
        with precision.format_scope("fp16"):
            assert ops.BF16 == torch.float16 and capi.lib().act_dtype == 1
            Wt = UNetWeights(pc, sd, "cpu")
        assert Wt.format == "fp16" and ops.BF16 == torch.bfloat16 and capi.lib().act_dtype == 0 and precision.precision() == "bf16"
        mixed, g_m = run(Wt)
        assert ops.BF16 == torch.bfloat16 and capi.lib().act_dtype == 0
        precision.set_precision("fp16", lib=emu_lib("f16"))
        pure_h, _ = run()
    finally:
        precision.set_precision("bf16", lib=emu_lib("bf16"))
    for k in ("cond_teacher_output", "uncond_teacher_output", "x_prev"):
        assert torch.equal(mixed[k], pure_h[k]), k
        assert not torch.equal(mixed[k], pure_b[k]), k
    assert torch.equal(mixed["noise_pred"], pure_b["noise_pred"])
    # the target pass starts from the teacher's x_prev: it follows the half teacher, at bf16-student precision
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())   # noqa: E731
    assert 0 < rel(mixed["target_noise_pred"], pure_b["target_noise_pred"]) < 2e-2
    assert abs(float(mixed["loss"]) - float(pure_b["loss"])) < 5e-2 * abs(float(pure_b["loss"]))
    # (the gradient is proportional to model_pred - target, a small difference on this narrow net: it moves by tens of percent with the target)
    assert float((g_m.double() * g_b.double()).sum() / (g_m.double().norm() * g_b.double().norm())) > 0.9 and not torch.equal(g_m, g_b)


def test_cli_teacher_precision_fp16_end_to_end(tmp_path, monkeypatch):
    """train_pcm_lora_sd15.py --teacher_precision fp16 as a program (narrow UNet, host emulators of both builds): the flag packs the frozen
    weights a second time in half, the steps run with the student in bfloat16 (no loss scaler), and the logged losses differ from the
    one-format run's on the same seeds (the teacher pass really ran in the other build)."""
    import importlib.util
    import json
    import math
    import os
    from safetensors.torch import save_file
    from pcm_amd import ops, precision
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "phased-consistency-model_amd")
    spec = importlib.util.spec_from_file_location("pcm_cli_teacher_fp16", os.path.join(pkg, "train_pcm_lora_sd15.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    g = torch.Generator().manual_seed(0)
    shards = tmp_path / "s"
    shards.mkdir()
    save_file({"latents": torch.randn(6, 4, 8, 8, generator=g), "prompt_embeds": torch.randn(6, 7, 64, generator=g),
               "uncond_prompt_embeds": torch.randn(7, 64, generator=g)}, str(shards / "a.safetensors"))
    monkeypatch.setenv("PCM_CLI_DEVICE", "cpu")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    precision.set_precision("bf16", lib=emu_lib("bf16"))
    precision.register_lib("fp16", emu_lib("f16"))
    logs = {}
    for mode in ("same", "fp16", "default"):
        out = tmp_path / mode
        cli.main(cli.parse_args(["--pretrained_teacher_model", "random", "--tiny_model", "--latents_dir", str(shards), "--train_batch_size", "1",
                                 "--learning_rate", "1e-3", "--multiphase", "2", "--seed", "1", "--output_dir", str(out), "--loss_type", "huber",
                                 "--max_train_steps", "2"] + (["--teacher_precision", mode] if mode != "default" else [])))
        assert ops.BF16 == torch.bfloat16 and capi.lib().act_dtype == 0 and precision.precision() == "bf16"
        logs[mode] = [json.loads(l) for l in open(out / "logs" / "text2image-fine-tune.jsonl")]
        assert [r["step"] for r in logs[mode]] == [1, 2] and all(math.isfinite(r["loss"]) for r in logs[mode])
    a, b = logs["same"][0]["loss"], logs["fp16"][0]["loss"]
    assert a != b and abs(a - b) < 0.1 * abs(a), (a, b)
    # round 6: without the flag the CLI follows the reference (train_pcm_lora_sd15.py:1217-1218 -- dtype-less autocast = IEEE half for the teacher)
    assert [r["loss"] for r in logs["default"]] == [r["loss"] for r in logs["fp16"]]
    assert cli.parse_args(["--pretrained_teacher_model", "x"]).teacher_precision == "reference"


def test_batched_time_embedding_projections_equal_the_per_resnet_ones():
    """Every resnet's time_emb_proj reads the same [B, 1280] input, so a pass computes them all in ONE GEMM (plus one rank-64 down-projection
    GEMM and a block-diagonal second segment with LoRA; pcm_amd/model.py UNet._temb_all) and scatters the fp32 result into per-resnet
    contiguous 16-bit row vectors with one segmented-pack launch.  Against one GEMM per resnet (PCM_TEMB_BATCH=0): the same products summed
    in the same order (the zero blocks add exact zeros) and the same single rounding of the fp32 accumulator -- prediction, loss and LoRA
    gradients agree to summation-order rounding, for the frozen pass (no LoRA), the student pass and the shared-prefix teacher pass."""
    from oracle import pcm_step as OS
    from oracle import unet_sd15 as O
    from pcm_amd import model as M
    from pcm_amd.model import LoraState, UNet, UNetWeights
    from pcm_amd.trainer import Distiller, StepConfig
    oc, pc = tiny_cfgs()
    sd = O.init_state_dict(oc, 0)
    W = UNetWeights(pc, sd, "cpu")
    assert W.temb_cat is not None and len(W.temb_off) == 17 and W.temb_cat.shape[0] == sum(n for _, n in W.temb_off.values())
    ocfg = OS.StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0)
    inp = OS.draw_inputs(2, ocfg, seed=7, latent_hw=8, ctx_len=7, ctx_dim=64)
    cfg = StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0)
    args = (inp["latents"], inp["prompt_embeds"], inp["uncond_prompt_embeds"], inp["noise"], inp["index"], inp["w"])

    def run(batched):
        old, M.FUSE_TEMB = M.FUSE_TEMB, batched
        try:
            lora = LoraState(pc, 64, 8.0, "cpu", seed=1, b_std=0.05)
            assert lora.temb is not None and lora.temb.paths == list(W.temb_off)
            D = Distiller(W, lora, cfg)
            out = D.forward_backward(*args)
            assert (D.student._temb is not None) == batched and (D.teacher._temb is not None) == batched
            return {k: out[k].clone() for k in ("noise_pred", "cond_teacher_output", "uncond_teacher_output", "target_noise_pred", "loss")}, lora.grads.clone(), lora
        finally:
            M.FUSE_TEMB = old

    # (1) the batched launches against the per-resnet GEMM, value by value: identical up to isolated one-ulp flips of the 16-bit rounding
    # (the per-lane partial sums of the small-M kernel visit the non-zero block at other k positions)
    from pcm_amd.model import layer_fwd
    lora0 = LoraState(pc, 64, 8.0, "cpu", seed=1, b_std=0.05)
    U = UNet(W, lora0)
    emb = torch.randn(4, W.temb_cat.shape[1], generator=torch.Generator().manual_seed(0)).bfloat16()
    flips = 0
    for path, (temb, t) in U._temb_all(emb, 4).items():
        st = {}
        ref = layer_fwd(W, lora0, path, emb, 4, save=st)
        assert temb.is_contiguous() and t.is_contiguous() and temb.shape == ref.shape and t.shape == st["t"].shape
        d = (temb.float() - ref.float()).abs()
        assert float((d / ref.float().abs().clamp_min(1e-3)).max()) <= 2.0 ** -7 and torch.equal(t, st["t"]), path
        flips += int((d > 0).sum())
    assert flips <= 8, flips                # measured: 1 of 7424 values
    # (2) whole step, both ways
    a, ga, la = run(True)
    b, gb, _ = run(False)
    rel = lambda x, y: float((x.double() - y.double()).norm() / (y.double().norm() + 1e-300))   # noqa: E731
    rels = {k: rel(a[k], b[k]) for k in a}
    print("batched vs per-resnet time_emb_proj:", {k: "%.2e" % v for k, v in rels.items()}, "grads %.2e" % rel(ga, gb))
    # single 16-bit values flip with the summation order; the narrow net amplifies that along teacher -> x_prev -> target (measured: student /
    # teacher predictions bitwise equal, target 5e-3, loss 4e-3, gradient 0.13 after ONE flipped value -- the level at which this config
    # reacts to ANY reordering, cf. the 0.24 of the fp16-teacher test above)
    assert max(rels[k] for k in ("noise_pred", "cond_teacher_output", "uncond_teacher_output")) < 3e-3 and rels["target_noise_pred"] < 2e-2 and rels["loss"] < 2e-2
    assert rel(ga, gb) < 0.3 and float((ga.double() * gb.double()).sum() / (ga.double().norm() * gb.double().norm())) > 0.95
    # the time_emb_proj modules themselves (their t and row vectors come out of the batched launches)
    base = la.grads.data_ptr()
    for path, m in la.modules.items():
        if path.endswith("time_emb_proj"):
            for t in (m.gA, m.gB):
                o0 = (t.data_ptr() - base) // 4
                x, y = ga[o0:o0 + t.numel()], gb[o0:o0 + t.numel()]
                assert float(y.norm()) > 0 and rel(x, y) < 0.5, (path, rel(x, y))


def test_fused_text_kv_of_the_lora_pass_equals_the_per_module_projections():
    """LoRA pass: the rank-64 down-projections of the text for EVERY cross-attention's to_k / to_v are one GEMM + one scatter launch, and per block
    K and V are one GEMM over [W_k; W_v] with a block-diagonal K = 2r second segment (UNet._text_kv_t / _attn_fwd).  Against one down-projection
    and one GEMM per module (PCM_TEXT_KV_LORA=0): the same products in the same order -- t_k / t_v, predictions, loss and gradients bitwise."""
    from oracle import pcm_step as OS
    from oracle import unet_sd15 as O
    from pcm_amd import model as M
    from pcm_amd.model import LoraState, UNet, UNetWeights, layer_fwd
    from pcm_amd.trainer import Distiller, StepConfig
    oc, pc = tiny_cfgs()
    sd = O.init_state_dict(oc, 0)
    W = UNetWeights(pc, sd, "cpu")
    lora0 = LoraState(pc, 64, 8.0, "cpu", seed=1, b_std=0.05)
    assert lora0.text_kv is not None and set(lora0.text_kv.paths) == set(W.kv_off)
    # (1) the pass-wide down-projection, value by value
    text = torch.randn(4 * 7, 64, generator=torch.Generator().manual_seed(0)).bfloat16()
    U = UNet(W, lora0)
    for b, (t_kv, t_k, t_v) in U._text_kv_t(text).items():
        for nme, t in (("to_k", t_k), ("to_v", t_v)):
            st = {}
            layer_fwd(W, lora0, b + nme, text, 28, save=st)
            assert t.is_contiguous() and torch.equal(t, st["t"]), b + nme
        assert torch.equal(t_kv[:, :64], t_k) and torch.equal(t_kv[:, 64:], t_v)
    # (2) whole step, both ways
    ocfg = OS.StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0)
    inp = OS.draw_inputs(2, ocfg, seed=7, latent_hw=8, ctx_len=7, ctx_dim=64)
    cfg = StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0)
    args = (inp["latents"], inp["prompt_embeds"], inp["uncond_prompt_embeds"], inp["noise"], inp["index"], inp["w"])

    def run(fused):
        old, M.FUSE_TEXT_KV_LORA = M.FUSE_TEXT_KV_LORA, fused
        try:
            lora = LoraState(pc, 64, 8.0, "cpu", seed=1, b_std=0.05)
            D = Distiller(W, lora, cfg)
            out = D.forward_backward(*args)
            assert (D.student._text_t is not None) == fused and D.teacher._text_t is None
            return {k: out[k].clone() for k in ("noise_pred", "target_noise_pred", "loss")}, lora.grads.clone(), lora
        finally:
            M.FUSE_TEXT_KV_LORA = old

    a, ga, la = run(True)
    b, gb, _ = run(False)
    rel = lambda x, y: float((x.double() - y.double()).norm() / (y.double().norm() + 1e-300))   # noqa: E731
    rels = {k: rel(a[k], b[k]) for k in a}
    print("fused text K|V vs per-module:", {k: "%.2e" % v for k, v in rels.items()}, "grads %.2e" % rel(ga, gb))
    # measured: bitwise -- the zero blocks add exact zeros, the K|V GEMM runs the kernel the per-module GEMMs run, attention reads the same values
    assert max(rels.values()) == 0.0 and rel(ga, gb) < 1e-6
    base = la.grads.data_ptr()
    for path, m in la.modules.items():          # the to_k / to_v modules themselves: their gradients come through the strided k / v and the scattered t
        if path.endswith(("attn2.to_k", "attn2.to_v")):
            for t in (m.gA, m.gB):
                o0 = (t.data_ptr() - base) // 4
                x, y = ga[o0:o0 + t.numel()], gb[o0:o0 + t.numel()]
                assert float(y.norm()) > 0 and rel(x, y) < 1e-6, (path, rel(x, y))


def test_cross_step_teacher_prefetch_gives_the_same_training_sequence():
    """Distiller.step(..., prefetch=next batch) (round 6): the frozen teacher's pass of batch k+1 is issued beside the student's work on batch k
    (a side stream on the GPU; the same call order on the host emulator) and picked up by the next call.  It reads nothing trainable, so three
    optimizer steps give the same losses, gradients and parameters as three plain steps -- bit for bit; a batch that was not announced (or
    announced and then replaced) falls back to computing its targets in the call."""
    from oracle import pcm_step as OS
    from oracle import unet_sd15 as O
    from pcm_amd.model import LoraState, UNetWeights
    from pcm_amd.trainer import Distiller, StepConfig
    oc, pc = tiny_cfgs()
    sd = O.init_state_dict(oc, 0)
    W = UNetWeights(pc, sd, "cpu")
    ocfg = OS.StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0)
    cfg = StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0, learning_rate=1e-3)
    keys = ("latents", "prompt_embeds", "uncond_prompt_embeds", "noise", "index", "w")
    batches = [tuple(OS.draw_inputs(2, ocfg, seed=20 + i, latent_hw=8, ctx_len=7, ctx_dim=64)[k] for k in keys) for i in range(4)]

    def run(prefetch):
        lora = LoraState(pc, 64, 8.0, "cpu", seed=1, b_std=0.05)
        D = Distiller(W, lora, cfg)
        losses = []
        for i in range(3):
            nxt = batches[i + 1] if prefetch else None
            if prefetch and i == 1:
                nxt = batches[0]                     # announced batch 0, the next call brings batch 2: must not use the stale targets
            out = D.step(*batches[i], prefetch=nxt)
            losses.append(float(out["loss"]))
        return losses, lora.params.clone(), lora.grads.clone()

    l0, p0, g0 = run(False)
    l1, p1, g1 = run(True)       # (call 1 runs on prefetched targets, call 2 on a wrong announcement)
    assert l0 == l1 and torch.equal(p0, p1) and torch.equal(g0, g1)
    assert len(set(l0)) == 3
