"""ops.set_deterministic(True) beyond the consistency step (round-5 review, item 7a): the adversarial trainers (BASELINE configs[2] / [4]) --
the discriminator heads' parameter gradients (pcm_rowdot_bwd_ws, pcm_groupnorm_param_grad_ws, bias pixel sums, the rank-64 slab form instead of
the dense conv weight gradient's atomics), the hinge losses (pcm_hinge_loss_ordered) and the MMDiT modulation gradients (pcm_mod_grad_ws) -- are
bitwise repeatable run to run, and the captured hipGraphs replay to the eager steps bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _adv_inputs(OS, ocfg, B, Hh, step):
    inp = OS.draw_inputs(B, ocfg, seed=40 + step, latent_hw=Hh, ctx_len=77, ctx_dim=64)
    g = torch.Generator().manual_seed(90 + step)
    inp["noise_fake"], inp["noise_real"] = torch.randn(B, 4, Hh, Hh, generator=g), torch.randn(B, 4, Hh, Hh, generator=g)
    inp["adv_u"] = torch.rand(B, generator=g)
    return [inp[k].cuda() for k in ("latents", "prompt_embeds", "uncond_prompt_embeds", "noise", "index", "w", "noise_fake", "noise_real", "adv_u")]


def test_sd15_adversarial_steps_are_bitwise_reproducible_and_graph_replay_equals_eager():
    """D, G, D, G on four twin trainers (eager, eager, captured hipGraphs, captured hipGraphs carrying the next batch's ODE-solver teacher pass as
    a forked branch) of a narrow SD1.5-topology UNet with 5 taps x 2 heads at a map size whose reductions span many workgroups: losses, LoRA
    and head parameters BITWISE equal after every step."""
    from oracle import pcm_step as OS
    from oracle import unet_sd15 as O
    from pcm_amd import capi, ops
    from pcm_amd.discriminator import Discriminator
    from pcm_amd.model import LoraState, UNetWeights
    from pcm_amd.trainer import AdvDistiller, StepConfig
    from pcm_amd.unet_spec import UNetConfig
    capi.set_lib(None)
    capi.lib()
    kw = dict(block_out_channels=(64, 128), layers_per_block=1, cross_attention_dim=64, heads=2, norm_num_groups=32)
    oc, pc = O.UNetConfig(**kw), UNetConfig(**kw)
    sd = O.init_state_dict(oc, 0)
    W = UNetWeights(pc, sd, "cuda")
    cfg = StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0, learning_rate=1e-4)
    ocfg = OS.StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0)
    ops.set_deterministic(True)
    try:
        trainers = []
        for _ in range(4):
            lora = LoraState(pc, 64, 8.0, "cuda", seed=1, b_std=0.05)
            disc = Discriminator((64, 128, 128, 128, 64), num_h_per_head=2, device="cuda", seed=2)
            trainers.append(AdvDistiller(W, lora, cfg, disc, adv_weight=0.1, adv_lr=1e-4))
        Da, Db, Dg, Dp = trainers
        B, Hh = 4, 32
        Dg.capture_adv(B, H=Hh, W=Hh, ctx_len=77, ctx_dim=64)
        Dp.capture_adv(B, H=Hh, W=Hh, ctx_len=77, ctx_dim=64, pipeline=True)
        batches = [_adv_inputs(OS, ocfg, B, Hh, step) for step in range(5)]
        for step in range(4):
            a = batches[step]
            key = "d_loss" if step % 2 == 0 else "loss_cm"
            # (Db: eager with the side-stream prefetch; step 2 announces nothing, so step 3 computes its own targets / runs the graph's eager prologue)
            nxt = tuple(batches[step + 1][:6]) if step != 2 else None
            la, lb = float(Da.step_adv(step, *a)[key]), float(Db.step_adv(step, *a, prefetch=nxt)[key])
            lg = float(Dg.step_adv_graphed(step, *a)[key])
            lp = float(Dp.step_adv_graphed(step, *a, prefetch=nxt)[key])
            assert la == lb == lg == lp, (step, key, la, lb, lg, lp)
            for x, y in ((Da, Db), (Da, Dg), (Da, Dp)):
                assert torch.equal(x.lora.params, y.lora.params) and torch.equal(x.disc.params, y.disc.params), (step, "state differs")
                assert torch.equal(x.disc.grads, y.disc.grads) if step % 2 == 0 else torch.equal(x.lora.grads, y.lora.grads), (step, "gradients differ")
    finally:
        ops.set_deterministic(False)


@pytest.mark.parametrize("global_step", [0, 1])
def test_sd15_adversarial_step_c3_shape_is_bitwise_reproducible(global_step):
    """the BASELINE configs[2] shape (SD1.5 UNet, 36 heads, bs 2) run twice from the same state: identical bits in every gradient and parameter"""
    from pcm_amd import capi, ops
    from pcm_amd.discriminator import ADAPTER_DIMS, Discriminator
    from pcm_amd.model import LoraState, UNetWeights
    from pcm_amd.trainer import AdvDistiller, StepConfig
    from pcm_amd.unet_spec import UNetConfig, random_state_dict
    capi.set_lib(None)
    capi.lib()
    ucfg = UNetConfig.sd15()
    with torch.no_grad():
        W = UNetWeights(ucfg, random_state_dict(ucfg, seed=0, device="cuda"), "cuda")
    cfg = StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0, learning_rate=5e-6)
    B = 2
    g = torch.Generator(device="cuda").manual_seed(3)
    a = [torch.randn(B, 4, 64, 64, generator=g, device="cuda"), torch.randn(B, 77, 768, generator=g, device="cuda"), torch.randn(B, 77, 768, generator=g, device="cuda"),
         torch.randn(B, 4, 64, 64, generator=g, device="cuda"), torch.tensor([30, 12], device="cuda"), torch.tensor([4.2, 4.8], device="cuda"),
         torch.randn(B, 4, 64, 64, generator=g, device="cuda"), torch.randn(B, 4, 64, 64, generator=g, device="cuda"), torch.rand(B, generator=g, device="cuda")]
    ops.set_deterministic(True)
    try:
        res = []
        for _ in range(2):
            lora = LoraState(ucfg, 64, 8.0, "cuda", seed=1, b_std=0.02)
            disc = Discriminator(ADAPTER_DIMS, num_h_per_head=4, device="cuda", seed=2)
            D = AdvDistiller(W, lora, cfg, disc, adv_weight=0.1, adv_lr=1e-5)
            out = D.step_adv(global_step, *a)
            torch.cuda.synchronize()
            res.append((float(out["d_loss" if global_step % 2 == 0 else "loss_cm"]), lora.params.clone(), disc.params.clone(),
                        (disc.grads if global_step % 2 == 0 else lora.grads).clone()))
            del D, lora, disc
        assert res[0][0] == res[1][0]
        for x, y in zip(res[0][1:], res[1][1:]):
            assert torch.equal(x, y)
        assert float(res[0][3].abs().max()) > 0
    finally:
        ops.set_deterministic(False)


@pytest.mark.parametrize("global_step", [0, 1])
def test_sd3_adversarial_step_is_bitwise_reproducible(global_step):
    """BASELINE configs[4] family (MMDiT, narrow config with the trainers' 22-entry LoRA list incl. the adaLN projections whose gradients go
    through pcm_mod_grad): one discriminator and one generator step, run twice from the same state"""
    from oracle import mmdit_sd3 as O
    from pcm_amd import capi, ops
    from pcm_amd.discriminator import Discriminator
    from pcm_amd.mmdit import MMDiTWeights, sd3_lora_state
    from pcm_amd.mmdit_spec import LORA_TARGETS_SD3_ADV, MMDiTConfig
    from pcm_amd.trainer_sd3 import SD3AdvDistiller, SD3StepConfig
    capi.set_lib(None)
    capi.lib()
    kw = dict(sample_size=32, num_layers=2, attention_head_dim=64, num_attention_heads=2, joint_attention_dim=96, caption_projection_dim=128,
              pooled_projection_dim=64, pos_embed_max_size=24)
    oc, pc = O.MMDiTConfig(**kw), MMDiTConfig(**kw)
    sd = O.init_state_dict(oc, 0)
    W = MMDiTWeights(pc, {k: v.cuda() for k, v in sd.items()}, "cuda")
    B, H, Wd, Lc = 4, 32, 32, 5
    g = torch.Generator().manual_seed(5)
    x0, noise = torch.randn(B, 16, H, Wd, generator=g), torch.randn(B, 16, H, Wd, generator=g)
    pe, upe = torch.randn(B, Lc, 96, generator=g), torch.randn(B, Lc, 96, generator=g)
    pp, upp = torch.randn(B, 64, generator=g), torch.randn(B, 64, generator=g)
    nf, nr = torch.randn(B, 16, H, Wd, generator=g, dtype=torch.float64), torch.randn(B, 16, H, Wd, generator=g, dtype=torch.float64)
    index, adv_u = torch.tensor([0, 49, 13, 30]), torch.tensor([0.0, 0.99, 0.5, 0.26])
    a = [t.cuda() for t in (x0, pe, pp, upe, upp, noise, index, nf, nr, adv_u)]
    ops.set_deterministic(True)
    try:
        res = []
        for _ in range(2):
            lora = sd3_lora_state(pc, 32, 8.0, "cuda", seed=1, b_std=0.05, targets=LORA_TARGETS_SD3_ADV, init="kaiming")
            disc = Discriminator([128] * 2, num_h_per_head=1, device="cuda", seed=4, ksize=1)
            D = SD3AdvDistiller(W, lora, SD3StepConfig(multiphase=4, learning_rate=1e-4), disc, adv_weight=0.1, adv_lr=1e-4)
            out = D.step_adv(global_step, *a)
            torch.cuda.synchronize()
            res.append((float(out["d_loss" if global_step % 2 == 0 else "loss_cm"]), lora.params.clone(), disc.params.clone(),
                        (disc.grads if global_step % 2 == 0 else lora.grads).clone()))
        assert res[0][0] == res[1][0]
        for x, y in zip(res[0][1:], res[1][1:]):
            assert torch.equal(x, y)
        assert float(res[0][3].abs().max()) > 0
    finally:
        ops.set_deterministic(False)
