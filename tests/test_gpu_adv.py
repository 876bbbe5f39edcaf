"""GPU parity of the adversarial step (D and G updates) at the real SD1.5 size and the BASELINE configs[2] shape (36 heads) vs the CPU oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("global_step", [0, 1])
def test_adv_step_c3_shape_full_size(global_step):
    """BASELINE configs[2] as configured: the SD1.5 UNet, all 36 heads (9 taps x 4, discriminator_sd15.py:371-393), batch 2, the reference's
    learning rates (lr 5e-6, adv_lr 1e-5) -- the step changes the heads (even) or the LoRA (odd) and the UPDATE is compared with the
    oracle's clip + AdamW on the oracle's gradients.  Writes gpurun_out/adv_c3_parity_step{0,1}.json."""
    import json
    import os
    import adv_cases as A
    from pcm_amd import capi
    from pcm_amd.discriminator import ADAPTER_DIMS
    capi.set_lib(None)
    capi.lib()
    kw = dict(block_out_channels=(320, 640, 1280, 1280), cross_attention_dim=768, heads=8, norm_num_groups=32)
    rep = A.case_adv_c3("cuda", kw, ADAPTER_DIMS, 2, 64, 77, 768, global_step, nh=4, index=[30, 12],
                        golden_name="sd15_adv_c3_bs2_step%d" % global_step)     # oracle side: committed fixture (tests/golden/make_golden_step.py)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rep, open("gpurun_out/adv_c3_parity_step%d.json" % global_step, "w"), indent=1)
    assert rep["heads"] == 36 and rep["fake_adv"] < 3e-3
    if global_step % 2 == 0:
        # bounds = measured on MI355X (profiles/r03_a_adv_c3_36heads_bs2_dstep.json: 1.1e-4, 0.9963, per tap >= 0.9904, 0.26 %, 0.937) with margin
        assert rep["d_loss_rel"] < 1e-3 and rep["lora_untouched"]
        assert rep["head_grad_cos"] > 0.99 and min(rep["head_grad_cos_per_tap"]) > 0.98 and rep["head_grad_norm_rel"] < 1e-2
        # first AdamW step with beta1 = 0: update = -lr * g / (|g| + eps) ~ -lr * sign(g): the cosine counts agreeing signs, the norm is lr * sqrt(n)
        assert rep["head_update_cos"] > 0.92 and abs(rep["head_update_norm_ratio"] - 1) < 2e-2 and rep["head_param_rel_after"] < 6e-4
    else:
        # measured (profiles/r03_b_adv_c3_36heads_bs2_gstep.json): loss_cm 7.9e-3, g_loss 5e-5, cos 0.9987, norm 0.54 %, update cos 0.944
        assert rep["loss_cm_rel"] < 1.2e-2 and rep["g_loss_rel"] < 1e-3 and rep["heads_untouched"]
        assert rep["lora_grad_cos"] > 0.995 and rep["lora_grad_norm_rel"] < 2e-2
        assert rep["lora_update_cos"] > 0.92 and abs(rep["lora_update_norm_ratio"] - 1) < 2e-2 and rep["lora_param_rel_after"] < 6e-4


def test_adv_steps_graph_replay_equals_eager():
    """AdvDistiller.capture_adv: D, G, D, G through the two captured hipGraphs == the same four eager steps on a twin trainer
    (narrow 2-level UNet, 5 tapped features x 2 heads; real learning rates so every step changes the state the next one reads)."""
    from oracle import pcm_step as OS
    from oracle import unet_sd15 as O
    from pcm_amd import capi
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
    dims = (64, 128, 128, 128, 64)
    cfg = StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0, learning_rate=1e-4)
    ocfg = OS.StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0)
    trainers = []
    for _ in range(2):
        lora = LoraState(pc, 64, 8.0, "cuda", seed=1, b_std=0.05)
        disc = Discriminator(dims, num_h_per_head=2, device="cuda", seed=2)
        trainers.append(AdvDistiller(W, lora, cfg, disc, adv_weight=0.1, adv_lr=1e-4))
    Dg, De = trainers
    B, Hh = 2, 16
    Dg.capture_adv(B, H=Hh, W=Hh, ctx_len=77, ctx_dim=64)
    for step in range(4):
        inp = OS.draw_inputs(B, ocfg, seed=40 + step, latent_hw=Hh, ctx_len=77, ctx_dim=64)
        g = torch.Generator().manual_seed(90 + step)
        inp["noise_fake"], inp["noise_real"] = torch.randn(B, 4, Hh, Hh, generator=g), torch.randn(B, 4, Hh, Hh, generator=g)
        inp["adv_u"] = torch.rand(B, generator=g)
        a = [inp[k].cuda() for k in ("latents", "prompt_embeds", "uncond_prompt_embeds", "noise", "index", "w", "noise_fake", "noise_real", "adv_u")]
        og = Dg.step_adv_graphed(step, *a)
        key = "d_loss" if step % 2 == 0 else "loss_cm"
        lg = float(og[key])
        oe = De.step_adv(step, *a)
        le = float(oe[key])
        # (step 0 sees identical state; later steps see states that differ by the fp32-atomics order of the earlier updates -- under the half build's
        #  11-bit significands that showed as 1.5e-4 on the fourth step in one of six runs, profiles/r06_f_*; the bitwise form of this
        #  comparison runs under set_deterministic: tests/test_gpu_deterministic_adv.py)
        assert abs(lg - le) <= (1e-5 if step == 0 else 5e-4) * abs(le) + 1e-9, (step, key, lg, le)
    rel = lambda x, y: float((x - y).norm() / (y.norm() + 1e-30))
    # Adam divides by sqrt(v): on entries whose gradient is at the fp32-atomics noise level the update direction itself is noise, so the
    # states agree to a fraction of one lr-sized step (lr 1e-4 against |param| ~ 2e-2), not bitwise
    assert rel(Dg.lora.params, De.lora.params) < 2e-4 and rel(Dg.disc.params, De.disc.params) < 2e-4
    assert Dg.step_count == De.step_count == 2
