"""GPU parity of the full phased-consistency step at the real SD1.5 UNet size (through the C ABI)
against the CPU fp32 oracle on identical seeded inputs.  Tolerances: the HIP path computes in bf16
with fp32 accumulation; the reference-owned PCM math is fp32/fp64 and must agree much tighter."""
import math
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    from oracle import unet_sd15 as O
    from pcm_amd import capi
    from pcm_amd.model import LoraState, UNetWeights
    from pcm_amd.unet_spec import UNetConfig
    capi.set_lib(None)
    capi.lib()
    assert torch.cuda.is_available()
    oc = O.UNetConfig.sd15()
    sd = O.init_state_dict(oc, 0)
    W = UNetWeights(UNetConfig.sd15(), sd, "cuda")
    return oc, sd, W


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("b_std", [0.0, 0.02])
def test_full_step_vs_oracle(setup, b_std):
    from oracle import pcm_step as OS
    from pcm_amd.model import LoraState
    from pcm_amd.trainer import Distiller, StepConfig
    from pcm_amd.unet_spec import UNetConfig
    oc, sd, W = setup
    B = 2
    ocfg = OS.StepConfig(multiphase=2, loss_type="huber", lr=5e-6, adam_weight_decay=1e-3, w_min=4.0, w_max=5.0)
    inp = OS.draw_inputs(B, ocfg, seed=453645634)
    inp["index"] = torch.tensor([13, 37])
    lora = LoraState(UNetConfig.sd15(), 64, 8.0, "cuda", seed=1, b_std=b_std)
    olora = {p: (lora.A_peft(m).detach().cpu().clone(), m.B.detach().cpu().clone()) for p, m in lora.modules.items()}
    def flat_peft(which):   # this build's flat buffers in the oracle's (peft) order and layout
        out = []
        for m in lora.modules.values():
            a, b = (m.A, m.B) if which == "p" else (m.gA, m.gB)
            out += [lora.to_peft(m, a).detach().cpu().reshape(-1), b.detach().cpu().reshape(-1)]
        return torch.cat(out)
    p_before = flat_peft("p")
    t0 = time.time()
    ref = OS.distill_step(oc, sd, olora, inp, ocfg, {}, 1)
    print("oracle step %.1f s" % (time.time() - t0))
    cfg = StepConfig(multiphase=2, loss_type="huber", learning_rate=5e-6, adam_weight_decay=1e-3, w_min=4.0, w_max=5.0)
    D = Distiller(W, lora, cfg)
    dev = {k: v.cuda() for k, v in inp.items()}
    out = D.step(dev["latents"], dev["prompt_embeds"], dev["uncond_prompt_embeds"], dev["noise"], dev["index"], dev["w"])
    torch.cuda.synchronize()
    assert torch.equal(out["start_timesteps"].cpu(), ref["start_timesteps"]) and torch.equal(out["timesteps"].cpu(), ref["timesteps"])
    assert torch.equal(out["end_timesteps"].cpu(), ref["end_timesteps"])
    # reference-owned fp32 math.  The HIP kernel is bit-exact against the committed golden fixtures
    # (tests/test_gpu_kernels.py::test_pcm_math_bit_exact_vs_reference_golden); the oracle evaluated
    # live on THIS host's CPU can differ in the last bit (torch's CPU sqrt is not correctly rounded on
    # every host: measured 5/16 last-bit differences vs torch's own GPU sqrt on the MI355X box).
    assert torch.allclose(out["noisy_model_input"].cpu(), ref["noisy_model_input"], rtol=3e-7, atol=1e-7)
    report = {}
    for k in ("noise_pred", "cond_teacher_output", "uncond_teacher_output", "x_prev", "target_noise_pred", "model_pred", "target"):
        report[k] = rel(out[k], ref[k])
    loss, rloss = float(out["loss"].item()), float(ref["loss"])
    report["loss_rel"] = abs(loss - rloss) / abs(rloss)
    # gradients (post-clip in the oracle; compare direction + norm) and updated parameters
    gn = math.sqrt(float(out["grad_sumsq"].item()))
    report["grad_norm_rel"] = abs(gn - float(ref["grad_norm"])) / float(ref["grad_norm"])
    coef = min(1.0, 1.0 / (float(ref["grad_norm"]) + 1e-6))
    flat_ref = torch.cat([g.reshape(-1) for g in ref["grads"]]) / coef
    report["grad_rel"] = rel(flat_peft("g"), flat_ref)
    flat_p = torch.cat([t.reshape(-1) for ab in olora.values() for t in ab])
    report["param_rel"] = rel(flat_peft("p"), flat_p)
    d_mine, d_ref = (flat_peft("p") - p_before).double(), (flat_p - p_before).double()
    report["update_cos"] = float((d_mine * d_ref).sum() / (d_mine.norm() * d_ref.norm() + 1e-30))
    print("b_std", b_std, {k: "%.3e" % v for k, v in report.items()}, "loss", loss, rloss)
    import json, os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/step_parity_bstd%g.json" % b_std, "w") as f:
        json.dump(dict(report, loss=loss, oracle_loss=rloss, b_std=b_std), f)
    # Bounds = ~1.5-2x what MI355X measures (profiles/step_parity_bstd*.json, r02_error_budget_sd15_bs2.json); every one of them is
    # BELOW the deviation of the reference's own bf16-autocast numerics from the same fp32 oracle (eps 1.2e-2, model_pred 2.6e-3,
    # target 2.9e-3, x_prev 1.6e-3, loss 9.4e-3), so a regression to "merely as good as the reference's mixed precision" fails here.
    assert report["noise_pred"] < 1.2e-2 and report["cond_teacher_output"] < 1.2e-2 and report["target_noise_pred"] < 1.2e-2
    assert report["x_prev"] < 1.6e-3 and report["model_pred"] < 2.6e-3 and report["target"] < 2.9e-3
    assert report["loss_rel"] < 9e-3
    assert report["grad_rel"] < 6e-2 and report["grad_norm_rel"] < 0.02
    assert report["param_rel"] < 2e-4 and report["update_cos"] > 0.94
