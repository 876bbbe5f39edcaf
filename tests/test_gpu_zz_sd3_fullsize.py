"""SD3-medium at its real size (24 layers x 1536, 128x128x16 latents -> 4096 image + 154 text tokens) on the MI355X, checked against
the fp32 oracle on ONE sample at the real size (forward without / with LoRA, rel-L2 <= 1.5e-2) and through size-independent properties: finiteness, batch independence
(no cross-sample coupling through the token-axis concat / fused q/k/v views at real strides), LoRA with B = 0 is the teacher, and
one full distillation step -- tests/mmdit_cases.py::run_property_case, which also runs on a narrow config on the emulator.
Runs last (file name) so that the first full-size execution of this path cannot shadow other tests."""
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def test_sd3_medium_full_size_properties():
    from mmdit_cases import run_property_case
    from pcm_amd import capi
    from pcm_amd.mmdit import MMDiTWeights, sd3_lora_state
    from pcm_amd.mmdit_spec import MMDiTConfig, random_state_dict
    capi.lib()
    dev = torch.device("cuda", 0)
    cfg = MMDiTConfig.sd3_medium()
    sd = random_state_dict(cfg, 0, "cpu")                  # seeded CPU draw: the committed oracle fixture was evaluated on the SAME weights
    W = MMDiTWeights(cfg, sd, dev)
    del sd
    torch.cuda.empty_cache()
    _oracle_parity_one_sample(cfg, W, dev)
    _oracle_parity_whole_step(cfg, W, dev)
    lora = sd3_lora_state(cfg, 32, 8.0, dev, seed=1)                     # B = 0 (reference init)
    loss, gnorm = run_property_case(dev, cfg, W, lora, 128, 154)
    torch.cuda.synchronize()
    print("SD3 full-size step: loss %.5f, grad norm %.4e, peak %.1f GB" % (loss, gnorm, torch.cuda.max_memory_allocated() / 1e9))


def _oracle_parity_one_sample(cfg, W, dev):
    """ONE sample at SD3-medium's real size (24 blocks x 1536, 4096 image + 154 text tokens) against the fp32 oracle
    (oracle/mmdit_sd3.py): transformer forward without and with LoRA (rank 32, B ~ N(0, 0.05)); the oracle's outputs come from
    tests/golden/step_sd3_fullsize_one_sample.safetensors (tests/step_golden_cases.py::ref_sd3_one_sample).  Bound: rel-L2 <= 1.5e-2."""
    import json
    import os
    import step_golden_cases as S
    from golden_fixture import golden
    from pcm_amd.mmdit import MMDiT, sd3_lora_state
    ref = golden("sd3_fullsize_one_sample", S.ref_sd3_one_sample)
    ref_t, ref_s = ref["teacher"], ref["student"]
    lora = sd3_lora_state(cfg, 32, 8.0, dev, seed=3, b_std=0.05)
    a = [v.to(dev) for v in S.sd3_one_sample_inputs()]
    out_t = MMDiT(W, None).forward(*a).cpu()
    out_s = MMDiT(W, lora).forward(*a).cpu()
    rel = lambda u, v: float((u.double() - v.double()).norm() / v.double().norm())   # noqa: E731
    rep = dict(teacher_rel_l2=rel(out_t, ref_t), student_rel_l2=rel(out_s, ref_s), lora_effect_rel=rel(ref_s, ref_t), oracle_seconds=ref["oracle_seconds"])
    print("SD3 full-size oracle parity (1 sample):", rep)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rep, open("gpurun_out/sd3_fullsize_oracle_parity.json", "w"), indent=1)
    assert rep["teacher_rel_l2"] < 1.5e-2 and rep["student_rel_l2"] < 1.5e-2, rep
    assert rep["lora_effect_rel"] > 3 * rep["student_rel_l2"], rep
    del lora
    torch.cuda.empty_cache()


def _oracle_parity_whole_step(cfg, W, dev):
    """ONE WHOLE distillation step of one sample at SD3-medium's real size against the fp32 oracle (train_pcm_lora_sd3.py:1270-1390 with the
    2-phase deterministic recipe of BASELINE configs[4]: student forward with LoRA r = 32, teacher cond / uncond, fixed-w CFG + Euler step,
    target forward, multiphase jump, huber loss, LoRA-only backward, clip, AdamW): forward tensors, loss, LoRA gradients (count-sketch),
    gradient norm, AdamW update.  Oracle side: tests/golden/step_sd3_fullsize_step_one_sample.safetensors
    (tests/step_golden_cases.py::ref_sd3_step_fullsize)."""
    import json
    import math
    import os
    import step_golden_cases as S
    from golden_fixture import golden, sk_cos, sk_rel, sketch
    from pcm_amd.mmdit import sd3_lora_state
    from pcm_amd.trainer_sd3 import SD3Distiller, SD3StepConfig
    ref = golden("sd3_fullsize_step_one_sample", S.ref_sd3_step_fullsize)
    lora = sd3_lora_state(cfg, 32, 8.0, dev, seed=3, b_std=0.05)
    p_before = S.sd3_lora_flat(lora, "p")
    assert sk_rel(sketch(p_before), ref["sk_param_before"]) < 1e-6
    D = SD3Distiller(W, lora, SD3StepConfig(multiphase=2, learning_rate=5e-6, adam_weight_decay=1e-2))
    gsc = float(D.loss_scale_dev.item()) if D.loss_scale_dev is not None else 1.0
    out = D.step(*(t.to(dev) for t in S.sd3_step_inputs()))
    torch.cuda.synchronize()
    assert torch.equal(out["end_index"].cpu(), ref["end_index"])
    assert torch.equal(out["noisy_model_input"].cpu(), ref["noisy_model_input"])                     # reference-owned math: bit-exact
    rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / (b.double().norm() + 1e-30))   # noqa: E731
    rep = {k: rel(out[k], ref[k]) for k in S.SD3_STEP_KEYS if k in out}
    loss, rloss = float(out["loss"]), float(ref["loss"])
    rep["loss_rel"] = abs(loss - rloss) / abs(rloss)
    g = S.sd3_lora_flat(lora, "g") / gsc
    rep["grad_norm_rel"] = abs(float(g.double().norm()) - float(ref["grad_norm"])) / float(ref["grad_norm"])
    sg = sketch(g)
    rep["grad_rel"], rep["grad_cos"] = sk_rel(sg, ref["sk_grad"]), sk_cos(sg, ref["sk_grad"])
    p_after = S.sd3_lora_flat(lora, "p")
    rep["param_rel"] = sk_rel(sketch(p_after), ref["sk_param_after"])
    rep["update_cos"] = sk_cos(sketch(p_after - p_before), ref["sk_update"])
    rep.update(loss=loss, oracle_loss=rloss, oracle_seconds=ref["oracle_seconds"])
    print("SD3-medium full-size WHOLE STEP vs fp32 oracle (1 sample):", {k: "%.3e" % v for k, v in rep.items()})
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rep, open("gpurun_out/sd3_fullsize_step_parity.json", "w"), indent=1)
    assert rep["model_output"] < 1.5e-2 and rep["cond_teacher_output"] < 1.5e-2, rep
    assert rep["x_prev"] < 3e-3 and rep["model_pred"] < 1.5e-2 and rep["target"] < 1.5e-2, rep
    assert rep["loss_rel"] < 2e-2, rep
    assert rep["grad_cos"] > 0.99 and rep["grad_rel"] < 0.12 and rep["grad_norm_rel"] < 0.03, rep
    assert rep["param_rel"] < 2e-4 and rep["update_cos"] > 0.9, rep
    for m in lora.modules.values():                                                                    # padded ranks stay exactly zero
        assert float(m.A[32:].abs().max()) == 0.0 and float(m.B[:, 32:].abs().max()) == 0.0
    del D, lora
    torch.cuda.empty_cache()
