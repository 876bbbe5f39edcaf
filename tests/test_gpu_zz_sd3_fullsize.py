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
    sd = random_state_dict(cfg, 0, dev)
    W = MMDiTWeights(cfg, sd, dev)
    sd_cpu = {k: v.cpu() for k, v in sd.items()}          # the oracle evaluates the SAME weights
    del sd
    torch.cuda.empty_cache()
    _oracle_parity_one_sample(cfg, sd_cpu, W, dev)
    del sd_cpu
    lora = sd3_lora_state(cfg, 32, 8.0, dev, seed=1)                     # B = 0 (reference init)
    loss, gnorm = run_property_case(dev, cfg, W, lora, 128, 154)
    torch.cuda.synchronize()
    print("SD3 full-size step: loss %.5f, grad norm %.4e, peak %.1f GB" % (loss, gnorm, torch.cuda.max_memory_allocated() / 1e9))


def _oracle_parity_one_sample(cfg, sd_cpu, W, dev):
    """ONE sample at SD3-medium's real size (24 blocks x 1536, 4096 image + 154 text tokens) against the fp32 oracle
    (oracle/mmdit_sd3.py): transformer forward without and with LoRA (rank 32, B ~ N(0, 0.05)).  Bound: rel-L2 <= 1.5e-2."""
    import json
    import os
    import time
    from oracle import mmdit_sd3 as O
    from pcm_amd.mmdit import MMDiT, sd3_lora_state
    oc = O.MMDiTConfig.sd3_medium()
    g = torch.Generator().manual_seed(7)
    x, t = torch.randn(1, 16, 128, 128, generator=g), torch.tensor([640.5])
    ctx, pooled = torch.randn(1, 154, 4096, generator=g), torch.randn(1, 2048, generator=g)
    lora = sd3_lora_state(cfg, 32, 8.0, dev, seed=3, b_std=0.05)
    olora = {p: (m.A[:32].detach().cpu().clone(), m.B[:, :32].detach().cpu().clone()) for p, m in lora.modules.items()}
    t0 = time.time()
    with torch.no_grad():
        ref_t = O.mmdit_forward(oc, sd_cpu, x, t, ctx, pooled)
        ref_s = O.mmdit_forward(oc, sd_cpu, x, t, ctx, pooled, olora, 8.0)
    cpu_s = time.time() - t0
    a = [v.to(dev) for v in (x, t, ctx, pooled)]
    out_t = MMDiT(W, None).forward(*a).cpu()
    out_s = MMDiT(W, lora).forward(*a).cpu()
    rel = lambda u, v: float((u.double() - v.double()).norm() / v.double().norm())   # noqa: E731
    rep = dict(teacher_rel_l2=rel(out_t, ref_t), student_rel_l2=rel(out_s, ref_s), lora_effect_rel=rel(ref_s, ref_t), oracle_seconds=cpu_s)
    print("SD3 full-size oracle parity (1 sample):", rep)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rep, open("gpurun_out/sd3_fullsize_oracle_parity.json", "w"), indent=1)
    assert rep["teacher_rel_l2"] < 1.5e-2 and rep["student_rel_l2"] < 1.5e-2, rep
    assert rep["lora_effect_rel"] > 3 * rep["student_rel_l2"], rep
    del lora
    torch.cuda.empty_cache()
