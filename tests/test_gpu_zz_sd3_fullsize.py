"""SD3-medium at its real size (24 layers x 1536, 128x128x16 latents -> 4096 image + 154 text tokens) on the MI355X, checked through
size-independent properties (the fp32 oracle of a 2 B-parameter model is not a seconds-scale CPU job): finiteness, batch independence
(no cross-sample coupling through the token-axis concat / fused q/k/v views at real strides), LoRA with B = 0 is the teacher, and
one full distillation step.  Runs last (file name) so that the first full-size execution of this path cannot shadow other tests."""
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


@pytest.fixture(scope="module")
def model():
    from pcm_amd import capi
    from pcm_amd.mmdit import MMDiTWeights, sd3_lora_state
    from pcm_amd.mmdit_spec import MMDiTConfig, random_state_dict
    capi.lib()
    dev = torch.device("cuda", 0)
    cfg = MMDiTConfig.sd3_medium()
    sd = random_state_dict(cfg, 0, dev)
    W = MMDiTWeights(cfg, sd, dev)
    del sd
    torch.cuda.empty_cache()
    return cfg, W, sd3_lora_state(cfg, 32, 8.0, dev, seed=1)          # B = 0 (reference init)


def _inputs(B, dev, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g, device=dev)   # noqa: E731
    return r(B, 16, 128, 128), torch.tensor([900.5, 120.25][:B], device=dev), r(B, 154, 4096), r(B, 2048)


def test_forward_properties(model):
    from pcm_amd.mmdit import MMDiT
    cfg, W, lora = model
    dev = W.device
    x, t, c, p = _inputs(2, dev)
    teacher = MMDiT(W, None)
    out = teacher.forward(x, t, c, p)
    assert out.shape == (2, 16, 128, 128) and bool(torch.isfinite(out).all())
    assert float(out.abs().max()) > 0

    def rel(a, b):
        return float((a - b).norm() / b.norm())
    # batch independence: sample 1 alone gives what it gives inside the batch of 2 (a coupling bug gives O(1); bf16 accumulation-order
    # differences between the M = 4250 and M = 8500 GEMM plans give ~1e-2 after 24 blocks)
    solo = teacher.forward(x[1:], t[1:], c[1:], p[1:])
    assert rel(solo[0], out[1]) < 5e-2, rel(solo[0], out[1])
    assert rel(out[0], out[1]) > 0.5                                   # (the two samples really are different)
    # LoRA with B = 0 (peft init): the student is the teacher
    stu = MMDiT(W, lora).forward(x, t, c, p)
    assert rel(stu, out) < 5e-2, rel(stu, out)


def test_one_distillation_step(model):
    from pcm_amd.trainer_sd3 import SD3Distiller, SD3StepConfig
    cfg, W, lora = model
    dev = W.device
    D = SD3Distiller(W, lora, SD3StepConfig(multiphase=2, num_euler_timesteps=100, learning_rate=5e-6, adam_weight_decay=1e-3))
    x, _, c, p = _inputs(2, dev, seed=1)
    _, _, uc, up = _inputs(2, dev, seed=2)
    noise = torch.randn(2, 16, 128, 128, device=dev)
    p0 = lora.params.clone()
    out = D.step(x, c, p, uc, up, noise, torch.tensor([7, 93], device=dev))
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out["loss"]).all()) and float(out["loss"]) > 0
    assert float(out["grad_sumsq"]) > 0 and not torch.equal(lora.params, p0)
    m = lora.modules["transformer_blocks.0.attn.to_q"]
    assert float(m.gB.abs().max()) > 0                                           # dB = s dy^T (x A^T) is non-zero even with B = 0
    assert float(m.gA[32:].abs().max()) == 0.0 and float(m.A[32:].abs().max()) == 0.0     # the rank padding stays inert
    print("SD3 full-size step: loss %.5f, grad norm %.4e, peak %.1f GB" % (float(out["loss"]), float(out["grad_sumsq"]) ** 0.5,
                                                                          torch.cuda.max_memory_allocated() / 1e9))
