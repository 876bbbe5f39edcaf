"""SD3-medium at its real size (24 layers x 1536, 128x128x16 latents -> 4096 image + 154 text tokens) on the MI355X, checked through
size-independent properties (the fp32 oracle of a 2 B-parameter model is not a seconds-scale CPU job): finiteness, batch independence
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
    del sd
    torch.cuda.empty_cache()
    lora = sd3_lora_state(cfg, 32, 8.0, dev, seed=1)                     # B = 0 (reference init)
    loss, gnorm = run_property_case(dev, cfg, W, lora, 128, 154)
    torch.cuda.synchronize()
    print("SD3 full-size step: loss %.5f, grad norm %.4e, peak %.1f GB" % (loss, gnorm, torch.cuda.max_memory_allocated() / 1e9))
