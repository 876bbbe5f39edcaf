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
    kw = dict(block_out_channels=(64, 128, 128, 128), cross_attention_dim=64, heads=2, norm_num_groups=32)
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
