"""Inference sampler on the host emulator: step kernel and the full CFG sampling loop vs the oracle restatement."""
import pytest
import torch

from emu_lib import emu_lib
from pcm_amd import capi, ops


@pytest.fixture(autouse=True)
def _use_emu():
    capi.set_lib(emu_lib())
    yield
    capi.set_lib(None)


def test_trailing_timesteps_match_the_restated_scheduler():
    from oracle import pcm_math as PM
    from pcm_amd.sampler import trailing_timesteps
    for n in (1, 2, 4, 8, 16, 50, 3, 7):
        assert trailing_timesteps(n) == PM.ddim_trailing_timesteps(n), n
    assert trailing_timesteps(4) == [999, 749, 499, 249]


def test_sampler_step_kernel_vs_oracle():
    from oracle import pcm_math as PM
    acp = PM.sd15_alphas_cumprod()
    g = torch.Generator().manual_seed(3)
    x, ec, eu = (torch.randn(2, 4, 8, 8, generator=g) for _ in range(3))
    for t, n in ((999, 4), (249, 4), (124, 8)):
        prev = t - 1000 // n
        a_prev = float(acp[prev]) if prev >= 0 else float(acp[0])
        got = ops.sampler_ddim_step(ec, eu, x, float(acp[t]), a_prev, 7.5)
        ref = PM.ddim_sampler_step(eu + 7.5 * (ec - eu), t, x, acp, n)
        assert torch.allclose(got, ref, rtol=1e-5, atol=1e-5), (t, (got - ref).abs().max())
        got1 = ops.sampler_ddim_step(ec, None, x, float(acp[t]), a_prev, 1.0)
        assert torch.allclose(got1, PM.ddim_sampler_step(ec, t, x, acp, n), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("guidance", [1.0, 7.5])
def test_sampling_loop_vs_oracle(guidance):
    from oracle import pcm_math as PM
    from oracle import unet_sd15 as O
    from pcm_amd.model import LoraState, UNet, UNetWeights
    from pcm_amd.sampler import DDIMTrailingSampler
    from pcm_amd.unet_spec import UNetConfig
    kw = dict(block_out_channels=(64, 128), layers_per_block=1, cross_attention_dim=64, heads=2, norm_num_groups=32)
    oc, pc = O.UNetConfig(**kw), UNetConfig(**kw)
    sd = O.init_state_dict(oc, 0)
    W = UNetWeights(pc, sd, "cpu")
    lora = LoraState(pc, 64, 8.0, "cpu", seed=1, b_std=0.05)
    olora = {p: (lora.A_peft(m).clone(), m.B.clone()) for p, m in lora.modules.items()}
    g = torch.Generator().manual_seed(5)
    B = 2
    lat = torch.randn(B, 4, 8, 8, generator=g)
    ctx, unc = torch.randn(B, 7, 64, generator=g), torch.randn(B, 7, 64, generator=g)
    acp = PM.sd15_alphas_cumprod()
    with torch.no_grad():
        ref = PM.ddim_sample(lambda x, t, c: O.unet_forward(oc, sd, x, t, c, olora, 8.0), ctx, unc, lat, 2, guidance, acp)
    out = DDIMTrailingSampler(UNet(W, lora)).sample(ctx, unc, num_inference_steps=2, guidance_scale=guidance, latents=lat)
    rel = float((out - ref).norm() / ref.norm())
    print("sampled latents rel err %.3e (guidance %.1f)" % (rel, guidance))
    assert rel < (3e-2 if guidance == 1.0 else 8e-2)


def test_sdxl_sampling_loop_with_added_conditioning_vs_oracle():
    """the SDXL script's validation loop: same DDIM-trailing sampler, UNet with text_time added conditioning, zero negative embeds."""
    from oracle import pcm_math as PM
    from oracle import unet_sd15 as O
    from pcm_amd.model import LoraState, UNet, UNetWeights
    from pcm_amd.sampler import DDIMTrailingSampler
    from pcm_amd.unet_spec import UNetConfig
    kw = dict(block_out_channels=(64, 128), cross_attention_dim=64, heads=(1, 2), down_attn=(False, True), transformer_depth=(1, 2),
              use_linear_projection=True, addition_time_embed_dim=32, projection_class_embeddings_input_dim=64 + 6 * 32, layers_per_block=1)
    oc, pc = O.UNetConfig(**kw), UNetConfig(**kw)
    sd = O.init_state_dict(oc, 0)
    W = UNetWeights(pc, sd, "cpu")
    lora = LoraState(pc, 64, 8.0, "cpu", seed=1, b_std=0.05)
    olora = {p: (lora.A_peft(m).clone(), m.B.clone()) for p, m in lora.modules.items()}
    g = torch.Generator().manual_seed(6)
    B = 2
    lat = torch.randn(B, 4, 8, 8, generator=g)
    ctx, unc = torch.randn(B, 7, 64, generator=g), torch.zeros(B, 7, 64)
    tids = torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * B)
    ac = dict(text_embeds=torch.randn(B, 64, generator=g), time_ids=tids)
    uac = dict(text_embeds=torch.zeros(B, 64), time_ids=tids)
    acp = PM.sd15_alphas_cumprod()

    def unet_fn(x, t, c):      # the oracle loop calls the positive branch with ctx and the negative one with unc
        return O.unet_forward(oc, sd, x, t, c, olora, 8.0, added_cond=ac if c is ctx else uac)
    for guidance in (1.0, 5.0):
        with torch.no_grad():
            ref = PM.ddim_sample(unet_fn, ctx, unc, lat, 2, guidance, acp)
        out = DDIMTrailingSampler(UNet(W, lora)).sample(ctx, unc, num_inference_steps=2, guidance_scale=guidance, latents=lat,
                                                        added_cond=ac, uncond_added_cond=uac)
        rel = float((out - ref).norm() / ref.norm())
        print("SDXL sampled latents rel err %.3e (guidance %.1f)" % (rel, guidance))
        assert rel < (3e-2 if guidance == 1.0 else 8e-2)
