"""GPU parity of the SDXL UNet wiring on a narrow config: forward, LoRA backward and one distillation step with
added conditioning vs the CPU oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfgs():
    from oracle import unet_sd15 as O
    from pcm_amd.unet_spec import UNetConfig
    kw = dict(block_out_channels=(64, 128, 128), cross_attention_dim=64, heads=(1, 2, 2), down_attn=(False, True, True),
              transformer_depth=(1, 2, 3), use_linear_projection=True, addition_time_embed_dim=32,
              projection_class_embeddings_input_dim=64 + 6 * 32)
    return O.UNetConfig(**kw), UNetConfig(**kw)


def sdxl_topology_step_case():
    """one distillation step of the narrow SDXL-topology UNet (added conditioning) on the GPU against the live fp32 oracle: returns the
    report (relative errors per tensor, loss, LoRA-gradient cosine / norm ratio); shared with tests/test_gpu_fp16.py (the half build)"""
    from oracle import pcm_step as OS
    from oracle import unet_sd15 as O
    from pcm_amd import capi
    from pcm_amd.model import LoraState, UNetWeights
    from pcm_amd.trainer import Distiller, StepConfig
    capi.set_lib(None)
    capi.lib()
    oc, pc = _cfgs()
    sd = O.init_state_dict(oc, 0)
    W = UNetWeights(pc, sd, "cuda")
    lora = LoraState(pc, 64, 8.0, "cuda", seed=1, b_std=0.05)
    olora = {p: (lora.A_peft(m).detach().cpu().clone().requires_grad_(True), m.B.detach().cpu().clone().requires_grad_(True))
             for p, m in lora.modules.items()}
    ocfg = OS.StepConfig(multiphase=4, loss_type="huber", w_min=6.0, w_max=7.0, num_ddim_timesteps=40)
    B = 2
    inp = OS.draw_inputs(B, ocfg, seed=7, latent_hw=32, ctx_len=77, ctx_dim=64)
    g = torch.Generator().manual_seed(3)
    tids = torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * B)
    inp["added_cond"] = dict(text_embeds=torch.randn(B, 64, generator=g), time_ids=tids)
    inp["uncond_added_cond"] = dict(text_embeds=torch.zeros(B, 64), time_ids=tids)
    ref = OS.distill_step_forward(oc, sd, olora, inp, ocfg)
    ref["loss"].backward()
    cfg = StepConfig(multiphase=4, loss_type="huber", w_min=6.0, w_max=7.0, num_ddim_timesteps=40)
    D = Distiller(W, lora, cfg)
    gsc = float(D.loss_scale_dev.item()) if D.loss_scale_dev is not None else 1.0      # half build: the gradient buffers hold S * grad
    cu = lambda v: {k: x.cuda() for k, x in v.items()} if isinstance(v, dict) else v.cuda()
    out = D.forward_backward(*(cu(inp[k]) for k in ("latents", "prompt_embeds", "uncond_prompt_embeds", "noise", "index", "w")),
                             added_cond=cu(inp["added_cond"]), uncond_added_cond=cu(inp["uncond_added_cond"]))
    torch.cuda.synchronize()
    rep = {}
    for k in ("noise_pred", "uncond_teacher_output", "x_prev", "target"):
        r = ref[k].detach().float()
        rep[k] = float((out[k].float().cpu() - r).norm() / r.norm())
    rep["loss"], rep["loss_oracle"] = float(out["loss"].item()), float(ref["loss"].detach())
    rep["loss_rel"] = abs(rep["loss"] - rep["loss_oracle"]) / abs(rep["loss_oracle"])
    mine = torch.cat([t.reshape(-1).cpu() for m in lora.modules.values() for t in (lora.gA_peft(m), m.gB)]).double() / gsc
    refg = torch.cat([t.grad.reshape(-1) for p in lora.modules for t in olora[p]]).double()
    rep["grad_cos"] = float((mine * refg).sum() / (mine.norm() * refg.norm()))
    rep["grad_norm_ratio"] = float(mine.norm() / refg.norm())
    print("sdxl-topology step:", {k: "%.4g" % v for k, v in rep.items()})
    return rep, dict(pc=pc, W=W, cfg=cfg, ocfg=ocfg, inp=inp, cu=cu, B=B)


def test_sdxl_topology_step_vs_oracle():
    from oracle import pcm_step as OS
    from pcm_amd.model import LoraState
    from pcm_amd.trainer import Distiller
    rep, c = sdxl_topology_step_case()
    pc, W, cfg, ocfg, inp, cu, B = (c[k] for k in ("pc", "W", "cfg", "ocfg", "inp", "cu", "B"))
    for k in ("noise_pred", "uncond_teacher_output", "x_prev", "target"):
        assert rep[k] < 3e-2, (k, rep)
    assert rep["loss_rel"] < 5e-2, rep
    # The Huber cotangent of this NARROW config makes the LoRA gradient a sum of cancelling terms (bf16 noise 10-25 % of the norm, as in
    # tests/test_emu_adv.py); the backward wiring itself is checked with a random cotangent at 2.4 % in tests/test_emu_unet.py.
    # Here: direction and magnitude.
    assert rep["grad_cos"] > 0.93 and 0.85 < rep["grad_norm_ratio"] < 1.15, rep
    # the same step through hipGraph replay (text_time conditioning as static graph inputs): three replays on three different input sets
    # must reproduce the eager step() on a twin trainer with the same state
    lora_g = LoraState(pc, 64, 8.0, "cuda", seed=1, b_std=0.05)
    lora_e = LoraState(pc, 64, 8.0, "cuda", seed=1, b_std=0.05)
    Dg, De = Distiller(W, lora_g, cfg), Distiller(W, lora_e, cfg)
    ac, uac = cu(inp["added_cond"]), cu(inp["uncond_added_cond"])
    Dg.capture(B, H=32, W=32, ctx_len=77, ctx_dim=64, added_cond=ac, uncond_added_cond=uac)
    for rep in range(3):
        inp2 = OS.draw_inputs(B, ocfg, seed=20 + rep, latent_hw=32, ctx_len=77, ctx_dim=64)
        a6 = [cu(inp2[k]) for k in ("latents", "prompt_embeds", "uncond_prompt_embeds", "noise", "index", "w")]
        ac2 = dict(text_embeds=ac["text_embeds"] * (1.0 + 0.1 * rep), time_ids=ac["time_ids"])
        og = Dg.step_graphed(*a6, added_cond=ac2, uncond_added_cond=uac)
        lg = float(og["loss"])
        oe = De.step(*a6, added_cond=ac2, uncond_added_cond=uac)
        # (first step: identical state, identical arithmetic; later steps run on parameters that already differ by the atomics-order noise of
        # the earlier gradients -- the bound on them below -- which moves the loss by up to a few 1e-4 relative: seen 1.7e-4 once in ~10 runs)
        tol = 1e-6 if rep == 0 else 5e-4
        assert abs(lg - float(oe["loss"])) <= tol * abs(float(oe["loss"])), (rep, lg, float(oe["loss"]))
        if rep == 0:      # identical state on both trainers: the gradient buffers may differ by the order of fp32 atomics only
            from test_gpu_bench_config import assert_grads_match_per_module
            assert_grads_match_per_module(lora_g, lora_g.grads, lora_e.grads)
    rel = float((lora_g.params - lora_e.params).norm() / lora_e.params.norm())
    assert rel < 2e-4, rel      # a fraction of one lr-sized Adam step (atomics-order noise on near-zero gradient entries), see test_gpu_adv.py


@pytest.mark.parametrize("global_step", [0, 1])
def test_sdxl_adv_step_vs_oracle(global_step):
    """SDXL adversarial D / G step (1x1-conv heads on the down + mid taps, added conditioning) on the GPU vs the oracle."""
    from oracle import pcm_step as OS
    from oracle import unet_sd15 as O
    from pcm_amd import capi
    from pcm_amd.discriminator import Discriminator
    from pcm_amd.model import LoraState, UNetWeights
    from pcm_amd.trainer import AdvDistiller, StepConfig
    capi.set_lib(None)
    capi.lib()
    oc, pc = _cfgs()
    sd = O.init_state_dict(oc, 0)
    W = UNetWeights(pc, sd, "cuda")
    lora = LoraState(pc, 64, 8.0, "cuda", seed=1, b_std=0.05)
    disc = Discriminator((64, 128, 128, 128), num_h_per_head=1, device="cuda", seed=2, ksize=1, taps="down_mid")
    olora = {p: (lora.A_peft(m).detach().cpu().clone(), m.B.detach().cpu().clone()) for p, m in lora.modules.items()}
    dsd = {k: v.cpu() for k, v in disc.state_dict().items()}
    ocfg = OS.StepConfig(multiphase=4, loss_type="huber", w_min=6.0, w_max=7.0, num_ddim_timesteps=40)
    B = 2
    inp = OS.draw_inputs(B, ocfg, seed=7, latent_hw=32, ctx_len=77, ctx_dim=64)
    g = torch.Generator().manual_seed(3)
    tids = torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * B)
    inp["added_cond"] = dict(text_embeds=torch.randn(B, 64, generator=g), time_ids=tids)
    inp["uncond_added_cond"] = dict(text_embeds=torch.zeros(B, 64), time_ids=tids)
    inp["noise_fake"], inp["noise_real"] = torch.randn(B, 4, 32, 32, generator=g), torch.randn(B, 4, 32, 32, generator=g)
    inp["adv_u"] = torch.rand(B, generator=g)
    ref = OS.distill_step_adv(oc, sd, olora, dsd, inp, ocfg, global_step, adv_weight=0.1, taps="down_mid")
    cfg = StepConfig(multiphase=4, loss_type="huber", w_min=6.0, w_max=7.0, num_ddim_timesteps=40, learning_rate=0.0)
    D = AdvDistiller(W, lora, cfg, disc, adv_weight=0.1, adv_lr=0.0)
    cu = lambda v: {k: x.cuda() for k, x in v.items()} if isinstance(v, dict) else v.cuda()
    out = D.step_adv(global_step, *(cu(inp[k]) for k in ("latents", "prompt_embeds", "uncond_prompt_embeds", "noise", "index", "w", "noise_fake",
                                                           "noise_real", "adv_u")), added_cond=cu(inp["added_cond"]), uncond_added_cond=cu(inp["uncond_added_cond"]))
    torch.cuda.synchronize()
    assert torch.equal(out["adv_timesteps"].cpu(), ref["adv_timesteps"])
    if global_step % 2 == 0:
        dl, rdl = out["d_loss"].item(), float(ref["d_loss"])
        mine, refg, cnt = [], [], {}
        for k, hd in disc.heads:
            h = cnt.get(k, 0); cnt[k] = h + 1
            for n, t in hd.g.items():
                mine.append(t.reshape(-1).cpu()); refg.append(ref["head_grads"][f"heads.{k}.{h}.{n}"].reshape(-1))
        mine, refg = torch.cat(mine).double(), torch.cat(refg).double()
        cos = float((mine * refg).sum() / (mine.norm() * refg.norm()))
        print("sdxl D step: d_loss %.5f / %.5f, head-grad cos %.4f" % (dl, rdl, cos))
        assert abs(dl - rdl) < 3e-2 * abs(rdl) and cos > 0.95
    else:
        mine = torch.cat([t.reshape(-1).cpu() for m in lora.modules.values() for t in (lora.gA_peft(m), m.gB)]).double()
        refg = torch.cat([g_.reshape(-1) for g_ in ref["lora_grads"]]).double()
        cos = float((mine * refg).sum() / (mine.norm() * refg.norm()))
        print("sdxl G step: loss_cm %.5f / %.5f, g_loss %.5f / %.5f, lora-grad cos %.4f" % (out["loss_cm"].item(), float(ref["loss_cm"]),
              out["g_loss"].item(), float(ref["g_loss"]), cos))
        assert abs(out["loss_cm"].item() - float(ref["loss_cm"])) < 5e-2 * abs(float(ref["loss_cm"]))
        assert abs(out["g_loss"].item() - float(ref["g_loss"])) < 3e-2 * abs(float(ref["g_loss"]))
        assert cos > 0.93
