"""GPU parity of the adversarial step (D and G updates) at the real SD1.5 size vs the CPU oracle (bs 1)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("global_step", [0, 1])
def test_adv_step_full_size(global_step):
    from oracle import pcm_step as OS
    from oracle import unet_sd15 as O
    from pcm_amd import capi
    from pcm_amd.discriminator import Discriminator
    from pcm_amd.model import LoraState, UNetWeights
    from pcm_amd.trainer import AdvDistiller, StepConfig
    from pcm_amd.unet_spec import UNetConfig
    capi.set_lib(None)
    capi.lib()
    oc = O.UNetConfig.sd15()
    sd = O.init_state_dict(oc, 0)
    W = UNetWeights(UNetConfig.sd15(), sd, "cuda")
    lora = LoraState(UNetConfig.sd15(), 64, 8.0, "cuda", seed=1, b_std=0.02)
    disc = Discriminator(device="cuda", seed=2, num_h_per_head=1)      # 9 heads (one per feature) keeps the CPU oracle affordable
    ocfg = OS.StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0)
    B = 1
    inp = OS.draw_inputs(B, ocfg, seed=11)
    inp["index"] = torch.tensor([30])
    g = torch.Generator().manual_seed(9)
    inp["noise_fake"], inp["noise_real"] = torch.randn(B, 4, 64, 64, generator=g), torch.randn(B, 4, 64, 64, generator=g)
    inp["adv_u"] = torch.rand(B, generator=g)
    olora = {p: (lora.A_peft(m).detach().cpu().clone(), m.B.detach().cpu().clone()) for p, m in lora.modules.items()}
    dsd = {k: v.cpu() for k, v in disc.state_dict().items()}
    ref = OS.distill_step_adv(oc, sd, olora, dsd, inp, ocfg, global_step, adv_weight=0.1)
    cfg = StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0, learning_rate=0.0)
    D = AdvDistiller(W, lora, cfg, disc, adv_weight=0.1, adv_lr=0.0)
    dev = {k: v.cuda() for k, v in inp.items()}
    out = D.step_adv(global_step, dev["latents"], dev["prompt_embeds"], dev["uncond_prompt_embeds"], dev["noise"], dev["index"], dev["w"],
                     dev["noise_fake"], dev["noise_real"], dev["adv_u"])
    torch.cuda.synchronize()
    rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / (b.double().norm() + 1e-6 * b.numel() ** 0.5))
    assert torch.equal(out["adv_timesteps"].cpu(), ref["adv_timesteps"])
    assert rel(out["fake_adv"], ref["fake_adv"]) < 3e-2
    if global_step % 2 == 0:
        dl, rdl = out["d_loss"].item(), float(ref["d_loss"])
        mine, refg, cnt = [], [], {}
        for k, hd in disc.heads:
            h = cnt.get(k, 0); cnt[k] = h + 1
            for n, t in hd.g.items():
                v = t.permute(0, 3, 1, 2) if n in ("conv1.0.weight", "conv2.0.weight") else t
                mine.append(v.reshape(-1).cpu()); refg.append(ref["head_grads"][f"heads.{k}.{h}.{n}"].reshape(-1))
        mine, refg = torch.cat(mine), torch.cat(refg)
        cos = float((mine.double() * refg.double()).sum() / (mine.double().norm() * refg.double().norm()))
        print("D step: d_loss %.5f / %.5f, head-grad rel %.3e cos %.4f" % (dl, rdl, rel(mine, refg), cos))
        assert abs(dl - rdl) < 2e-2 * abs(rdl) and cos > 0.97
    else:
        mine = torch.cat([t.reshape(-1).cpu() for m in lora.modules.values() for t in (lora.gA_peft(m), m.gB)])
        refg = torch.cat([g_.reshape(-1) for g_ in ref["lora_grads"]])
        cos = float((mine.double() * refg.double()).sum() / (mine.double().norm() * refg.double().norm()))
        print("G step: loss_cm %.5f / %.5f, g_loss %.5f / %.5f, lora-grad rel %.3e cos %.4f" % (out["loss_cm"].item(), float(ref["loss_cm"]),
              out["g_loss"].item(), float(ref["g_loss"]), rel(mine, refg), cos))
        assert abs(out["loss_cm"].item() - float(ref["loss_cm"])) < 2e-2 * abs(float(ref["loss_cm"]))
        assert abs(out["g_loss"].item() - float(ref["g_loss"])) < 2e-2 * abs(float(ref["g_loss"]))
        assert cos > 0.95


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
        assert abs(lg - le) <= 1e-5 * abs(le) + 1e-9, (step, key, lg, le)
    rel = lambda x, y: float((x - y).norm() / (y.norm() + 1e-30))
    # Adam divides by sqrt(v): on entries whose gradient is at the fp32-atomics noise level the update direction itself is noise, so the
    # states agree to a fraction of one lr-sized step (lr 1e-4 against |param| ~ 2e-2), not bitwise
    assert rel(Dg.lora.params, De.lora.params) < 2e-4 and rel(Dg.disc.params, De.disc.params) < 2e-4
    assert Dg.step_count == De.step_count == 2
