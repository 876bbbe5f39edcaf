"""Host-emulation end-to-end check of the adversarial step (discriminator + generator updates) on a tiny
SD1.5-topology config against the oracle's autograd (reference: train_pcm_lora_sd15_adv.py, discriminator_sd15.py)."""
import pytest
import torch

from emu_lib import emu_lib
from pcm_amd import capi


@pytest.fixture(autouse=True)
def _use_emu():
    capi.set_lib(emu_lib())
    yield
    capi.set_lib(None)


@pytest.mark.slow
@pytest.mark.parametrize("global_step", [0, 1])
def test_sdxl_adv_step_vs_oracle(global_step):
    """SDXL adversarial step (discriminator_sdxl.py: taps after the down blocks + mid only, one 1x1-conv head per tap; added
    conditioning through every UNet call) on a narrow SDXL-topology config."""
    _run_adv(global_step, sdxl=True)


@pytest.mark.slow
@pytest.mark.parametrize("global_step", [0, 1])
def test_adv_step_vs_oracle(global_step):
    _run_adv(global_step, sdxl=False)


def _run_adv(global_step, sdxl):
    from oracle import pcm_step as OS
    from oracle import unet_sd15 as O
    from pcm_amd.discriminator import Discriminator
    from pcm_amd.model import LoraState, UNetWeights
    from pcm_amd.trainer import AdvDistiller, StepConfig
    from pcm_amd.unet_spec import UNetConfig
    kw = dict(block_out_channels=(64, 128), layers_per_block=1, cross_attention_dim=64, heads=2, norm_num_groups=32)   # 2-level UNet: 5 tapped features
    if sdxl:
        kw = dict(block_out_channels=(64, 128), layers_per_block=1, cross_attention_dim=64, heads=(1, 2), norm_num_groups=32, down_attn=(False, True),
                  transformer_depth=(1, 2), use_linear_projection=True, addition_time_embed_dim=32, projection_class_embeddings_input_dim=64 + 6 * 32)
    oc, pc = O.UNetConfig(**kw), UNetConfig(**kw)
    sd = O.init_state_dict(oc, 0)
    W = UNetWeights(pc, sd, "cpu")
    lora = LoraState(pc, 64, 8.0, "cpu", seed=1, b_std=0.05)
    dims = (64, 128, 128, 128, 64)    # feature widths of this UNet (down x2, mid, up x2)
    disc = Discriminator(dims, num_h_per_head=1, device="cpu", seed=2)
    taps = True
    if sdxl:
        taps = "down_mid"
        disc = Discriminator((64, 128, 128), num_h_per_head=1, device="cpu", seed=2, ksize=1, taps=taps)
    ocfg = OS.StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0)
    B = 2
    inp = OS.draw_inputs(B, ocfg, seed=5, latent_hw=8, ctx_len=7, ctx_dim=64)
    g = torch.Generator().manual_seed(9)
    inp["noise_fake"] = torch.randn(B, 4, 8, 8, generator=g)
    inp["noise_real"] = torch.randn(B, 4, 8, 8, generator=g)
    inp["adv_u"] = torch.rand(B, generator=g)
    ac = uac = None
    if sdxl:
        tids = torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * B)
        ac = inp["added_cond"] = dict(text_embeds=torch.randn(B, 64, generator=g), time_ids=tids)
        uac = inp["uncond_added_cond"] = dict(text_embeds=torch.zeros(B, 64), time_ids=tids)
    olora = {p: (lora.A_peft(m).clone(), m.B.clone()) for p, m in lora.modules.items()}
    dsd = disc.state_dict()
    ref = OS.distill_step_adv(oc, sd, olora, {k.replace("heads.", "heads."): v for k, v in dsd.items()}, inp, ocfg, global_step, adv_weight=0.1, taps=taps)
    # the oracle's discriminator_forward expects nh heads per feature: match num_h_per_head=1
    cfg = StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0, learning_rate=0.0)
    D = AdvDistiller(W, lora, cfg, disc, adv_weight=0.1, adv_lr=0.0)
    p_lora, p_disc = lora.params.clone(), disc.params.clone()
    out = D.step_adv(global_step, inp["latents"], inp["prompt_embeds"], inp["uncond_prompt_embeds"], inp["noise"], inp["index"], inp["w"],
                     inp["noise_fake"], inp["noise_real"], inp["adv_u"], added_cond=ac, uncond_added_cond=uac)
    assert torch.equal(out["adv_timesteps"], ref["adv_timesteps"])
    rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-6 * b.numel() ** 0.5))
    assert rel(out["fake_adv"], ref["fake_adv"]) < 3e-2
    if global_step % 2 == 0:
        assert abs(out["d_loss"].item() - float(ref["d_loss"])) < 3e-2 * abs(float(ref["d_loss"]))
        mine = disc.state_dict()
        gstate = {}
        cnt = {}
        for k, hd in disc.heads:
            h = cnt.get(k, 0); cnt[k] = h + 1
            for n, t in hd.g.items():
                v = t
                if n in ("conv1.0.weight", "conv2.0.weight"):
                    v = v.permute(0, 3, 1, 2) if disc.ksize == 3 else v.view(hd.C, hd.C, 1, 1)
                elif n == "conv_out.weight":
                    v = v.view(1, hd.C, 1, 1)
                gstate[f"heads.{k}.{h}.{n}"] = v
        num = den = 0.0
        for n, gr in ref["head_grads"].items():
            num += float(((gstate[n] - gr) ** 2).sum()); den += float((gr ** 2).sum())
            assert rel(gstate[n], gr) < 0.4, (n, rel(gstate[n], gr))   # 1x1-pixel features + 4-element GN groups are bf16-noisy
        print("d step: d_loss %.5f (oracle %.5f), head-grad rel err %.3e" % (out["d_loss"].item(), float(ref["d_loss"]), (num / den) ** 0.5))
        # the hinge gradient is +1/n on fake rows and -1/n on real rows: every parameter gradient is a heavily
        # cancelling sum, and on this tiny config (1x1..8x8 maps, 4-channel GN groups) bf16 rounding of the summands
        # shows up as 10-30 % relative error; tests/kernel_cases.py::case_discriminator_heads checks the same code
        # with non-cancelling cotangents at < 10 %.  Here: direction + magnitude.
        mine_all = torch.cat([gstate[n].reshape(-1) for n in ref["head_grads"]])
        ref_all = torch.cat([gr.reshape(-1) for gr in ref["head_grads"].values()])
        cos = float((mine_all * ref_all).sum() / (mine_all.norm() * ref_all.norm()))
        assert cos > 0.97 and (num / den) ** 0.5 < 0.25, (cos, (num / den) ** 0.5)
        assert torch.equal(lora.params, p_lora)            # the student is untouched on discriminator steps
    else:
        assert abs(out["loss_cm"].item() - float(ref["loss_cm"])) < 3e-2 * abs(float(ref["loss_cm"]))
        assert abs(out["g_loss"].item() - float(ref["g_loss"])) < 3e-2 * abs(float(ref["g_loss"]))
        mine = torch.cat([t.reshape(-1) for m in lora.modules.values() for t in (lora.gA_peft(m), m.gB)])
        refg = torch.cat([g_.reshape(-1) for g_ in ref["lora_grads"]])
        print("g step: loss_cm %.5f/%.5f g_loss %.5f/%.5f lora-grad rel err %.3e" % (out["loss_cm"].item(), float(ref["loss_cm"]),
              out["g_loss"].item(), float(ref["g_loss"]), rel(mine, refg)))
        # with every logit on the active side of the hinge the cotangent is CONSTANT over pixels and GroupNorm's
        # backward removes exactly that component: the adversarial part of the gradient is a near-total cancellation
        # on this tiny config and carries ~30 % bf16 noise (measured: d loss/d fake_adv rel err 0.30 with zero hinge
        # flips); the same code path with generic cotangents is checked at 4 % (teacher input gradient) and < 10 %
        # (heads) in tests/kernel_cases.py / tools.  Here: losses tight, gradient direction + magnitude.
        cos = float((mine.double() * refg.double()).sum() / (mine.double().norm() * refg.double().norm()))
        assert cos > 0.95 and rel(mine, refg) < 0.35, (cos, rel(mine, refg))
        assert torch.equal(disc.params, p_disc)            # the heads are untouched on generator steps


@pytest.mark.slow
@pytest.mark.parametrize("global_step", [0, 1])
def test_adv_step_four_heads_per_tap_real_learning_rates(global_step):
    """the BASELINE configs[2] SHAPE on the emulator: four heads per tapped feature, batch 2, non-zero lr / adv_lr; compares the head / LoRA
    updates with the oracle's clip + AdamW (tests/adv_cases.py; the full-size run is tests/test_gpu_adv.py)."""
    import adv_cases as A
    kw = dict(block_out_channels=(64, 128), layers_per_block=1, cross_attention_dim=64, heads=2, norm_num_groups=32)
    rep = A.case_adv_c3("cpu", kw, (64, 128, 128, 128, 64), 2, 8, 7, 64, global_step)      # 8x8 latents: 17 s per step on the emulator
    assert rep["heads"] == 20 and rep["fake_adv"] < 5e-3
    if global_step % 2 == 0:
        # (2x2 .. 8x8 feature maps: the cosine is a noisy number here -- 0.9924 / 0.9892 for two GELU evaluations that differ by 3e-5;
        # at the real size tests/test_gpu_adv.py asserts 0.99 / 0.98 on measured 0.9963 / 0.9904)
        assert rep["d_loss_rel"] < 5e-3 and rep["lora_untouched"] and rep["head_grad_cos"] > 0.985 and min(rep["head_grad_cos_per_tap"]) > 0.975
        assert rep["head_update_cos"] > 0.9 and abs(rep["head_update_norm_ratio"] - 1) < 1e-2
    else:
        assert rep["loss_cm_rel"] < 2e-2 and rep["g_loss_rel"] < 5e-3 and rep["heads_untouched"]
        # (2x2 .. 8x8 feature maps: the LoRA gradient is a few-hundred-term cancelling sum here -- measured cos 0.952, update cos 0.811;
        # at the real size tests/test_gpu_adv.py asserts 0.95 / 0.85 on measured 0.9987 / 0.944)
        assert rep["lora_grad_cos"] > 0.93 and rep["lora_update_cos"] > 0.75 and abs(rep["lora_update_norm_ratio"] - 1) < 1e-2


@pytest.mark.slow
def test_segmented_capture_cuts_at_every_collective(monkeypatch):
    """AdvDistiller.capture_adv at world_size > 1 cuts the captured step where the eager path issues a collective (trainer.SegmentedGraph /
    Distiller._collective).  With a recording stand-in for torch.cuda.CUDAGraph the cut placement is checked on the CPU: the D step of a
    5-tap discriminator becomes graph, [bucket all-reduce, graph] x 5, [wait for the buckets], graph; the G step graph, [LoRA exchange],
    graph -- and nothing communicates while capturing."""
    from oracle import pcm_step as OS
    from oracle import unet_sd15 as O
    from pcm_amd import trainer as T
    from pcm_amd.discriminator import Discriminator
    from pcm_amd.model import LoraState, UNetWeights
    from pcm_amd.unet_spec import UNetConfig

    class FakeGraph:
        def __init__(self):
            self.log = []

        def capture_begin(self, **kw):
            self.log.append(("begin", kw.get("pool")))

        def capture_end(self):
            self.log.append(("end", None))

        def pool(self):
            return "pool"

        def replay(self):
            self.log.append(("replay", None))
    monkeypatch.setattr(torch.cuda, "CUDAGraph", FakeGraph)
    monkeypatch.setattr(T, "SEG_FORCE", True)          # world_size 1: every collective is a no-op host action, the cuts are the same
    kw = dict(block_out_channels=(64, 128), layers_per_block=1, cross_attention_dim=64, heads=2, norm_num_groups=32)
    oc, pc = O.UNetConfig(**kw), UNetConfig(**kw)
    W = UNetWeights(pc, O.init_state_dict(oc, 0), "cpu")
    lora = LoraState(pc, 64, 8.0, "cpu", seed=1, b_std=0.05)
    disc = Discriminator((64, 128, 128, 128, 64), num_h_per_head=2, device="cpu", seed=2)
    D = T.AdvDistiller(W, lora, T.StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0, learning_rate=1e-4), disc, adv_weight=0.1, adv_lr=1e-4)
    ocfg = OS.StepConfig(multiphase=2, loss_type="huber", w_min=4.0, w_max=5.0)
    inp = OS.draw_inputs(2, ocfg, seed=5, latent_hw=8, ctx_len=7, ctx_dim=64)
    g = torch.Generator().manual_seed(9)
    extra = [torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 4, 8, 8, generator=g), torch.rand(2, generator=g)]
    args = [inp[k] for k in ("latents", "prompt_embeds", "uncond_prompt_embeds", "noise", "index", "w")] + extra
    calls = []
    monkeypatch.setattr(D, "_disc_bucket", lambda a, b: calls.append(("bucket", a, b)))
    monkeypatch.setattr(D, "_disc_finish_exchange", lambda: calls.append(("finish",)))
    monkeypatch.setattr(D, "all_reduce_grads", lambda: calls.append(("lora",)))
    shapes = {}
    for gs in (0, 1):
        D._seg = seg = T.SegmentedGraph()
        seg.begin()
        D.step_adv(gs, *args)
        seg.end()
        D._seg = None
        assert calls == [], "a collective ran during capture"
        kinds = ["g" if isinstance(it, FakeGraph) else "h" for it in seg.items]
        shapes[gs] = "".join(kinds)
        assert all(it.log[0][0] == "begin" and it.log[-1][0] == "end" for it in seg.items if isinstance(it, FakeGraph))
        assert [it.log[0][1] for it in seg.items if isinstance(it, FakeGraph)][1:] == ["pool"] * (kinds.count("g") - 1)   # one shared pool
        seg.replay()
        if gs == 0:
            assert [c[0] for c in calls] == ["bucket"] * 5 + ["finish"], calls
            offs = [(c[1], c[2]) for c in calls[:5]]
            assert all(b > a for a, b in offs) and sorted(offs) == offs and offs[0][0] == 0 and offs[-1][1] <= disc.numel   # tap order, disjoint ranges
        else:
            assert calls == [("lora",)], calls
        calls.clear()
    assert shapes == {0: "g" + "hg" * 6, 1: "ghg"}, shapes
    assert D.step_count == 0      # the captured pass does not advance the host-side step counter (step_adv_graphed does)


@pytest.mark.parametrize("global_step", [0, 1])
def test_adv_step_with_fp16_teacher_pass_keeps_the_discriminator_in_the_build_format(global_step):
    """AdvDistiller(teacher_weights = half packing): only the ODE-solver teacher pass (sd15_adv.py:1312, a dtype-less autocast) runs in IEEE
    half; the discriminator's feature passes through the frozen UNet -- which are back-propagated -- stay in the process's format.  So on
    the same inputs ``fake_adv`` (student only) is BITWISE the one-format step's, the target-side quantities move a little with the better
    teacher, the step applies, and the process is back in bfloat16 afterwards."""
    import adv_cases as A
    from oracle import unet_sd15 as O
    from pcm_amd import ops, precision
    from pcm_amd.model import UNetWeights
    from pcm_amd.trainer import AdvDistiller
    kw = dict(block_out_channels=(64, 128), layers_per_block=1, cross_attention_dim=64, heads=2, norm_num_groups=32)
    dims = (64, 128, 128, 128, 64)
    precision.set_precision("bf16", lib=emu_lib("bf16"))
    precision.register_lib("fp16", emu_lib("f16"))

    def run(split):
        oc, pc, lora, disc, ocfg, cfg, inp = A._setup("cpu", kw, dims, 2, 8, 7, 64, 1, 5e-6, None, 11)
        sd = O.init_state_dict(oc, 0)
        W, Wt = UNetWeights(pc, sd, "cpu"), None
        if split:
            with precision.format_scope("fp16"):
                Wt = UNetWeights(pc, sd, "cpu", need_bwd=False)
        D = AdvDistiller(W, lora, cfg, disc, adv_weight=0.1, adv_lr=1e-5, teacher_weights=Wt)
        p_l, p_d = lora.params.clone(), disc.params.clone()
        out = D.step_adv(global_step, inp["latents"], inp["prompt_embeds"], inp["uncond_prompt_embeds"], inp["noise"], inp["index"], inp["w"],
                         inp["noise_fake"], inp["noise_real"], inp["adv_u"])
        assert ops.BF16 == torch.bfloat16 and capi.lib().act_dtype == 0 and precision.precision() == "bf16"
        moved = (not torch.equal(lora.params, p_l), not torch.equal(disc.params, p_d))
        return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in out.items()}, moved

    try:
        a, moved_a = run(False)
        b, moved_b = run(True)
    finally:
        precision.set_precision("bf16", lib=emu_lib("bf16"))
    assert moved_a == moved_b == ((False, True) if global_step % 2 == 0 else (True, False))
    assert torch.equal(a["fake_adv"], b["fake_adv"]) and torch.equal(a["model_pred"], b["model_pred"])
    rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())   # noqa: E731
    assert 0 < rel(b["target"], a["target"]) < 2e-2
    key = "d_loss" if global_step % 2 == 0 else "g_loss"
    assert abs(float(b[key]) - float(a[key])) < 5e-2 * abs(float(a[key])) + 1e-6, (float(a[key]), float(b[key]))


def test_adversarial_steps_with_teacher_prefetch_are_the_same_steps():
    """AdvDistiller.step_adv(..., prefetch=next batch): the ODE-solver teacher pass of the next batch is issued ahead (side stream on the GPU,
    same call order here) and the next step picks its results up -- a D and a G step give bitwise the losses, heads and LoRA of plain steps."""
    import adv_cases as A
    from oracle import unet_sd15 as O
    from pcm_amd.model import UNetWeights
    from pcm_amd.trainer import AdvDistiller
    kw = dict(block_out_channels=(64, 128), layers_per_block=1, cross_attention_dim=64, heads=2, norm_num_groups=32)
    dims = (64, 128, 128, 128, 64)
    names = ("latents", "prompt_embeds", "uncond_prompt_embeds", "noise", "index", "w", "noise_fake", "noise_real", "adv_u")

    def run(prefetch):
        oc, pc, lora, disc, ocfg, cfg, inp0 = A._setup("cpu", kw, dims, 2, 8, 7, 64, 1, 1e-4, None, 11)
        inp1 = A._setup("cpu", kw, dims, 2, 8, 7, 64, 1, 1e-4, None, 12)[-1]
        W = UNetWeights(pc, O.init_state_dict(oc, 0), "cpu")
        D = AdvDistiller(W, lora, cfg, disc, adv_weight=0.1, adv_lr=1e-4)
        b0, b1 = [inp0[k] for k in names], [inp1[k] for k in names]
        d = D.step_adv(0, *b0, prefetch=tuple(b1[:6]) if prefetch else None)
        if prefetch:
            assert D._prefetched is not None
        g = D.step_adv(1, *b1)
        assert D._prefetched is None
        return float(d["d_loss"]), float(g["loss_cm"]), float(g["g_loss"]), lora.params.clone(), disc.params.clone()

    a, b = run(False), run(True)
    assert a[:3] == b[:3] and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
