"""The IEEE-half build of the kernel library (lib/libpcm_hip_f16.so, csrc/pcm_common.h -DPCM_ACT_F16; precision.set_precision("fp16")):
the reference's ``--mixed_precision=fp16`` (train_pcm_lora_sd15.sh:9) and the arithmetic in which BASELINE.json's parity target --
"loss curves matching reference to 1e-3 rel" -- is checked DIRECTLY against the plain fp32 oracle (half keeps 11 significand bits; the
bf16 build is bounded by its 8).  Oracle side: the committed fp32 fixtures of tests/step_golden_cases.py."""
import json
import math
import os

import pytest
import torch

import kernel_cases as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _half_build():
    from pcm_amd import capi, precision
    assert torch.cuda.is_available()
    # the TOOLS build of the half library: the kernel cases below force kernel families / read plan codes through pcm_debug_* hooks, which
    # the product libraries do not export; the step / trainer tests further down run the same kernels (default knobs = the product's)
    precision.set_precision("fp16", tools=True)          # raises if the library is missing: no fallback
    assert capi.lib().act_dtype == 1 and K.ops.BF16 == torch.float16
    yield
    torch.cuda.synchronize()
    precision.set_precision("bf16")
    assert capi.lib().act_dtype == 0 and K.ops.BF16 == torch.bfloat16


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


# ---- the kernels, against torch fp32 on the same half-rounded inputs (the bf16 cases and tolerances: a format slip -- a bf16 constant, a
# bf16 MFMA left in -- is a tens-of-percent error, cf. the first run of this build: the attention "ones" column held bf16 1.0)
def test_kernels_half_build():
    K.case_groupnorm("cuda", 2, 4096, 320, 32, 1)
    K.case_groupnorm("cuda", 2, 1024, 640, 32, 0, eps=1e-6)
    K.case_layernorm("cuda", 4099, 320)
    K.case_elementwise("cuda")
    K.case_edge_convs("cuda", B=2, H=64, W=64, C0=320)
    K.case_timestep_embedding("cuda")
    K.case_pack("cuda")
    K.case_wgrad_plain("cuda", 4096, 320, 320)
    K.case_wgrad_multi("cuda")
    K.case_wgrad_conv("cuda", 2, 16, 16, 320, 1, 0)
    K.case_lora_repack("cuda")
    assert K.case_gemm_n64("cuda", 8192, 1280) <= 0 and K.case_gemm_n64("cuda", 65536, 320) <= 0        # (these cases return the excess over tolerance)
    K.case_conv_r64("cuda", 4, 32, 32, 640)
    assert K.case_gemm_smallm("cuda", 16, 1280, (1280, 64), 0, True, False) <= 0 and K.case_gemm_smallm("cuda", 2, 9216, (1536,), 1, True, False) <= 0
    assert K.case_gemm_geglu("cuda") <= 0 and K.case_gemm_geglu("cuda", M=8192, K=1280, inner=5120) <= 0


@pytest.mark.parametrize("B,H,Lq,Lk,d", [(1, 8, 1024, 1024, 40), (2, 8, 256, 77, 80), (2, 8, 64, 64, 160), (1, 8, 4096, 4096, 40), (1, 10, 1100, 1100, 64)])
def test_attention_half_build(B, H, Lq, Lk, d):
    K.case_attention("cuda", B, H, Lq, Lk, d)


@pytest.mark.parametrize("B,H,Lq,Lk,d", [(2, 5, 1024, 77, 64), (2, 8, 256, 77, 160), (2, 8, 256, 256, 160), (1, 4, 200, 400, 32), (1, 8, 512, 512, 40)])
@pytest.mark.parametrize("gain", [4, 8, 16])
def test_attention_half_build_late_key_beyond_the_half_range(B, H, Lq, Lk, d, gain):
    """a late key ~25 .. 100 (log2 domain) above the first tile's row maximum: p exceeds 65504, inf once packed to IEEE half -- the one-time
    check must see it for the head dims whose row sum is the fp32 VALU sum too (SDXL d = 64 at Lk = 77, SD1.5 d = 160): round-5 advisor finding,
    emulator twin in tests/test_emu_fp16.py"""
    K.case_attention("cuda", B, H, Lq, Lk, d, spike=True, prescaled=True, spike_overflow=True, spike_gain=gain)


@pytest.mark.parametrize("family,which", [("big", w) for w in K.GEMM_BIG_CASES[:6]] + [("4w", w) for w in K.GEMM_4W_CASES[:3]])
def test_gemm_tiles_half_build(family, which):
    excess, err = (K.case_gemm_big if family == "big" else K.case_gemm_4w)("cuda", which)
    assert excess <= 0, (family, which, err)


def test_plain_gemm_half_precision_gain():
    """the same GEMM through the half build is ~8x closer to fp32-on-fp32-inputs than bf16 storage allows: output rounding 2^-12 vs 2^-9"""
    from pcm_amd import ops
    g = torch.Generator().manual_seed(3)
    M, N, Kd = 1024, 640, 1280
    x, w = torch.randn(M, Kd, generator=g), torch.randn(N, Kd, generator=g) * 0.03
    xh, wh = x.half().cuda(), w.half().cuda()
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    ops.gemm([ops.Seg(xh, wh)], M, N, out)
    ref = xh.float().cpu() @ wh.float().cpu().t()
    r = rel(out, ref)
    assert r < 4e-4, r      # half output rounding alone: 2^-12 / sqrt(3) = 1.4e-4 rms


# ---- the step
@pytest.fixture(scope="module")
def sd15():
    from oracle import unet_sd15 as O
    from pcm_amd.model import UNetWeights
    from pcm_amd.unet_spec import UNetConfig
    return UNetWeights(UNetConfig.sd15(), O.init_state_dict(O.UNetConfig.sd15(), 0), "cuda")


@pytest.mark.parametrize("b_std", [0.0, 0.02])
def test_fp16_step_vs_fp32_oracle(sd15, b_std):
    """BASELINE configs[0] (SD1.5 size, bs 2, 2 phases, huber, AdamW) against the fp32 oracle fixture: the loss inside 1e-3."""
    import step_golden_cases as S
    from golden_fixture import golden, sk_cos, sk_rel, sketch
    from pcm_amd.model import LoraState
    from pcm_amd.trainer import Distiller
    from pcm_amd.unet_spec import UNetConfig
    ref = golden(S.step_name(b_std), lambda: S.ref_sd15_step(b_std))
    inp = S.step_inputs()
    lora = LoraState(UNetConfig.sd15(), 64, 8.0, "cuda", seed=1, b_std=b_std)
    p_before = S.lora_flat(lora, "p")
    _, cfg = S.step_cfgs(2)
    D = Distiller(W := sd15, lora, cfg)
    assert D.loss_scale_dev is not None and W.layers[next(iter(W.layers))].w_fwd.dtype == torch.float16
    dev = {k: v.cuda() for k, v in inp.items()}
    out = D.step(dev["latents"], dev["prompt_embeds"], dev["uncond_prompt_embeds"], dev["noise"], dev["index"], dev["w"])
    torch.cuda.synchronize()
    scale = 65536.0
    assert float(D.loss_scale_dev.item()) == scale and int(D.loss_good_dev.item()) == 1 and int(D.step_dev.item()) == 1   # a finite step
    rep = {k: rel(out[k], ref[k]) for k in S.KEYS7}
    loss, rloss = float(out["loss"].item()), float(ref["loss"])
    rep["loss_rel"] = abs(loss - rloss) / abs(rloss)
    gn = math.sqrt(float(out["grad_sumsq"].item())) / scale
    rep["grad_norm_rel"] = abs(gn - float(ref["grad_norm"])) / float(ref["grad_norm"])
    rep["grad_rel"] = sk_rel(sketch(S.lora_flat(lora, "g") / scale), ref["sk_grad"])
    p_after = S.lora_flat(lora, "p")
    rep["param_rel"] = sk_rel(sketch(p_after), ref["sk_param_after"])
    rep["update_cos"] = sk_cos(sketch(p_after - p_before), ref["sk_update"])
    print("fp16 b_std", b_std, {k: "%.3e" % v for k, v in rep.items()}, "loss", loss, rloss)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(dict(rep, loss=loss, oracle_loss=rloss, b_std=b_std), open("gpurun_out/fp16_step_parity_bstd%g.json" % b_std, "w"))
    # MI355X (profiles/r04_u_fp16_step_vs_fp32_oracle.json): eps 1.11e-3, x_prev 1.6e-4, model_pred 2.4e-4, target 2.8e-4, loss 2.0e-4 / 3.3e-5,
    # gradient 5.2e-3 / 5.8e-3, update cosine 0.997 / 0.996 -- every bound below is under HALF of what the bf16 build measures
    assert rep["loss_rel"] < 1e-3, rep                       # north star: 1e-3 against the plain fp32 reference arithmetic
    assert rep["noise_pred"] < 2e-3 and rep["cond_teacher_output"] < 2e-3 and rep["target_noise_pred"] < 2e-3, rep
    assert rep["x_prev"] < 3e-4 and rep["model_pred"] < 5e-4 and rep["target"] < 6e-4, rep
    assert rep["grad_rel"] < 1.2e-2 and rep["grad_norm_rel"] < 4e-3, rep
    assert rep["param_rel"] < 6e-5 and rep["update_cos"] > 0.99, rep


def test_fp16_loss_curve_20_steps_vs_fp32_oracle(sd15):
    """20 consecutive optimizer steps at the real size (bs 2, 2 phases, lr 5e-6, B = 0 start, fresh inputs per step) against the fp32
    oracle's loss along ITS OWN AdamW trajectory (tests/golden/step_sd15_curve20_bs2.safetensors): mean |HIP - fp32| / fp32 <= 1e-3,
    no growth over the updates (sd15.py:1283-1301)."""
    import step_golden_cases as S
    from golden_fixture import golden, sk_cos, sk_rel, sketch
    from pcm_amd.model import LoraState
    from pcm_amd.trainer import Distiller
    from pcm_amd.unet_spec import UNetConfig
    ref = golden("sd15_curve20_bs2", S.ref_curve20)
    lora = LoraState(UNetConfig.sd15(), 64, 8.0, "cuda", seed=1, b_std=0.0)
    p0 = S.lora_flat(lora, "p")
    _, scfg = S.step_cfgs(2)
    D = Distiller(sd15, lora, scfg)
    rows = []
    for step in range(1, S.CURVE_STEPS + 1):
        inp = {k: v.cuda() for k, v in S.curve_inputs(step).items()}
        out = D.step(inp["latents"], inp["prompt_embeds"], inp["uncond_prompt_embeds"], inp["noise"], inp["index"], inp["w"])
        lh, lf, l16 = float(out["loss"].item()), ref["fp32"][step - 1], ref["bf16_autocast"][step - 1]
        rows.append(dict(step=step, hip_fp16=lh, oracle_fp32=lf, hip_vs_fp32=(lh - lf) / lf, ref_bf16_vs_fp32=(l16 - lf) / lf))
    mean = lambda key, rr=rows: sum(abs(r[key]) for r in rr) / len(rr)      # noqa: E731
    p1 = S.lora_flat(lora, "p")
    rep = dict(mean_abs_hip_fp16_vs_fp32=mean("hip_vs_fp32"), max_abs_hip_fp16_vs_fp32=max(abs(r["hip_vs_fp32"]) for r in rows),
               mean_signed=sum(r["hip_vs_fp32"] for r in rows) / len(rows), first5=mean("hip_vs_fp32", rows[:5]), last5=mean("hip_vs_fp32", rows[-5:]),
               mean_abs_ref_bf16_autocast_vs_fp32=mean("ref_bf16_vs_fp32"), loss_scale_after=float(D.loss_scale_dev.item()),
               optimizer_steps=int(D.step_dev.item()), param_rel_after_20=sk_rel(sketch(p1), ref["sk_param_after"]),
               update_cos_after_20=sk_cos(sketch(p1 - p0), ref["sk_update"]), rows=rows)
    print(json.dumps({k: v for k, v in rep.items() if k != "rows"}, indent=1))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rep, open("gpurun_out/fp16_loss_curve_20_real_size.json", "w"), indent=1)
    assert all(math.isfinite(r["hip_fp16"]) for r in rows) and rep["optimizer_steps"] == S.CURVE_STEPS      # no skipped update
    assert rep["mean_abs_hip_fp16_vs_fp32"] <= 1e-3, rep
    assert rep["last5"] <= rep["first5"] + 5e-4, rep
    assert rep["param_rel_after_20"] < 3e-4 and rep["update_cos_after_20"] > 0.99, rep


def test_fp16_graph_replay_equals_eager_and_overflow_is_skipped():
    """narrow UNet: (1) hipGraph replay of the loss-scaled step == eager launches (up to the order of the gradient atomics); (2) a non-finite gradient norm leaves the
    parameters and Adam moments untouched, halves the scale and does not count as an optimizer step (GradScaler semantics)."""
    from oracle import unet_sd15 as O
    from oracle import pcm_step as OS
    from pcm_amd.model import LoraState, UNetWeights
    from pcm_amd.trainer import Distiller, StepConfig
    from pcm_amd.unet_spec import UNetConfig
    kw = dict(block_out_channels=(64, 128, 128, 128), cross_attention_dim=64, heads=2, norm_num_groups=32)
    sd = O.init_state_dict(O.UNetConfig(**kw), 0)
    ocfg = OS.StepConfig(multiphase=2, loss_type="huber", lr=5e-6, adam_weight_decay=1e-3, w_min=4.0, w_max=5.0)
    inp = {k: v.cuda() for k, v in OS.draw_inputs(2, ocfg, seed=7, latent_hw=16, ctx_len=77, ctx_dim=64).items()}
    args = (inp["latents"], inp["prompt_embeds"], inp["uncond_prompt_embeds"], inp["noise"], inp["index"], inp["w"])
    cfg = StepConfig(multiphase=2, loss_type="huber", learning_rate=5e-6, adam_weight_decay=1e-3, w_min=4.0, w_max=5.0)
    res = []
    for graphed in (False, True):
        W = UNetWeights(UNetConfig(**kw), sd, "cuda")
        lora = LoraState(UNetConfig(**kw), 64, 8.0, "cuda", seed=1, b_std=0.02)
        D = Distiller(W, lora, cfg)
        if graphed:
            D.capture(2, H=16, W=16, ctx_len=77, ctx_dim=64)
        for _ in range(3):
            out = (D.step_graphed if graphed else D.step)(*args)
        torch.cuda.synchronize()
        res.append((float(out["loss"].item()), lora.params.clone(), float(D.loss_scale_dev.item()), int(D.step_dev.item())))
    # same kernels in the same order; the LoRA gradients are fp32 atomics, so the two runs may differ by summation order only
    # (three updates deep: the third step's loss and the parameters carry the atomics-order noise of the first two -- bounds as in tests/test_gpu_adv.py)
    assert abs(res[0][0] - res[1][0]) <= 1e-4 * abs(res[0][0]) and rel(res[1][1], res[0][1]) < 1e-4 and res[0][2:] == res[1][2:] == (65536.0, 3)
    # (2) overflow: poison one gradient after the backward, then run the optimizer leg
    out = D.forward_backward(*args)
    lora.grads[5] = float("inf")
    p, m, v = lora.params.clone(), lora.exp_avg.clone(), lora.exp_avg_sq.clone()
    D.optimizer_step()
    torch.cuda.synchronize()
    assert torch.equal(lora.params, p) and torch.equal(lora.exp_avg, m) and torch.equal(lora.exp_avg_sq, v)
    assert float(D.loss_scale_dev.item()) == 32768.0 and int(D.loss_good_dev.item()) == 0 and int(D.step_dev.item()) == 3
    out = D.step(*args)                      # the next step runs at the lower scale and is applied
    torch.cuda.synchronize()
    assert int(D.step_dev.item()) == 4 and not torch.equal(lora.params, p) and math.isfinite(float(out["loss"].item()))


@pytest.mark.parametrize("global_step", [0, 1])
def test_fp16_adv_step_c3_shape_full_size(global_step):
    """BASELINE configs[2]'s step (SD1.5 UNet, 36 heads, bs 2, lr 5e-6 / adv_lr 1e-5) through the half build: discriminator step (even) and
    generator step (odd) against the fp32-oracle fixture (tests/adv_cases.py); both backward seeds carry the device-side loss scale, the two
    optimizers share one GradScaler state (train_pcm_lora_sd15_adv.py:1383-1431 under --mixed_precision=fp16)."""
    import adv_cases as A
    from pcm_amd.discriminator import ADAPTER_DIMS
    kw = dict(block_out_channels=(320, 640, 1280, 1280), cross_attention_dim=768, heads=8, norm_num_groups=32)
    rep = A.case_adv_c3("cuda", kw, ADAPTER_DIMS, 2, 64, 77, 768, global_step, nh=4, index=[30, 12],
                        golden_name="sd15_adv_c3_bs2_step%d" % global_step)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rep, open("gpurun_out/fp16_adv_c3_parity_step%d.json" % global_step, "w"), indent=1)
    assert rep["heads"] == 36 and rep["fake_adv"] < 6e-4
    if global_step % 2 == 0:
        assert rep["d_loss_rel"] < 1e-3 and rep["lora_untouched"]
        assert rep["head_grad_cos"] > 0.995 and min(rep["head_grad_cos_per_tap"]) > 0.99 and rep["head_grad_norm_rel"] < 5e-3
        assert rep["head_update_cos"] > 0.95 and abs(rep["head_update_norm_ratio"] - 1) < 2e-2 and rep["head_param_rel_after"] < 6e-4
    else:
        assert rep["loss_cm_rel"] < 2e-3 and rep["g_loss_rel"] < 1e-3 and rep["heads_untouched"]
        assert rep["lora_grad_cos"] > 0.999 and rep["lora_grad_norm_rel"] < 5e-3
        assert rep["lora_update_cos"] > 0.97 and abs(rep["lora_update_norm_ratio"] - 1) < 2e-2 and rep["lora_param_rel_after"] < 3e-4


def test_fp16_adv_steps_graph_replay_equals_eager():
    """D, G, D, G through the two captured hipGraphs == eager, with the loss scale and the shared GradScaler state inside the graphs
    (the bf16 suite's case, tests/test_gpu_adv.py, run under the half build: its capi.set_lib(None) re-loads the current precision's library)"""
    from pcm_amd import capi
    from test_gpu_adv import test_adv_steps_graph_replay_equals_eager as case
    case()
    assert capi.lib().act_dtype == 1


def test_fp16_sdxl_topology_step_vs_oracle():
    """the SDXL wiring (added conditioning, linear projections, 1 / 2 / 3-deep transformers; narrow config, live fp32 oracle) through the
    half build: same case as tests/test_gpu_sdxl.py, bounds ~10x tighter than the bf16 build's"""
    from test_gpu_sdxl import sdxl_topology_step_case
    rep, _ = sdxl_topology_step_case()
    for k in ("noise_pred", "uncond_teacher_output", "x_prev", "target"):
        assert rep[k] < 3e-3, (k, rep)
    # MI355X: eps 1.2e-3, x_prev 2.8e-4, loss 2.1e-4, gradient cosine 0.9991 (bf16 build: 9.7e-3, 2.3e-3, 9.2e-3, 0.975)
    assert rep["loss_rel"] < 2e-3 and rep["grad_cos"] > 0.995 and 0.98 < rep["grad_norm_ratio"] < 1.02, rep


@pytest.mark.parametrize("case", ["step", "step_nocfg", "adv_d", "adv_g", "fwd_bwd"])
def test_fp16_sd3_mmdit_vs_oracle(case):
    """the SD3 / MMDiT trainers through the half build (every recipe of text_to_image_sd3/run.sh passes --mixed_precision=fp16): the narrow-config
    cases of tests/test_gpu_mmdit.py -- forward / LoRA backward, the flow-matching distillation step, the adversarial D and G steps -- with the
    loss-scaled backward (tests/mmdit_cases.py divides the scale out of the gradient buffers it compares)"""
    import mmdit_cases as M
    if case == "fwd_bwd":
        M.run_case("cuda")
    elif case.startswith("step"):
        M.run_step_case("cuda", case == "step_nocfg")
    else:
        M.run_adv_case("cuda", 0 if case == "adv_d" else 1)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("family", ["sdxl", "sd3"])
def test_fp16_full_size_one_sample_vs_fp32_oracle(family):
    """SDXL (2.57 B-parameter UNet, 128x128x4 latents) and SD3-medium (24 x 1536 MMDiT, 4096 + 154 tokens) at their REAL sizes through the
    half build: the full-size cases of tests/test_gpu_zy_sdxl_fullsize.py / test_gpu_zz_sd3_fullsize.py (one sample against the committed
    fp32-oracle fixture, finiteness, batch independence, one loss-scaled distillation step that is applied) -- no overflow at these depths,
    and the one-sample error a factor of ~6 under the bf16 build's (7.5e-3 / 1.16e-2)."""
    if family == "sdxl":
        from test_gpu_zy_sdxl_fullsize import test_sdxl_full_size_properties as case
        name = "sdxl_fullsize_oracle_parity.json"
    else:
        from test_gpu_zz_sd3_fullsize import test_sd3_medium_full_size_properties as case
        name = "sd3_fullsize_oracle_parity.json"
    case()
    rep = json.load(open(os.path.join("gpurun_out", name)))
    json.dump(rep, open(os.path.join("gpurun_out", "fp16_" + name), "w"), indent=1)
    print("fp16", family, rep)
    t, st = (next(v for k, v in rep.items() if k.startswith(w)) for w in ("teacher", "student"))
    # MI355X: SDXL 1.11e-3 / 1.10e-3, SD3-medium 1.66e-3 / 1.69e-3 (profiles/r04_ze_fp16_*_fullsize_oracle_parity.json)
    assert t < 3e-3 and st < 3e-3, rep
