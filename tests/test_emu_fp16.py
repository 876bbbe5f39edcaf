"""Host-emulation (CPU) checks of the IEEE-half build of the kernels (-DPCM_ACT_F16, csrc/pcm_common.h): the same parity cases through
tests/emu/libpcm_emu_f16.so, the device-side loss scaler, and the precision switch itself (pcm_amd/precision.py)."""
import math

import pytest
import torch

import kernel_cases as K
from emu_lib import emu_lib
from pcm_amd import capi, ops, precision


@pytest.fixture(autouse=True)
def _half_emu():
    precision.set_precision("fp16", lib=emu_lib("f16"))
    assert ops.BF16 == torch.float16 and capi.lib().act_dtype == 1
    yield
    precision.set_precision("bf16", lib=emu_lib("bf16"))
    capi.set_lib(None)
    assert ops.BF16 == torch.bfloat16


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_kernel_cases_half_build():
    K.case_groupnorm("cpu", 2, 40, 320, 32, 1)
    K.case_layernorm("cpu", 9, 320)
    K.case_elementwise("cpu")
    K.case_timestep_embedding("cpu")
    K.case_pack("cpu")
    K.case_wgrad_plain("cpu", 70, 64, 128)
    K.case_wgrad_conv("cpu", 1, 6, 5, 64, 1, 0)
    assert K.case_gemm_n64("cpu", 200, 192) <= 0
    assert K.case_gemm_smallm("cpu", 13, 320, (64, 64), 1, True, False) <= 0 and K.case_gemm_smallm("cpu", 2, 144, (96,), 0, True, True) <= 0
    K.case_conv_r64("cpu", 1, 8, 8, 64, expect_kernel=False)
    K.case_lora_repack("cpu")


@pytest.mark.parametrize("Lq,Lk,d", [(70, 70, 40), (33, 77, 80), (64, 64, 160)])
def test_attention_half_build(Lq, Lk, d):
    K.case_attention("cpu", 1, 2, Lq, Lk, d)


@pytest.mark.parametrize("Lq,Lk,d", [(40, 128, 32), (40, 128, 64), (40, 128, 160), (40, 77, 160), (40, 200, 40), (40, 150, 80)])
@pytest.mark.parametrize("gain", [4, 8, 16, 40])
def test_attention_half_build_late_key_beyond_the_half_range(Lq, Lk, d, gain):
    """round-5 advisor finding: a late key ~25 .. 100 (log2 domain) above the first tile's row maximum makes p = exp2(s' - m_1) exceed 65504;
    packed to IEEE half it is inf in the PV MFMA.  Head dims whose row sum is the fp32 VALU sum (32 / 64 / 160: no ones column in V) kept a
    finite sum, so the one-time check never fired and the rows came out NaN.  The check now reads the output accumulators too."""
    K.case_attention("cpu", 1, 1, Lq, Lk, d, spike=True, prescaled=True, spike_overflow=True, spike_gain=gain)


def test_gemm_tiles_half_build():
    for excess, err in (K.case_gemm_big("cpu", "plain_lora"), K.case_gemm_big("cpu", "conv"), K.case_gemm_4w("cpu", "plain_lora")):
        assert excess <= 0, err
    assert K.case_gemm_geglu("cpu") <= 0


def test_half_conversions_and_mfma_are_ieee_half():
    """a plain GEMM on values that bfloat16 cannot hold (11-bit significands) is exact to the output rounding: the half MFMA + converts are in"""
    g = torch.Generator().manual_seed(0)
    M, N, Kd = 64, 64, 64
    x = (torch.randint(1024, 2048, (M, Kd), generator=g).float() / 1024).half()       # 1.0 .. 2.0 in steps of 2^-10: exact in half only
    w = (torch.randint(-8, 9, (N, Kd), generator=g).float() / 8).half()
    assert not torch.equal(x.float().bfloat16().float(), x.float())
    out = torch.empty(M, N, dtype=torch.float32)
    ops.gemm([ops.Seg(x, w)], M, N, out)
    assert torch.equal(out, x.float() @ w.float().t())            # every partial sum is exactly representable in fp32
    out16 = torch.empty(M, N, dtype=torch.float16)
    ops.gemm([ops.Seg(x, w)], M, N, out16)
    assert torch.equal(out16, (x.float() @ w.float().t()).half())  # round-to-nearest-even into half
    big = torch.full((M, Kd), 300.0).half()
    ops.gemm([ops.Seg(big, big[:N])], M, N, out16)                # 64 * 9e4 = 5.76e6 > 65504: half output saturates to inf (no wrap, no NaN)
    assert bool(torch.isinf(out16).all())


def test_device_side_grad_scaler_semantics():
    """pcm_scale_f32_dev / pcm_adamw_clip_step_scaled / pcm_loss_scale_update == torch.cuda.amp.GradScaler(65536, 2, 0.5, interval) around
    clip_grad_norm_ + AdamW (train_pcm_lora_sd15.py:1296-1299 under --mixed_precision=fp16)."""
    n, S0 = 1000, 65536.0
    g = torch.Generator().manual_seed(1)
    p = torch.randn(n, generator=g)
    grad = torch.randn(n, generator=g) * 1e-3
    m, v = torch.zeros(n), torch.zeros(n)
    scale, good, step = torch.tensor([S0]), torch.zeros(1, dtype=torch.int32), torch.zeros(1, dtype=torch.int64)
    lr = torch.tensor([1e-3])
    sq = torch.zeros(1, dtype=torch.float64)
    # reference: torch AdamW on the unscaled gradient with the same clip
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)

    def hip_step(gs):
        step.add_(1)
        ops.sumsq(gs, sq)
        ops.adamw_clip_step_scaled(p, gs, m, v, sq, 0.01, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 1.0, step, lr, scale)
        ops.loss_scale_update(scale, good, step, sq, growth=2.0, backoff=0.5, interval=2)

    d = grad.clone()
    ops.scale_by_dev(d, scale)
    assert torch.equal(d, grad * S0)
    hip_step(d)                                                   # finite: applied
    pr.grad = grad.clone()
    torch.nn.utils.clip_grad_norm_([pr], 0.01)
    opt.step()
    assert torch.allclose(p, pr.detach(), rtol=2e-6, atol=1e-7) and int(step) == 1 and int(good) == 1 and float(scale) == S0
    bad = d.clone(); bad[3] = float("nan")
    p0, m0, v0 = p.clone(), m.clone(), v.clone()
    hip_step(bad)                                                 # non-finite: skipped, scale halves, the step is not counted
    assert torch.equal(p, p0) and torch.equal(m, m0) and torch.equal(v, v0) and int(step) == 1 and int(good) == 0 and float(scale) == S0 / 2
    for k in range(2):                                            # two finite steps at the new scale: applied, then the scale grows back
        d = grad.clone(); ops.scale_by_dev(d, scale)
        hip_step(d)
        pr.grad = grad.clone()
        torch.nn.utils.clip_grad_norm_([pr], 0.01)
        opt.step()
    assert torch.allclose(p, pr.detach(), rtol=5e-6, atol=1e-7) and int(step) == 3 and int(good) == 0 and float(scale) == S0


def test_skipped_step_moves_neither_ema_nor_step_count_nor_lr_position():
    """A non-finite loss-scaled gradient: GradScaler semantics skip the optimizer step -- and with it (accelerate) the lr scheduler step; the
    EMA of the unchanged parameters must not advance either.  Through the trainer's own _optimizer_apply on a tiny LoRA state."""
    from pcm_amd.model import LoraState, UNetWeights
    from pcm_amd.trainer import Distiller, StepConfig
    from pcm_amd.unet_spec import UNetConfig as PC
    from oracle import unet_sd15 as O
    import train_pcm_lora_sd15 as cli
    kw = dict(block_out_channels=(64, 128, 128), cross_attention_dim=64, heads=2, norm_num_groups=32)
    W = UNetWeights(PC(**kw), O.init_state_dict(O.UNetConfig(**kw), 0), "cpu")
    lora = LoraState(PC(**kw), 64, 8.0, "cpu", seed=1, b_std=0.02)
    D = Distiller(W, lora, StepConfig(multiphase=2, ema_rate=0.9))
    assert D.loss_scale_dev is not None and D.ema is not None

    class A:
        mixed_precision, lr_scheduler, learning_rate, lr_warmup_steps, max_train_steps = "fp16", "linear", 1e-3, 2, 10
    g = torch.Generator().manual_seed(3)
    lora.grads.copy_(torch.randn(lora.grads.shape, generator=g) * 1e-3 * float(D.loss_scale_dev))
    D.optimizer_step()                                       # finite: applied
    assert D.applied_steps() == 1 and cli.sched_pos(D, A, 1) == 1
    p1, e1, s1 = lora.params.clone(), D.ema.clone(), float(D.loss_scale_dev)
    lora.grads[5] = float("inf")
    D.optimizer_step()                                       # overflow: everything stays, the scale backs off
    assert torch.equal(lora.params, p1) and torch.equal(D.ema, e1), "a skipped step moved the parameters or their EMA"
    assert D.applied_steps() == 1 and cli.sched_pos(D, A, 2) == 1 and float(D.loss_scale_dev) == s1 / 2
    assert cli.lr_at(A, cli.sched_pos(D, A, 2)) == cli.lr_at(A, 1)          # the schedule did not advance
    A.lr_scheduler = "constant"
    assert cli.sched_pos(D, A, 2) == 2                                     # constant schedule: no device read needed
    lora.grads.copy_(torch.randn(lora.grads.shape, generator=g) * 1e-3 * float(D.loss_scale_dev))
    D.optimizer_step()
    assert D.applied_steps() == 2 and not torch.equal(D.ema, e1)


def test_tiny_unet_forward_half_build_closer_to_fp32_than_bf16():
    """teacher forward of a 3-level SD1.5-topology UNet through the half emulator build against the fp32 oracle: the error sits at the
    half rounding level (the bf16 build: ~8x that, tests/test_emu_unet.py)."""
    from oracle import unet_sd15 as O
    from pcm_amd.model import UNet, UNetWeights
    from pcm_amd.unet_spec import UNetConfig as PC
    kw = dict(block_out_channels=(64, 128, 128), cross_attention_dim=64, heads=2, norm_num_groups=32)
    oc, pc = O.UNetConfig(**kw), PC(**kw)
    sd = O.init_state_dict(oc, 0)
    g = torch.Generator().manual_seed(11)
    x, t, ctx = torch.randn(1, 4, 8, 8, generator=g), torch.tensor([419]), torch.randn(1, 7, 64, generator=g)
    ref = O.unet_forward(oc, sd, x, t, ctx)
    W = UNetWeights(pc, sd, "cpu")
    assert W.layers[next(iter(W.layers))].w_fwd.dtype == torch.float16
    out = UNet(W, None).forward(x, t, ctx)
    r = rel(out.float(), ref)
    print("tiny UNet, half build vs fp32 oracle: rel-L2 %.3e" % r)
    assert math.isfinite(r) and r < 2.5e-3, r


def test_precision_switch_loads_the_matching_library():
    """the product libraries (dlopen works without a GPU): each reports its format, set_precision refuses a mismatch and rebinds the dtype"""
    from pcm_amd import build as B
    bf, h = capi.Lib(B.build(variant="bf16")), capi.Lib(B.build(variant="f16"))
    assert (bf.act_dtype, h.act_dtype) == (0, 1)
    assert not [n for n in capi._PROTOS if n not in h.fn], "libpcm_hip_f16.so must export the whole C ABI"
    with pytest.raises(RuntimeError, match="built for"):
        precision.set_precision("fp16", lib=bf)
    with pytest.raises(ValueError):
        precision.set_precision("fp32")
    precision.set_precision("bf16")
    assert capi.lib().path == capi.DEFAULT_LIB and ops.BF16 == torch.bfloat16 and precision.act_dtype() == torch.bfloat16
    precision.set_precision("fp16")
    assert capi.lib().path == capi.F16_LIB and ops.BF16 == torch.float16
    capi.set_lib(None)
    assert capi.lib().path == capi.F16_LIB            # set_lib(None) re-loads the CURRENT precision's library
    precision.set_precision("bf16")
    assert capi.lib().path == capi.DEFAULT_LIB


def test_cli_mixed_precision_fp16_end_to_end(tmp_path, monkeypatch):
    """train_pcm_lora_sd15.py --mixed_precision=fp16 as a program (narrow UNet, host emulator of the half build): the flag selects the half
    library, steps are loss-scaled and applied, the checkpoint carries the GradScaler state and a resumed run restores it."""
    import importlib.util
    import json
    import os
    from safetensors.torch import save_file
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "phased-consistency-model_amd")
    spec = importlib.util.spec_from_file_location("pcm_cli_fp16", os.path.join(pkg, "train_pcm_lora_sd15.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    g = torch.Generator().manual_seed(0)
    shards = tmp_path / "s"
    shards.mkdir()
    save_file({"latents": torch.randn(6, 4, 8, 8, generator=g), "prompt_embeds": torch.randn(6, 7, 64, generator=g),
               "uncond_prompt_embeds": torch.randn(7, 64, generator=g)}, str(shards / "a.safetensors"))
    monkeypatch.setenv("PCM_CLI_DEVICE", "cpu")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    real, asked = precision.set_precision, []
    precision.set_precision("bf16", lib=emu_lib("bf16"))          # the program starts in the default format ...

    def via_emulator(name, lib=None):                              # ... and its set_precision("fp16") gets the emulator's half build
        asked.append(name)
        real(name, lib=emu_lib("f16" if name == "fp16" else "bf16"))
    monkeypatch.setattr(precision, "set_precision", via_emulator)
    out = tmp_path / "o"
    common = ["--pretrained_teacher_model", "random", "--tiny_model", "--latents_dir", str(shards), "--train_batch_size", "1", "--learning_rate", "1e-3",
              "--multiphase", "2", "--seed", "1", "--output_dir", str(out), "--mixed_precision", "fp16", "--loss_type", "huber", "--checkpointing_steps", "2"]
    cli.main(cli.parse_args(common + ["--max_train_steps", "2"]))
    assert asked == ["fp16"] and ops.BF16 == torch.float16 and capi.lib().act_dtype == 1
    st = json.load(open(out / "checkpoint-2" / "trainer_state.json"))
    assert st["loss_scale"] == 65536.0 and st["loss_scale_good_steps"] == 2 and st["optimizer_step"] == 2
    log = [json.loads(l) for l in open(out / "logs" / "text2image-fine-tune.jsonl")]
    assert [r["step"] for r in log] == [1, 2] and all(math.isfinite(r["loss"]) and 0 < r["grad_norm"] < 1e3 for r in log)   # the UNSCALED norm is logged
    cli.main(cli.parse_args(common + ["--max_train_steps", "3", "--resume_from_checkpoint", "latest"]))
    st = json.load(open(out / "checkpoint-2" / "trainer_state.json"))
    log = [json.loads(l) for l in open(out / "logs" / "text2image-fine-tune.jsonl")]
    assert log[-1]["step"] == 3 and math.isfinite(log[-1]["loss"])


@pytest.mark.parametrize("case", ["step", "adv_d", "adv_g"])
def test_sd3_mmdit_half_build(case):
    """SD3 / MMDiT trainers through the half emulator build: the flow-matching step and the adversarial D / G steps with the loss-scaled
    backward (narrow config, live oracle; tests/mmdit_cases.py)"""
    import mmdit_cases as M
    if case == "step":
        M.run_step_case("cpu")
    else:
        M.run_adv_case("cpu", 0 if case == "adv_d" else 1)


def test_operands_of_the_other_format_are_refused():
    """a bfloat16 tensor handed to the half build would be read as garbage: the op layer refuses it (and the other way round after switching back)"""
    x16, w16 = torch.randn(64, 64).half(), torch.randn(64, 64).half()
    xb, wb = x16.float().bfloat16(), w16.float().bfloat16()
    out = torch.empty(64, 64, dtype=torch.float16)
    with pytest.raises(TypeError, match="operands must be"):
        ops.gemm([ops.Seg(xb, wb)], 64, 64, out)
    ops.gemm([ops.Seg(x16, w16)], 64, 64, out)
    precision.set_precision("bf16", lib=emu_lib("bf16"))
    with pytest.raises(TypeError, match="operands must be"):
        ops.gemm([ops.Seg(x16, w16)], 64, 64, torch.empty(64, 64, dtype=torch.bfloat16))
