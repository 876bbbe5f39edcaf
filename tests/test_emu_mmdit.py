"""SD3 transformer (MMDiT) on the host emulator vs the CPU oracle.  The same cases run on the GPU in
tests/test_gpu_mmdit.py."""
import math

import pytest
import torch

from emu_lib import emu_lib
from pcm_amd import capi


@pytest.fixture(autouse=True)
def _use_emu():
    capi.set_lib(emu_lib())
    yield
    capi.set_lib(None)


def test_spec_counts():
    """the enumeration reproduces the published SD3-medium size and the reference's LoRA placement."""
    from oracle import mmdit_sd3 as O
    from pcm_amd import mmdit_spec as P
    oc, pc = O.MMDiTConfig.sd3_medium(), P.MMDiTConfig.sd3_medium()
    assert O.param_spec(oc) == P.param_spec(pc)
    assert sum(math.prod(s) for _, s in P.param_spec(pc)) == 2_028_328_000
    lt = P.lora_target_modules(pc)
    assert lt == O.lora_target_modules(oc) and len(lt) == 24 * 6 + 1 and lt[-1][0] == "proj_out"
    assert not any("add_" in p or "context" in p for p, _ in lt)
    assert torch.equal(O.sincos_pos_embed(oc)[:, :7], P.sincos_pos_embed(pc)[:, :7])


from mmdit_cases import run_accumulation_case, run_online_target_modes_case, run_prefetch_case, run_wgrad_defer_case, run_adv_case, run_case, run_sampler_case, run_step_case  # noqa: E402


@pytest.mark.slow
def test_mmdit_forward_backward_vs_oracle():
    run_case("cpu")


@pytest.mark.slow
@pytest.mark.parametrize("nocfg", [False, True])
def test_sd3_distillation_step_vs_oracle(nocfg):
    run_step_case("cpu", nocfg)


@pytest.mark.slow
def test_sd3_latent_sampler_vs_oracle():
    run_sampler_case("cpu")


@pytest.mark.slow
def test_mmdit_full_lora_list_forward_backward_vs_oracle():
    """the adversarial trainers' LoRA placement: gradients through the context stream, the adaLN modulation vectors, the
    time / text embedders, context_embedder and the patch-embedding conv."""
    run_case("cpu", adv_targets=True)


@pytest.mark.slow
@pytest.mark.parametrize("global_step", [0, 1])
def test_sd3_adversarial_step_vs_oracle(global_step):
    run_adv_case("cpu", global_step)


@pytest.mark.slow
def test_mmdit_unfused_qkv_schedule(monkeypatch):
    """PCM_MMDIT_QKV=0 debug path: the six q/k/v projections as separate layers give the same parity."""
    from pcm_amd import mmdit
    monkeypatch.setattr(mmdit, "FUSE_QKV", False)
    run_case("cpu", adv_targets=True)


@pytest.mark.slow
def test_gradient_accumulation_matches_the_full_batch_step():
    run_accumulation_case("cpu")


@pytest.mark.slow
def test_online_and_target_forward_as_one_pass_or_two_is_the_same_step():
    run_online_target_modes_case("cpu")


@pytest.mark.slow
def test_weight_gradient_jobs_collected_across_modules_give_the_same_gradients():
    run_wgrad_defer_case("cpu")


@pytest.mark.slow
def test_teacher_prefetch_gives_the_same_training_sequence():
    run_prefetch_case("cpu")


@pytest.mark.slow
def test_property_case_on_a_narrow_config():
    """the checks the GPU suite applies to SD3-medium at full size (tests/test_gpu_zz_sd3_fullsize.py), on the emulator."""
    from mmdit_cases import run_property_case
    from oracle import mmdit_sd3 as O
    from pcm_amd.mmdit import MMDiTWeights, sd3_lora_state
    from pcm_amd.mmdit_spec import MMDiTConfig
    kw = dict(sample_size=16, num_layers=3, attention_head_dim=64, num_attention_heads=2, joint_attention_dim=96, caption_projection_dim=128,
              pooled_projection_dim=64, pos_embed_max_size=12)
    pc = MMDiTConfig(**kw)
    W = MMDiTWeights(pc, O.init_state_dict(O.MMDiTConfig(**kw), 0), "cpu")
    run_property_case("cpu", pc, W, sd3_lora_state(pc, 32, 8.0, "cpu", seed=1), 8, 5)


def test_sd3_fp16_teacher_next_to_a_bf16_student_is_exactly_the_two_pure_builds():
    """SD3Distiller(teacher_weights = the frozen MMDiT weights packed in IEEE half): the reference's teacher pass sits under a dtype-less
    ``torch.autocast("cuda")`` (train_pcm_lora_sd3.py:1334).  Same statement as the UNet twin in tests/test_emu_unet.py: teacher output and
    x_prev bitwise an all-half process's, the online prediction bitwise the all-bfloat16 process's."""
    from oracle import mmdit_sd3 as O
    from pcm_amd import ops, precision
    from pcm_amd.mmdit import MMDiTWeights, sd3_lora_state
    from pcm_amd.mmdit_spec import MMDiTConfig
    from pcm_amd.trainer_sd3 import SD3Distiller, SD3StepConfig
    kw = dict(sample_size=16, num_layers=2, attention_head_dim=64, num_attention_heads=2, joint_attention_dim=96, caption_projection_dim=128,
              pooled_projection_dim=64, pos_embed_max_size=12)
    oc, pc = O.MMDiTConfig(**kw), MMDiTConfig(**kw)
    sd = O.init_state_dict(oc, 0)
    B, H, Wd, Lc = 2, 8, 8, 5
    g = torch.Generator().manual_seed(5)
    x0, noise = torch.randn(B, 16, H, Wd, generator=g), torch.randn(B, 16, H, Wd, generator=g)
    pe, upe = torch.randn(B, Lc, 96, generator=g), torch.randn(B, Lc, 96, generator=g)
    pp, upp = torch.randn(B, 64, generator=g), torch.randn(B, 64, generator=g)
    args = (x0, pe, pp, upe, upp, noise, torch.tensor([49, 13]))
    cfg = SD3StepConfig(multiphase=4)

    def run(teacher_weights=None):
        W = MMDiTWeights(pc, sd, "cpu")
        lora = sd3_lora_state(pc, 32, 8.0, "cpu", seed=1, b_std=0.1)
        out = SD3Distiller(W, lora, cfg, teacher_weights=teacher_weights).forward_backward(*args)
        return {k: out[k].clone() for k in ("cond_teacher_output", "uncond_teacher_output", "x_prev", "model_output", "loss")}

    precision.set_precision("bf16", lib=emu_lib("bf16"))
    precision.register_lib("fp16", emu_lib("f16"))
    try:
        pure_b = run()
        with precision.format_scope("fp16"):
            Wt = MMDiTWeights(pc, sd, "cpu", need_bwd=False)
        assert Wt.format == "fp16" and ops.BF16 == torch.bfloat16 and capi.lib().act_dtype == 0
        mixed = run(Wt)
        assert ops.BF16 == torch.bfloat16 and capi.lib().act_dtype == 0 and precision.precision() == "bf16"
        precision.set_precision("fp16", lib=emu_lib("f16"))
        pure_h = run()
    finally:
        precision.set_precision("bf16", lib=emu_lib("bf16"))
    for k in ("cond_teacher_output", "uncond_teacher_output", "x_prev"):
        assert torch.equal(mixed[k], pure_h[k]) and not torch.equal(mixed[k], pure_b[k]), k
    assert torch.equal(mixed["model_output"], pure_b["model_output"])
    assert abs(float(mixed["loss"]) - float(pure_b["loss"])) < 5e-2 * abs(float(pure_b["loss"]))
