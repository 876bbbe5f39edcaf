"""SD3 variant on the MI355X: MMDiT forward / backward and one PCM distillation step vs the CPU oracle (same cases as
tests/test_emu_mmdit.py, through the real HIP library)."""
import pytest

pytestmark = pytest.mark.gpu


def test_mmdit_forward_backward_vs_oracle():
    from mmdit_cases import run_case
    run_case("cuda")


@pytest.mark.parametrize("nocfg", [False, True])
def test_sd3_distillation_step_vs_oracle(nocfg):
    from mmdit_cases import run_step_case
    run_step_case("cuda", nocfg)


def test_sd3_latent_sampler_vs_oracle():
    from mmdit_cases import run_sampler_case
    run_sampler_case("cuda")


def test_mmdit_full_lora_list_forward_backward_vs_oracle():
    from mmdit_cases import run_case
    run_case("cuda", adv_targets=True)


@pytest.mark.parametrize("global_step", [0, 1])
def test_sd3_adversarial_step_vs_oracle(global_step):
    from mmdit_cases import run_adv_case
    run_adv_case("cuda", global_step)


def test_teacher_prefetch_and_pipelined_graph_give_the_same_training_sequence():
    """eager steps, eager steps with the next batch's teacher pass on a side stream, and the pipelined hipGraph: bitwise the same three steps"""
    from mmdit_cases import run_prefetch_case
    run_prefetch_case("cuda", graphs=True)


def test_online_and_target_forward_as_one_pass_or_two_is_the_same_step():
    from mmdit_cases import run_online_target_modes_case
    run_online_target_modes_case("cuda")


def test_weight_gradient_jobs_collected_across_modules_give_the_same_gradients():
    from mmdit_cases import run_wgrad_defer_case
    run_wgrad_defer_case("cuda")
