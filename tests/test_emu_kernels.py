"""Host-emulation (CPU) runs of the kernel parity cases: index math of every kernel at small sizes."""
import pytest

import kernel_cases as K
from emu_lib import emu_lib
from pcm_amd import capi


@pytest.fixture(autouse=True)
def _use_emu():
    capi.set_lib(emu_lib())
    yield
    capi.set_lib(None)


@pytest.mark.parametrize("B,HW,C,G,act", [(2, 40, 320, 32, 1), (1, 9, 64, 8, 0), (2, 16, 2560, 32, 1), (1, 30, 960, 32, 1), (1, 300, 64, 8, 1),
                                            (3, 70, 640, 32, 0)])
def test_groupnorm(B, HW, C, G, act):
    K.case_groupnorm("cpu", B, HW, C, G, act)


@pytest.mark.parametrize("M,C", [(9, 320), (5, 640), (6, 1280), (3, 64)])
def test_layernorm(M, C):
    K.case_layernorm("cpu", M, C)


def test_elementwise():
    K.case_elementwise("cpu")


def test_edge_convs():
    K.case_edge_convs("cpu")
    K.case_edge_convs("cpu", B=2, H=5, W=8, C0=64)     # W%8==0: the 4-pixel conv_in and the 8-pixel conv_out kernels


def test_timestep_embedding():
    K.case_timestep_embedding("cpu")


def test_pcm_math_bit_exact_vs_reference_golden(golden):
    K.case_pcm_math("cpu", golden)


def test_pcm_fm_math_bit_exact_vs_reference_golden(golden_fm):
    K.case_pcm_fm_math("cpu", golden_fm)


def test_pcm_math_random_shapes_bit_exact_vs_oracle():
    K.case_pcm_math_random_shapes("cpu")


def test_mmdit_ops():
    K.case_mmdit_ops("cpu")


def test_optim():
    K.case_optim("cpu")


def test_pack():
    K.case_pack("cpu")


@pytest.mark.parametrize("M,N,K", [(300, 192, 72), (64, 64, 64)])
def test_wgrad_plain(M, N, K):
    import kernel_cases
    kernel_cases.case_wgrad_plain("cpu", M, N, K)


@pytest.mark.parametrize("stride,src_mode", [(1, 0), (2, 0), (1, 1)])
def test_wgrad_conv(stride, src_mode):
    K.case_wgrad_conv("cpu", 2, 6, 5, 64, stride, src_mode)


@pytest.mark.parametrize("B,H,Lq,Lk,d,spike", [(1, 2, 100, 77, 40, False), (2, 1, 70, 130, 80, True), (1, 1, 40, 64, 160, False), (1, 1, 300, 330, 40, True), (1, 2, 90, 90, 64, True)])
def test_attention(B, H, Lq, Lk, d, spike):
    K.case_attention("cpu", B, H, Lq, Lk, d, spike)


def test_lora_repack():
    K.case_lora_repack("cpu")


def test_adv_kernels(golden):
    K.case_adv_kernels("cpu", golden)


def test_discriminator_heads():
    K.case_discriminator_heads("cpu")


@pytest.mark.slow
def test_teacher_input_grad():
    K.case_teacher_input_grad("cpu")


@pytest.mark.parametrize("B,H,Lq,Lk,d", [(1, 2, 130, 130, 40), (1, 1, 70, 200, 80), (1, 2, 154, 154, 64)])
def test_attention_packed_transposed_operands(B, H, Lq, Lk, d):
    """pcm_attn_*_ws with the one-off packed V^T / K^T / Q^T / dO^T tile images (ragged tails zero-filled by the packer)."""
    from pcm_amd import capi
    dll = capi.lib().dll
    dll.pcm_debug_attn_pack_min_len(64)
    try:
        assert dll.pcm_attn_workspace_bytes(B, H, Lq, Lk, d, 1) > 0
        K.case_attention("cpu", B, H, Lq, Lk, d, spike=True)
    finally:
        dll.pcm_debug_attn_pack_min_len(1024)
