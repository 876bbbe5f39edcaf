"""GPU parity of every C-ABI kernel vs torch fp32 on the same bf16-rounded inputs, at UNet sizes."""
import pytest
import torch

import kernel_cases as K

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _hip():
    from pcm_amd import capi
    assert torch.cuda.is_available()
    # TOOLS build: several cases select kernels / read counters through pcm_debug_* hooks (tests/test_gpu_product_lib.py runs the hook-free
    # cases on the product library)
    capi.set_lib(capi.tools_lib())  # raises loudly if libpcm_hip_tools.so is missing
    yield
    torch.cuda.synchronize()
    capi.set_lib(None)


@pytest.mark.parametrize("B,HW,C,G,act", [(2, 4096, 320, 32, 1), (2, 1024, 640, 32, 0), (2, 256, 2560, 32, 1),
                                          (1, 4096, 960, 32, 1), (3, 64, 1280, 32, 1), (2, 1024, 1920, 32, 1)])
def test_groupnorm(B, HW, C, G, act):
    K.case_groupnorm("cuda", B, HW, C, G, act, eps=1e-6 if act == 0 else 1e-5)


@pytest.mark.parametrize("M,C", [(4099, 320), (1024, 640), (300, 1280)])
def test_layernorm(M, C):
    K.case_layernorm("cuda", M, C)


def test_elementwise():
    K.case_elementwise("cuda")


def test_edge_convs():
    K.case_edge_convs("cuda", B=2, H=64, W=64, C0=320)


def test_timestep_embedding():
    K.case_timestep_embedding("cuda")


def test_pcm_math_bit_exact_vs_reference_golden(golden):
    K.case_pcm_math("cuda", golden)


def test_pcm_math_random_shapes_bit_exact_vs_oracle():
    K.case_pcm_math_random_shapes("cuda")


def test_mmdit_ops():
    K.case_mmdit_ops("cuda")


def test_pcm_fm_math_bit_exact_vs_reference_golden(golden_fm):
    K.case_pcm_fm_math("cuda", golden_fm)


def test_optim():
    K.case_optim("cuda")


def test_pack():
    K.case_pack("cuda")


@pytest.mark.parametrize("M,N,K_", [(4096, 320, 320), (1000, 1280, 768), (16, 640, 1280)])
def test_wgrad_plain(M, N, K_):
    K.case_wgrad_plain("cuda", M, N, K_)


def test_reproducible_reduction_forms_bitwise_run_to_run():
    """ops.set_deterministic at step-like sizes: the slab / partial forms agree with the atomic forms to summation rounding and two runs are
    bitwise equal (LoRA weight gradients, GroupNorm statistics, pixel sums, gradient norm, loss)"""
    K.case_reproducible_reductions("cuda", big=True)


def test_wgrad_multi_job_launch():
    K.case_wgrad_multi("cuda")


@pytest.mark.parametrize("stride,src_mode,C", [(1, 0, 320), (2, 0, 64), (1, 1, 128)])
def test_wgrad_conv(stride, src_mode, C):
    K.case_wgrad_conv("cuda", 2, 16, 16, C, stride, src_mode)


# dense 3x3 weight gradient (wgrad_dense.hip) on the discriminator-head geometries: the 8x8 tap at 1280 channels (one owner per tile:
# read-add-write epilogue), 16x16 at 640 and 32x32 at 320 (M split + atomics, ragged 128-channel tile), a 64x64 tap, rectangular images
@pytest.mark.parametrize("B,H,W,Cin,Cout,alpha", [(4, 8, 8, 1280, 1280, 1.0), (4, 16, 16, 640, 640, 1.0), (2, 32, 32, 320, 320, 0.5),
                                                  (1, 64, 64, 320, 320, 1.0), (3, 16, 40, 192, 128, 1.0)])
def test_wgrad_dense_conv3x3(B, H, W, Cin, Cout, alpha):
    K.case_wgrad_dense("cuda", B, H, W, Cin, Cout, alpha)


@pytest.mark.parametrize("B,H,Lq,Lk,d,spike", [(1, 8, 1024, 1024, 40, False), (2, 8, 256, 77, 80, False),
                                               (2, 8, 256, 256, 160, True), (1, 8, 64, 64, 160, False),
                                               (1, 2, 4096, 4096, 40, True), (1, 8, 4096, 77, 40, False),
                                               # SDXL level 2 / level 1, SD3 joint tokens (ragged), d = 32: packed path at H*d >= 1280
                                               (1, 20, 1024, 1024, 64, True), (1, 10, 2176, 2176, 64, False),
                                               (1, 24, 1178, 1178, 64, True), (1, 40, 1024, 1024, 32, False)])
def test_attention(B, H, Lq, Lk, d, spike):
    K.case_attention("cuda", B, H, Lq, Lk, d, spike)


@pytest.mark.parametrize("B,H,Lq,Lk,d,at", [(2, 8, 1024, 1024, 80, (300, 700)), (1, 8, 4096, 4096, 40, (1000, 3000)), (1, 10, 2176, 2176, 64, (64, 2100)),
                                            (1, 2, 200, 77, 40, None), (1, 4, 333, 1000, 32, (500,))])
def test_attention_pipelined_forward(B, H, Lq, Lk, d, at):
    """the software-pipelined forward (attention_fwd.hip) FORCED on every head dim it is built for, incl. shapes the by-shape default would
    give to the first kernel: long streams with reference moves inside the steady-state loop, a two-tile stream, ragged tails"""
    from pcm_amd import capi
    dll = capi.lib().dll
    dll.pcm_debug_attn_fwd_variant(1)
    try:
        K.case_attention("cuda", B, H, Lq, Lk, d, spike=True, spike_at=at)
    finally:
        dll.pcm_debug_attn_fwd_variant(-1)


@pytest.mark.parametrize("B,H,Lq,Lk,d,spike", [(1, 8, 1024, 1024, 40, False), (2, 8, 256, 77, 80, False), (2, 8, 256, 256, 160, True),
                                               (1, 2, 4096, 4096, 40, True), (1, 8, 4096, 77, 40, False), (2, 8, 1024, 1024, 80, True),
                                               (1, 20, 1024, 1024, 64, True), (1, 10, 2176, 2176, 64, False)])
@pytest.mark.parametrize("track,dma", [(0, 1), (1, 1), (0, 0)])
def test_attention_prescaled_query(B, H, Lq, Lk, d, spike, track, dma):
    """csrc/attention_ps.hip at the step's shapes (SD1.5 levels, text cross-attention, SDXL head dim 64): the softmax scale lives in q, the
    reference subtraction rides the MFMA at d = 40, no running maximum after the first tile; track = 1: the tracking fallback from the start"""
    from pcm_amd import capi
    dll = capi.lib().dll
    dll.pcm_debug_attn_ps_track(track)
    dll.pcm_debug_attn_ps_dma(dma)            # K / V (Q / dO) tiles by double-buffered LDS-DMA, or through registers
    try:
        K.case_attention("cuda", B, H, Lq, Lk, d, spike, prescaled=True)
    finally:
        dll.pcm_debug_attn_ps_track(0)
        dll.pcm_debug_attn_ps_dma(1)


@pytest.mark.parametrize("B,H,Lq,Lk,d", [(1, 8, 1024, 1024, 40), (1, 10, 1024, 1024, 64)])
def test_attention_prescaled_query_overflow_fallback(B, H, Lq, Lk, d):
    """a late key ~250 (log2 domain) above the first tile's maximum -> non-finite row sum -> the workgroup repeats with tracking"""
    K.case_attention("cuda", B, H, Lq, Lk, d, spike=True, prescaled=True, spike_overflow=True)
    K.case_attention("cuda", B, H, Lq, Lk, d, spike=True, spike_at=(300, 700), prescaled=True)


def test_lora_repack():
    K.case_lora_repack("cuda")


def test_adv_kernels(golden):
    K.case_adv_kernels("cuda", golden)


def test_discriminator_heads():
    K.case_discriminator_heads("cuda", dims=(320, 1280), hw=(32, 8), B=2, nh=2)


def test_teacher_input_grad():
    K.case_teacher_input_grad("cuda")
