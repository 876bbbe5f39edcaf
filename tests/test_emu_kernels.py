"""Host-emulation (CPU) runs of the kernel parity cases: index math of every kernel at small sizes."""
import ctypes

import pytest
import torch

import kernel_cases as K
from emu_lib import emu_lib
from pcm_amd import capi, ops


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


# geometries the LDS-DMA + transpose-read kernels take (wgrad_tr.hip): 8x8 (8 image rows per 64-pixel stage), 16x16 (4 rows), 64-wide
# (one row per stage, two channel tiles), 128-wide (two stages per image row); the plain cases above cover the plain kernel incl. ragged M
@pytest.mark.parametrize("B,H,W,C", [(64, 8, 8, 64), (16, 16, 16, 128), (32, 2, 64, 64), (32, 1, 128, 64), (64, 4, 16, 64)])
def test_wgrad_conv_transpose_read_kernel(B, H, W, C):
    import ctypes
    from pcm_amd import capi
    cnt = capi.lib().dll.pcm_debug_wgrad_tr_count
    cnt.restype = ctypes.c_long
    n0 = cnt(1)
    K.case_wgrad_conv("cpu", B, H, W, C, 1, 0)
    assert cnt(1) == n0 + 1      # the khwc-layout call took the transpose-read kernel (the peft-layout call stays on wgrad.hip)


# dense 3x3 weight gradient (wgrad_dense.hip): one 8x8 patch; borders on all four sides + several patches per image + two images; a
# ragged input-channel tile (72 of 128) and two output tiles; enough stages for the M split (atomic epilogue) under the emulator's grid cap
@pytest.mark.parametrize("B,H,W,Cin,Cout,alpha", [(1, 8, 8, 64, 64, 1.0), (2, 16, 24, 72, 128, 0.5), (8, 16, 16, 128, 64, 1.0), (1, 8, 16, 192, 64, 1.0)])
def test_wgrad_dense_conv3x3(B, H, W, Cin, Cout, alpha):
    K.case_wgrad_dense("cpu", B, H, W, Cin, Cout, alpha)


def test_wgrad_dense_conv3x3_rejects_unsupported_geometry():
    import torch
    from pcm_amd import capi, ops
    x = torch.zeros(1, 6, 8, 64, dtype=torch.bfloat16); dy = torch.zeros(48, 64, dtype=torch.bfloat16); dW = torch.zeros(64, 9 * 64)
    with pytest.raises(capi.PcmError, match="pcm_conv3x3_wgrad_bf16"):
        ops.conv3x3_wgrad(x, dy, dW, 1, 6, 8)
    assert not ops.conv3x3_wgrad_ok(6, 8, 64, 64) and not ops.conv3x3_wgrad_ok(8, 8, 64, 96)


def test_reproducible_reduction_forms():
    """ops.set_deterministic: slabs / partials + ordered finalize instead of fp32 / fp64 atomics (the emulator runs workgroups serially, so
    the bitwise run-to-run half of the case is trivially true here; the GPU test is the one that can fail on it)"""
    K.case_reproducible_reductions("cpu")


def test_reproducible_wgrad_refuses_a_short_workspace():
    a = capi.WgradArgs()
    big, small = torch.zeros(256, 64, dtype=ops.BF16), torch.zeros(256, 64, dtype=ops.BF16)
    out, ws = torch.zeros(64, 64), torch.zeros(16)
    a.big, a.small_, a.out = capi.ptr(big), capi.ptr(small), capi.ptr(out)
    a.ldb, a.G, a.mode, a.lds_, a.M, a.g_stride, a.r_stride, a.alpha = 64, 64, capi.SEG_PLAIN, 64, 256, 64, 1, 1.0
    a.stride = 1
    need = capi.lib().dll.pcm_lora_wgrad_workspace_bytes(ctypes.byref(a))
    assert need >= 64 * 64 * 4
    a.workspace, a.workspace_bytes = capi.ptr(ws), ws.numel() * 4
    with pytest.raises(capi.PcmError, match="workspace too small"):
        capi.lib().call("pcm_lora_wgrad_bf16", ctypes.byref(a), capi.Lib.stream())


def test_wgrad_multi_job_launch():
    import ctypes
    from pcm_amd import capi
    cnt = capi.lib().dll.pcm_debug_wgrad_tr_multi_count
    cnt.restype = ctypes.c_long
    n0 = cnt()
    K.case_wgrad_multi("cpu")
    assert cnt() == n0 + 1       # 9 plain jobs = ONE multi-job launch of 8 + a single-job launch; the 10th (3x3 view) runs on its own kernel


def test_wgrad_transpose_read_matches_register_transposing_kernel():
    """A/B of the two implementations of the same contract on one ragged plain shape (both must agree with the reference)."""
    from pcm_amd import capi
    dll = capi.lib().dll
    for mode in (0, 1):
        dll.pcm_debug_wgrad_tr(mode)
        try:
            K.case_wgrad_plain("cpu", 333, 200, 136)
        finally:
            dll.pcm_debug_wgrad_tr(1)


@pytest.mark.parametrize("B,H,Lq,Lk,d,spike", [(1, 2, 100, 77, 40, False), (2, 1, 70, 130, 80, True), (1, 1, 40, 64, 160, False), (1, 1, 300, 330, 40, True), (1, 2, 90, 90, 64, True)])
def test_attention(B, H, Lq, Lk, d, spike):
    K.case_attention("cpu", B, H, Lq, Lk, d, spike)


@pytest.mark.parametrize("B,H,Lq,Lk,d,spike", [(1, 2, 100, 77, 40, False), (2, 1, 70, 130, 80, True), (1, 1, 40, 64, 160, False), (1, 1, 300, 330, 40, True),
                                               (1, 2, 90, 90, 64, True), (1, 1, 33, 64, 40, False), (1, 1, 70, 13, 40, False), (1, 1, 40, 400, 32, True)])
@pytest.mark.parametrize("track,dma", [(0, 1), (1, 1), (0, 0)])
def test_attention_prescaled_query(B, H, Lq, Lk, d, spike, track, dma, monkeypatch):
    """csrc/attention_ps.hip: forward + dQ' + dK/dV with the softmax scale folded into q (spare-slot reference subtraction at d = 40, no
    running maximum after the first key tile); track = 1 forces the per-tile maximum tracking the overflow fallback runs; dma: K / V (Q / dO)
    tiles by double-buffered LDS-DMA (run with the emulator's DMA landing lazily, i.e. only at the counted wait: a read placed before its
    wait would see stale LDS) or through registers"""
    monkeypatch.setenv("PCM_EMU_LAZY_DMA", "1")
    dll = capi.lib().dll
    dll.pcm_debug_attn_ps_track(track)
    dll.pcm_debug_attn_ps_dma(dma)
    try:
        K.case_attention("cpu", B, H, Lq, Lk, d, spike, prescaled=True)
    finally:
        dll.pcm_debug_attn_ps_track(0)
        dll.pcm_debug_attn_ps_dma(1)


@pytest.mark.parametrize("B,H,Lq,Lk,d", [(1, 2, 70, 330, 40), (1, 1, 40, 200, 64), (1, 1, 140, 150, 80)])
def test_attention_prescaled_query_overflow_fallback(B, H, Lq, Lk, d):
    """a late key ~250 (log2 domain) above the first tile's row maximum: exp2 against the first tile's reference is inf, the one-time row-sum
    check after the loop catches it and the workgroup repeats with maximum tracking -- results as accurate as ever; and mid-stream
    spikes that stay finite (reference never moved: P up to 2^30) keep their relative precision"""
    K.case_attention("cpu", B, H, Lq, Lk, d, spike=True, prescaled=True, spike_overflow=True)
    K.case_attention("cpu", B, H, Lq, Lk, d, spike=True, spike_at=(70, 130), prescaled=True)


@pytest.mark.parametrize("B,H,Lq,Lk,d,at", [(1, 1, 70, 520, 40, (130, 300)), (1, 2, 40, 450, 64, (70, 200, 330)), (1, 1, 40, 400, 80, (100,)),
                                            (1, 1, 40, 390, 32, (64, 128, 320)), (1, 1, 33, 64, 40, None), (1, 1, 33, 128, 64, (70,))])
def test_attention_pipelined_forward_steady_state(B, H, Lq, Lk, d, at):
    """enough full key tiles for several trips of the branch-free two-body steady-state loop of the software-pipelined forward
    (attention_fwd.hip), reference moves INSIDE it (the deferred O rescale), and the one- / two-tile corner cases"""
    from pcm_amd import capi
    dll = capi.lib().dll
    dll.pcm_debug_attn_fwd_variant(1)          # (the default picks a kernel by shape: force the pipelined one)
    try:
        K.case_attention("cpu", B, H, Lq, Lk, d, spike=True, spike_at=at)
    finally:
        dll.pcm_debug_attn_fwd_variant(-1)


def test_attention_first_forward_kernel_still_correct():
    """variant 0 (attention.hip's dependent-chain forward) stays selectable for A/B timing and is what head dim 160 uses"""
    from pcm_amd import capi
    dll = capi.lib().dll
    dll.pcm_debug_attn_fwd_variant(0)
    try:
        K.case_attention("cpu", 1, 1, 70, 330, 40, spike=True, spike_at=(100,))
    finally:
        dll.pcm_debug_attn_fwd_variant(-1)


@pytest.mark.parametrize("B,H,Lq,Lk,d", [(2, 5, 300, 200, 40), (1, 8, 260, 130, 64), (3, 3, 130, 77, 80)])
def test_attention_forward_xcd_aware_block_map(B, H, Lq, Lk, d):
    """the opt-in XCD-aware block -> (query block, head, image) map of the first forward kernel is a bijection: full groups of eight
    (image, head) pairs are permuted, a ragged last group (B * H not a multiple of 8) keeps the plain map -- same results as the plain map"""
    from pcm_amd import capi
    dll = capi.lib().dll
    dll.pcm_debug_attn_fwd_variant(0)
    dll.pcm_debug_attn_xcd_remap(1)
    try:
        K.case_attention("cpu", B, H, Lq, Lk, d, spike=True)
    finally:
        dll.pcm_debug_attn_xcd_remap(0)
        dll.pcm_debug_attn_fwd_variant(-1)


def test_lora_repack():
    K.case_lora_repack("cpu")


def test_adv_kernels(golden):
    K.case_adv_kernels("cpu", golden)


def test_discriminator_heads():
    K.case_discriminator_heads("cpu")


@pytest.mark.slow
def test_teacher_input_grad():
    K.case_teacher_input_grad("cpu")


def test_emulator_enforces_launch_limits():
    """The emulator refuses what a gfx950 CU refuses (round-1 bug class: a 164 864 B dynamic-LDS request passed every CPU test):
    > 64 KB of dynamic LDS without the per-kernel attribute, > 160 KB at all, > 1024 threads, grid.y > 65535."""
    from pcm_amd import capi
    dll = capi.lib().dll
    import ctypes
    f = dll.pcm_emu_try_launch
    f.argtypes = [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_long]
    assert f(65536, 256, 1, -1) == 0
    assert f(65537, 256, 1, -1) != 0                 # needs hipFuncSetAttribute first
    assert f(0, 2048, 1, -1) != 0 and f(0, 256, 70000, -1) != 0
    assert f(0, 256, 1, 163841) == -1                # the attribute itself is bounded by the 160 KiB of a CU
    assert f(163840, 256, 1, 163840) == 0
    assert f(164864, 256, 1, -1) != 0                # the round-1 SDXL request (64 rows x (1280 + 8) bf16)


# wide-head shapes: SDXL level-2 and SD3 widths (H*d = 1280 / 1536), d = 32; ragged tails on both sequence axes.  (Round 1 packed transposed
# operand images for these -- the LDS over-subscription that broke the full-size GPU tests; the operands are LDS transpose reads now.)
@pytest.mark.parametrize("B,H,Lq,Lk,d", [(1, 2, 130, 130, 40), (1, 1, 70, 200, 80), (1, 2, 154, 154, 64), (1, 20, 70, 70, 64),
                                         (1, 24, 130, 100, 64), (1, 17, 70, 70, 32)])
def test_attention_wide_heads_and_ragged_tiles(B, H, Lq, Lk, d):
    from pcm_amd import capi
    assert capi.lib().dll.pcm_attn_workspace_bytes(B, H, Lq, Lk, d, 1) == 0     # no call needs a workspace any more
    K.case_attention("cpu", B, H, Lq, Lk, d, spike=True)


def test_abi_rejects_bad_arguments_with_a_message():
    """error behaviour of the boundary: a bad call returns a non-zero code and leaves a message naming the entry point
    (capi raises PcmError with it); nothing is launched."""
    L, S = capi.lib(), capi.Lib.stream
    p = ops.ptr
    bf = torch.zeros(64, 64, dtype=torch.bfloat16)
    f32 = torch.zeros(64, 64)
    f64 = torch.zeros(64, 64, dtype=torch.float64)
    idx = torch.zeros(4, dtype=torch.int64)
    cases = [
        ("pcm_groupnorm_stats", (p(bf), p(f64), 1, 64, 60, 32, S())),                                   # C % G != 0
        ("pcm_layernorm_fwd", (p(bf), p(f32), p(f32), p(bf), p(f32), p(f32), 1, 4096, 1e-5, S())),      # C > 2048
        ("pcm_layernorm_mod_fwd", (p(bf), p(f32), p(f32), p(bf), p(f32), p(f32), 64, 64, 1e-6, 48, S())),  # M not B * rows_per_batch
        ("pcm_rowgate_fma", (p(bf), p(f32), None, p(bf), 64, 64, 48, S())),
        ("pcm_attn_fwd", (p(bf), p(bf), p(bf), p(bf), p(f32), 1, 1, 64, 64, 64, 60, 64, 64, 0.125, S())),   # ld % 8 != 0
        ("pcm_patchify2x2", (p(f32), p(bf), 1, 4, 3, 4, 0, S())),                                        # odd H
        ("pcm_unpatchify2x2", (p(f32), p(f32), 1, 4, 4, 4, 2, S())),                                     # order not in {0, 1}
        ("pcm_fm_sampler_step", (p(f32), None, 1.0, p(f32), 0.0, 0.0, None, p(f32), 64, S())),           # sigma must be > 0 (division)
        ("pcm_fm_phase_jump", (p(f32), 0, p(f32), p(idx), p(f32), p(f64), p(idx), 0, 0, p(f64), None, None, 4, 16, S())),   # no phase edges
        ("pcm_gelu_tanh_fwd", (p(bf), p(bf), 12, S())),                                                   # n % 8 != 0
        ("pcm_mod_grad", (p(bf), p(bf), p(f32), None, p(f32), None, 1, 64, 64, S())),                    # mean without rstd
        ("pcm_geglu_fwd", (p(bf), p(bf), 4, 12, S())),
        ("pcm_concat_channels", (p(bf), 12, p(bf), 8, p(bf), 4, S())),
    ]
    for name, args in cases:
        with pytest.raises(capi.PcmError) as ei:
            L.call(name, *args)
        assert name.replace("_fwd", "") .split("pcm_")[1][:6] in str(ei.value) or "pcm_" in str(ei.value), (name, str(ei.value))
    a = capi.WgradArgs()
    a.big, a.small_, a.out, a.M, a.G, a.lds_, a.ldb, a.mode = p(bf), p(bf), p(f32), 64, 64, 32, 64, capi.SEG_PLAIN      # small ld < 64
    with pytest.raises(capi.PcmError, match="pcm_lora_wgrad_bf16"):
        L.call("pcm_lora_wgrad_bf16", ctypes.byref(a), S())
    with pytest.raises(capi.PcmError, match="not exported"):
        L.call("pcm_no_such_entry")


@pytest.mark.parametrize("which", K.GEMM_EPI_FUSION_CASES)
def test_gemm_epilogue_fusions_second_output_and_groupnorm_statistics(which):
    """abi 5: a skip tensor's second home (out2) and the next GroupNorm's statistics (chstats) from the producing contraction's epilogue"""
    K.case_gemm_epilogue_fusions("cpu", which)


@pytest.mark.parametrize("which,form", [("lora_res", 1), ("tail_strided", 1), ("plain", 2), ("res_tail_strided", 2)])
def test_gemm_weights_stationary_kernel(which, form):
    """csrc/gemm_ws.hip: the short-K projections of the 64x64 level with the weight slice held in registers (form 1: four waves x 80 columns,
    form 2: eight waves x 40 columns); all four cases of both forms run on the GPU"""
    import kernel_cases as KC
    KC.case_gemm_ws("cpu", which, form)
