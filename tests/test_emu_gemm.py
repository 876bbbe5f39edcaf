"""Host-emulation checks of pcm_gemm_bf16 index math (MFMA fragment layout, LDS swizzle, im2col
masks, LoRA second segment, epilogue) against plain torch fp32 on the same bf16-rounded inputs."""
import pytest
import torch
import torch.nn.functional as F

from emu_lib import emu_lib
from pcm_amd import capi, ops


@pytest.fixture(autouse=True)
def _use_emu():
    capi.set_lib(emu_lib())
    yield
    capi.set_lib(None)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).bfloat16()


def test_plain_gemm_bias_residual_lora():
    M, N, K = 200, 192, 136   # ragged M, N%128!=0, K tail (136 = 2*64 + 8)
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1)
    t, bl = rnd(M, 64, seed=3), rnd(N, 64, seed=4, scale=0.1)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(5))
    res = rnd(M, N, seed=6)
    out = torch.empty(M, N, dtype=torch.bfloat16)
    ops.gemm([ops.Seg(x, w), ops.Seg(t, bl)], M, N, out, bias=bias, residual=res)
    ref = x.float() @ w.float().T + t.float() @ bl.float().T + bias + res.float()
    assert torch.allclose(out.float(), ref, rtol=1e-2, atol=2e-2), (out.float() - ref).abs().max()
    # asymmetric check: fp32 output, silu, alpha
    out32 = torch.empty(M, N, dtype=torch.float32)
    ops.gemm([ops.Seg(x, w)], M, N, out32, act=capi.ACT_SILU, alpha=0.5)
    ref = F.silu(0.5 * (x.float() @ w.float().T))
    assert torch.allclose(out32, ref, rtol=1e-4, atol=1e-4), (out32 - ref).abs().max()


@pytest.mark.parametrize("stride,src_mode", [(1, capi.SRC_DIRECT), (2, capi.SRC_DIRECT),
                                             (1, capi.SRC_UPSAMPLE2), (1, capi.SRC_ZEROINS2)])
def test_conv3x3_implicit_gemm(stride, src_mode):
    B, Hs, Ws, Ci, Co = 2, 6, 5, 64, 64
    x = rnd(B, Hs, Ws, Ci, seed=7)
    w = rnd(Co, Ci, 3, 3, seed=8, scale=0.05)
    xn = x.float().permute(0, 3, 1, 2)
    if src_mode == capi.SRC_UPSAMPLE2:
        xv = F.interpolate(xn, scale_factor=2.0, mode="nearest")
    elif src_mode == capi.SRC_ZEROINS2:
        xv = torch.zeros(B, Ci, 2 * Hs, 2 * Ws)
        xv[:, :, ::2, ::2] = xn
    else:
        xv = xn
    ref = F.conv2d(xv, w.float(), None, stride=stride, padding=1)
    Ho, Wo = ref.shape[2], ref.shape[3]
    M = B * Ho * Wo
    wk = w.permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous()
    temb = rnd(B, Co, seed=9)
    out = torch.empty(M, Co, dtype=torch.float32)
    ops.gemm([ops.Seg(x, wk, conv=dict(Hs=Hs, Ws=Ws, stride=stride, src_mode=src_mode))], M, Co, out,
             rowvec=temb, rows_per_batch=Ho * Wo, Ho=Ho, Wo=Wo)
    ref = ref + temb.float()[:, :, None, None]
    ref = ref.permute(0, 2, 3, 1).reshape(M, Co)
    assert torch.allclose(out, ref, rtol=1e-3, atol=1e-3), (out - ref).abs().max()


def test_gemm_rejects_bad_args():
    x, w = rnd(8, 12), rnd(8, 12)
    with pytest.raises(capi.PcmError):
        ops.gemm([ops.Seg(x, w)], 8, 8, torch.empty(8, 8, dtype=torch.bfloat16))  # K%8 != 0


def test_split_k_plain_and_conv():
    """Under-filled grids with a long K loop take the split-K path (slab reduction + finalize)."""
    M, N, K = 130, 192, 1032
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    t, bl = rnd(M, 64, seed=3), rnd(N, 64, seed=4, scale=0.1)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(5))
    res = rnd(M, N, seed=6)
    out = torch.empty(M, N, dtype=torch.float32)
    ops.gemm([ops.Seg(x, w), ops.Seg(t, bl)], M, N, out, bias=bias, residual=res, act=capi.ACT_SILU)
    ref = F.silu(x.float() @ w.float().T + t.float() @ bl.float().T + bias) + res.float()
    assert torch.allclose(out, ref, rtol=1e-3, atol=2e-3), (out - ref).abs().max()
    B, Hs, Ws, Ci, Co = 1, 5, 6, 128, 64
    xc = rnd(B, Hs, Ws, Ci, seed=7)
    wc = rnd(Co, Ci, 3, 3, seed=8, scale=0.05)
    refc = F.conv2d(xc.float().permute(0, 3, 1, 2), wc.float(), None, padding=1).permute(0, 2, 3, 1).reshape(-1, Co)
    outc = torch.empty(B * Hs * Ws, Co, dtype=torch.bfloat16)
    ops.gemm([ops.Seg(xc, wc.permute(0, 2, 3, 1).reshape(Co, -1).contiguous(), conv=dict(Hs=Hs, Ws=Ws))], B * Hs * Ws, Co, outc, Ho=Hs, Wo=Ws)
    assert torch.allclose(outc.float(), refc, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("lazy_dma", ["0", "1"])
@pytest.mark.parametrize("which", __import__("kernel_cases").GEMM_BIG_CASES)
def test_big_tile_kernel(which, lazy_dma, monkeypatch):
    """gemm8p (256-row phased tile, counted LDS-DMA waits).  Run with the DMA landing at issue AND only at the counted
    wait: agreement under both rules out a dependence on the landing time (RAW too-early reads, WAR early restages)."""
    import kernel_cases as KC
    monkeypatch.setenv("PCM_EMU_LAZY_DMA", lazy_dma)
    excess, err = KC.case_gemm_big("cpu", which)
    assert excess <= 0, (which, lazy_dma, err)


@pytest.mark.parametrize("md,co", [(1, 0), (1, 1)])
@pytest.mark.parametrize("which", ["conv", "conv_conv", "conv_splitk"])
def test_big_tile_conv_address_variants(which, md, co):
    """gemm8p variants of the stride-1 direct 3x3 view (gemm8p.hip): the shipped one is the per-tap re-key in tap-outer order (what the other
    gemm8p tests run); the mask + delta addressing (md = 1) and the chunk-outer K order (co = 1) are A/B options that measured slower on the
    GPU and must keep giving the same results"""
    import kernel_cases as KC
    from pcm_amd import capi
    dll = capi.lib().dll
    dll.pcm_debug_gemm_conv_md(md)
    dll.pcm_debug_gemm_conv_order(co)
    try:
        excess, err = KC.case_gemm_big("cpu", which)
    finally:
        dll.pcm_debug_gemm_conv_md(-1)
        dll.pcm_debug_gemm_conv_order(-1)
    assert excess <= 0, (which, md, co, err)


def test_big_tile_conv_order_by_shape():
    """conv order 2 = by shape: the gemm8p launcher takes the chunk-outer K order on feature maps of at most 8x8 and the tap-outer order
    otherwise (an opt-in: the whole-step A/B of round 3 showed no gain); the default (-1) is tap-outer everywhere.  Same results either way."""
    import kernel_cases as KC
    from pcm_amd import capi
    dll = capi.lib().dll
    dll.pcm_debug_gemm_conv_md(-1)
    try:
        for order, expect in ((2, {"conv": 0, "conv_small_map": 3}), (-1, {"conv": 0, "conv_small_map": 0})):
            dll.pcm_debug_gemm_conv_order(order)
            for which, variant in expect.items():
                excess, err = KC.case_gemm_big("cpu", which)
                assert excess <= 0, (which, err)
                assert dll.pcm_debug_last_gemm8p_variant() == variant, (order, which, dll.pcm_debug_last_gemm8p_variant())
    finally:
        dll.pcm_debug_gemm_conv_order(-1)


# (K % 160 == 0 or K % 128 == 0: the K-split kernel -- 160- / 128-column pieces, one or two 16-row fragments per block (emulator: two from M = 128),
# 1 / 2 / 4 / 8 waves; otherwise the chunked streaming kernel)
@pytest.mark.parametrize("M,K", [(200, 320), (64, 1280), (130, 768), (77, 192), (40, 640), (33, 2560), (20, 960), (150, 1536), (129, 6144), (50, 2048), (140, 640), (33, 5120), (260, 2560), (300, 6144)])
def test_rank64_streaming_kernel(M, K):
    import kernel_cases as KC
    assert KC.case_gemm_n64("cpu", M, K) <= 0


# patch geometries of conv_r64.hip: 8-wide (16 rows per patch), 16-wide (8 rows), 32-wide (4 rows), 64-wide (2 rows), 128-wide (two 64-column
# patches per row pair), two images, two channel chunks; and shapes it must leave to the generic tile (8x8 image: no 128-pixel patch)
@pytest.mark.parametrize("lazy_dma", ["0", "1"])
@pytest.mark.parametrize("B,H,W,C", [(1, 16, 8, 64), (2, 8, 16, 128), (1, 4, 32, 64), (2, 2, 64, 64), (1, 2, 128, 64), (1, 16, 16, 192)])
def test_conv_lora_down_projection_halo_kernel(B, H, W, C, lazy_dma, monkeypatch):
    import kernel_cases as KC
    monkeypatch.setenv("PCM_EMU_LAZY_DMA", lazy_dma)
    KC.case_conv_r64("cpu", B, H, W, C)


def test_conv_lora_down_projection_fallback_shapes():
    import kernel_cases as KC
    KC.case_conv_r64("cpu", 2, 8, 8, 64, expect_kernel=False)


@pytest.mark.parametrize("lazy_dma", [False, True])
@pytest.mark.parametrize("which", __import__("kernel_cases").GEMM_4W_CASES)
def test_short_k_kernel(which, lazy_dma, monkeypatch):
    """gemm4w.hip under both extremes of the LDS-DMA landing time (at issue / only at the counted wait that retires it)"""
    if lazy_dma:
        monkeypatch.setenv("PCM_EMU_LAZY_DMA", "1")
    import kernel_cases as KC
    excess, err = KC.case_gemm_4w("cpu", which)
    assert excess <= 0, (which, excess, err)


def test_fused_geglu_epilogue_short_k_kernel():
    import kernel_cases as KC
    assert KC.case_gemm_geglu("cpu", big_mode=3) <= 0
    assert KC.case_gemm_geglu("cpu", M=256, K=64, inner=160, big_mode=3) <= 0


def test_fused_geglu_epilogue():
    import kernel_cases as KC
    assert KC.case_gemm_geglu("cpu") <= 0
    assert KC.case_gemm_geglu("cpu", M=256, K=64, inner=128) <= 0


@pytest.mark.parametrize("M,N,Ks,act,bias,f32", [(16, 64, (128,), 0, True, False), (2, 144, (96,), 1, True, False), (13, 320, (64, 64), 1, True, False),
                                                   (16, 32, (32,), 0, False, True), (4, 100, (160, 32), 0, True, True), (1, 16, (2048,), 0, True, False)])
def test_batch_row_projection_kernel(M, N, Ks, act, bias, f32):
    """gemm_smallm.hip: M <= 16 (ragged M and N, one or two K segments, K steps not a multiple of waves x unroll)"""
    import kernel_cases as KC
    assert KC.case_gemm_smallm("cpu", M, N, Ks, act, bias, f32, alpha=0.5 if f32 else 1.0) <= 0
