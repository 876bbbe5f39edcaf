"""GPU parity of pcm_gemm_bf16 vs torch fp32 on the same bf16-rounded inputs (through the C ABI)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _hip():
    from pcm_amd import capi
    assert torch.cuda.is_available()
    # the TOOLS build (same kernels + the pcm_debug_* hooks these cases force kernel families with); the product library's own run of the
    # hook-free cases is tests/test_gpu_product_lib.py, and every step-level GPU test runs on it
    capi.set_lib(capi.tools_lib())  # raises loudly if libpcm_hip_tools.so is missing
    yield
    capi.set_lib(None)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).bfloat16().cuda()


@pytest.mark.parametrize("M,N,K", [(200, 192, 136), (4096, 320, 320), (1024, 1280, 1280), (77 * 2, 640, 768),
                                   (16, 1280, 320), (8192, 2560, 320)])
def test_plain_gemm(M, N, K):
    from pcm_amd import capi, ops
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    t, bl = rnd(M, 64, seed=3), rnd(N, 64, seed=4, scale=0.05)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(5)).cuda()
    res = rnd(M, N, seed=6)
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")
    ops.gemm([ops.Seg(x, w), ops.Seg(t, bl)], M, N, out, bias=bias, residual=res)
    ref = x.float() @ w.float().T + t.float() @ bl.float().T + bias + res.float()
    torch.cuda.synchronize()
    err = (out - ref).abs().max().item()
    assert err < 2e-3 * max(1.0, ref.abs().max().item()), err
    outb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm([ops.Seg(x, w)], M, N, outb, act=capi.ACT_SILU, alpha=0.5)
    ref = F.silu(0.5 * (x.float() @ w.float().T))
    assert torch.allclose(outb.float(), ref, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("stride,src_mode", [(1, 0), (2, 0), (1, 1), (1, 2)])
@pytest.mark.parametrize("B,Hs,Ci,Co", [(2, 16, 64, 64), (2, 32, 320, 640), (1, 8, 1280, 1280)])
def test_conv3x3(stride, src_mode, B, Hs, Ci, Co):
    from pcm_amd import capi, ops
    Ws = Hs
    x = rnd(B, Hs, Ws, Ci, seed=7)
    w = rnd(Co, Ci, 3, 3, seed=8, scale=0.02)
    xn = x.float().permute(0, 3, 1, 2)
    if src_mode == capi.SRC_UPSAMPLE2:
        xv = F.interpolate(xn, scale_factor=2.0, mode="nearest")
    elif src_mode == capi.SRC_ZEROINS2:
        xv = torch.zeros(B, Ci, 2 * Hs, 2 * Ws, device="cuda")
        xv[:, :, ::2, ::2] = xn
    else:
        xv = xn
    ref = F.conv2d(xv, w.float(), None, stride=stride, padding=1)
    Ho, Wo = ref.shape[2], ref.shape[3]
    M = B * Ho * Wo
    wk = w.permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous()
    temb = rnd(B, Co, seed=9)
    out = torch.empty(M, Co, dtype=torch.float32, device="cuda")
    ops.gemm([ops.Seg(x, wk, conv=dict(Hs=Hs, Ws=Ws, stride=stride, src_mode=src_mode))], M, Co, out,
             rowvec=temb, rows_per_batch=Ho * Wo, Ho=Ho, Wo=Wo)
    ref = (ref + temb.float()[:, :, None, None]).permute(0, 2, 3, 1).reshape(M, Co)
    torch.cuda.synchronize()
    err = (out - ref).abs().max().item()
    assert err < 3e-3 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("which", __import__("kernel_cases").GEMM_BIG_CASES)
def test_big_tile_kernel(which):
    """gemm8p (256-row phased tile) on hardware: buffer-resource zero fill, LDS-DMA above 64 KB, counted waits."""
    import kernel_cases as KC
    for _ in range(3):   # repeated: a landing-time race would show as run-to-run differences
        excess, err = KC.case_gemm_big("cuda", which)
        assert excess <= 0, (which, err)


def test_big_tile_kernel_large_shapes_match_small_tile():
    """Full-size SD1.5 shapes: the phased kernel must agree with the 4-wave kernel (both fp32-accumulate the same bf16 data)."""
    from pcm_amd import capi, ops
    dll = capi.lib().dll
    g = torch.Generator(device="cuda").manual_seed(3)
    for (B, Hs, Ci, Co) in [(4, 64, 320, 320), (4, 32, 640, 640), (8, 16, 1280, 1280), (4, 32, 1280, 640)]:
        x = torch.randn(B, Hs, Hs, Ci, device="cuda", generator=g).bfloat16()
        w = (torch.randn(Co, 9 * Ci, device="cuda", generator=g) * 0.02).bfloat16()
        t = torch.randn(B * Hs * Hs, 64, device="cuda", generator=g).bfloat16()
        bl = (torch.randn(Co, 64, device="cuda", generator=g) * 0.1).bfloat16()
        M = B * Hs * Hs
        outs = []
        for mode in (0, 2):
            dll.pcm_debug_gemm_big_mode(mode)
            out = torch.empty(M, Co, device="cuda", dtype=torch.bfloat16)
            ops.gemm([ops.Seg(x, w, conv=dict(Hs=Hs, Ws=Hs)), ops.Seg(t, bl)], M, Co, out, Ho=Hs, Wo=Hs)
            outs.append(out.float())
        dll.pcm_debug_gemm_big_mode(1)
        torch.cuda.synchronize()
        d = (outs[0] - outs[1]).abs().max().item()
        assert d <= 2e-2 * outs[0].abs().max().item() + 1e-3, (B, Hs, Ci, Co, d)


@pytest.mark.parametrize("M,K", [(65536, 320), (4096, 1280), (1232, 768), (16384, 640), (200, 192), (131072, 320), (65536, 2560), (8192, 1280), (2048, 2560), (1024, 960), (32768, 640),
                                 (8192, 6144), (8192, 5120), (4096, 10240), (8192, 1536), (616, 2048), (9000, 1280), (4096, 5120), (16384, 1536),
                                 # the K-split kernel at 64 rows per block with 8 / 16 waves (RF = 4: M >= 16384; 128 KB of static LDS, one block per CU)
                                 (16384, 1280), (16384, 2560), (16384, 5120)])
def test_rank64_streaming_kernel(M, K):
    """(which kernel of gemm_n64.hip runs depends on M as well as K -- K-split with 16 / 32 / 64 rows per block up to M = 16384, the chunked
    streaming kernel above -- so per-row results of the rank-64 projections can differ in the last bit between batch sizes; the
    half-batch == half-of-the-full-batch bit identity of tests/test_emu_unet.py holds for the forward's base GEMMs, not for this path)"""
    import kernel_cases as KC
    assert KC.case_gemm_n64("cuda", M, K) <= 0


@pytest.mark.parametrize("B,H,W,C", [(4, 64, 64, 320), (4, 32, 32, 640), (8, 16, 16, 1280), (2, 128, 128, 320), (3, 16, 8, 64), (2, 8, 8, 1280)])
def test_conv_lora_down_projection_halo_kernel(B, H, W, C):
    import kernel_cases as KC
    KC.case_conv_r64("cuda", B, H, W, C, expect_kernel=not (H == 8 and W == 8))


@pytest.mark.parametrize("which", __import__("kernel_cases").GEMM_4W_CASES)
def test_short_k_kernel(which):
    import kernel_cases as KC
    for _ in range(3):   # repeated: a landing-time race would show as run-to-run differences
        excess, err = KC.case_gemm_4w("cuda", which)
        assert excess <= 0, (which, excess, err)


def test_short_k_kernel_matches_big_tile_on_step_shapes():
    """the planner's short-K choice (gemm4w.hip) against the 256-row tile on launches of the bs-16 step: same operands, same epilogue"""
    from pcm_amd import capi, ops
    dll = capi.lib().dll
    g = torch.Generator().manual_seed(3)
    for (M, N, K, r) in ((32768, 320, 320, 64), (16384, 2560, 320, 64), (8192, 1280, 1280, 0), (4096, 960, 320, 192)):
        x, w = torch.randn(M, K, generator=g).bfloat16().cuda(), (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
        segs = [ops.Seg(x, w)]
        if r:
            segs.append(ops.Seg(torch.randn(M, r, generator=g).bfloat16().cuda(), (torch.randn(N, r, generator=g) * 0.05).bfloat16().cuda()))
        res = torch.randn(M, N, generator=g).bfloat16().cuda()
        outs = []
        for mode in (2, 3):
            dll.pcm_debug_gemm_big_mode(mode)
            try:
                o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
                ops.gemm(segs, M, N, o, residual=res)
                plan = dll.pcm_debug_last_gemm_plan()
                assert (plan > 10000) == (mode == 3), (mode, plan)
                outs.append(o.float())
            finally:
                dll.pcm_debug_gemm_big_mode(1)
        # same K order and fp32 accumulation: equal up to one bf16 ulp of a few outputs
        assert float((outs[0] - outs[1]).abs().max()) <= 2.0 ** -7 * float(outs[0].abs().max()), (M, N, K)
        assert float(((outs[0] - outs[1]).norm() / outs[0].norm())) < 1e-3


def test_fused_geglu_epilogue_short_k_kernel():
    import kernel_cases as KC
    assert KC.case_gemm_geglu("cuda", big_mode=3) <= 0
    assert KC.case_gemm_geglu("cuda", M=32768, K=320, inner=1280, big_mode=3) <= 0


def test_fused_geglu_epilogue():
    import kernel_cases as KC
    assert KC.case_gemm_geglu("cuda") <= 0
    assert KC.case_gemm_geglu("cuda", M=8192, K=1280, inner=5120) <= 0
    assert KC.case_gemm_geglu("cuda", M=32768, K=320, inner=1280) <= 0


@pytest.mark.parametrize("M,N,Ks,act,bias,f32", [(16, 1280, (320,), 1, True, False), (16, 1280, (1280, 64), 0, True, False), (16, 640, (1280,), 0, True, False),
                                                   (2, 9216, (1536,), 0, True, False), (4, 9216, (1536,), 0, True, False), (16, 64, (1280,), 0, False, False),
                                                   (8, 2816, (1280,), 1, True, True), (15, 324, (96, 64), 1, True, False)])
def test_batch_row_projection_kernel(M, N, Ks, act, bias, f32):
    """gemm_smallm.hip on the batch-row shapes of the four configs (time-embedding MLPs, time_emb_proj with / without the LoRA segment, adaLN)"""
    import kernel_cases as KC
    for _ in range(2):
        assert KC.case_gemm_smallm("cuda", M, N, Ks, act, bias, f32) <= 0


@pytest.mark.parametrize("which", __import__("kernel_cases").GEMM_EPI_FUSION_CASES)
def test_gemm_epilogue_fusions_second_output_and_groupnorm_statistics(which):
    """abi 5: a skip tensor's second home (out2) and the next GroupNorm's statistics (chstats) from the producing contraction's epilogue"""
    import kernel_cases as KC
    KC.case_gemm_epilogue_fusions("cuda", which)


@pytest.mark.parametrize("which,form", [(w, 1) for w in __import__("kernel_cases").GEMM_WS_CASES] + [(w, 2) for w in __import__("kernel_cases").GEMM_WS8_CASES])
def test_gemm_weights_stationary_kernel(which, form):
    """csrc/gemm_ws.hip: the short-K projections of the 64x64 level with the weight slice held in registers (four-wave and eight-wave form)"""
    import kernel_cases as KC
    KC.case_gemm_ws("cuda", which, form)
