"""The PRODUCT library (lib/libpcm_hip.so) at kernel level: no pcm_debug_* symbol, no environment switch, and the hook-free parity cases
of tests/kernel_cases.py on it (tests/test_gpu_kernels.py / test_gpu_gemm.py run the TOOLS build of the same sources, because many of their
cases force kernel families through the hooks).  The step-level GPU tests (test_gpu_step.py, test_gpu_bench_config.py, ...) all run on
the product library too."""
import subprocess

import pytest
import torch

import kernel_cases as K

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _product():
    from pcm_amd import capi
    capi.set_lib(None)
    assert torch.cuda.is_available()
    lib = capi.lib()
    assert lib.path == capi.DEFAULT_LIB
    yield
    torch.cuda.synchronize()


def test_product_library_exports_no_debug_hooks():
    from pcm_amd import capi
    out = subprocess.run(["nm", "-D", capi.DEFAULT_LIB], capture_output=True, text=True).stdout
    syms = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert syms and not [s for s in syms if s.startswith("pcm_debug")], [s for s in syms if s.startswith("pcm_debug")]
    assert not hasattr(capi.lib().dll, "pcm_debug_gemm_big_mode")
    assert capi.lib().dll.pcm_abi_version() >= 5


def test_hook_free_kernel_cases_on_the_product_library():
    K.case_groupnorm("cuda", 2, 4096, 320, 32, 1)
    K.case_layernorm("cuda", 4099, 320)
    K.case_elementwise("cuda")
    K.case_edge_convs("cuda", B=2, H=64, W=64, C0=320)
    K.case_optim("cuda")
    K.case_pack("cuda")
    K.case_wgrad_plain("cuda", 4096, 320, 320)
    K.case_wgrad_conv("cuda", 2, 16, 16, 320, 1, 0)
    K.case_wgrad_dense("cuda", 2, 32, 32, 128, 128)
    K.case_lora_repack("cuda")
    K.case_reproducible_reductions("cuda", big=True)


@pytest.mark.parametrize("B,H,Lq,Lk,d,prescaled", [(1, 8, 1024, 1024, 40, True), (1, 8, 1024, 1024, 40, False), (2, 8, 256, 77, 80, True),
                                                   (1, 10, 2176, 2176, 64, True), (1, 20, 1024, 1024, 64, False)])
def test_attention_on_the_product_library(B, H, Lq, Lk, d, prescaled):
    K.case_attention("cuda", B, H, Lq, Lk, d, True, prescaled=prescaled)


def test_gemm_plan_code_is_a_pure_query():
    """pcm_gemm_plan_code (include/pcm_hip.h): the kernel family / K split a call takes, as a function of its arguments -- what bench.py's
    roofline leg classes launches with; the same arguments give the same code before and after unrelated launches"""
    import ctypes as C
    from pcm_amd import capi, ops
    x, w = torch.randn(131072, 320, device="cuda").to(ops.BF16), torch.randn(320, 320, device="cuda").to(ops.BF16)
    out = torch.empty(131072, 320, dtype=ops.BF16, device="cuda")
    prof, ops.GEMM_PROFILE = ops.GEMM_PROFILE, []
    try:
        ops.gemm([ops.Seg(x, w)], 131072, 320, out)
        code = ops.GEMM_PROFILE[-1][4]
        ops.gemm([ops.Seg(x[:32], w)], 32, 320, out[:32])           # an unrelated launch in between
        ops.gemm([ops.Seg(x, w)], 131072, 320, out)
        assert ops.GEMM_PROFILE[-1][4] == code == 5001, (code, ops.GEMM_PROFILE[-1][4])
    finally:
        ops.GEMM_PROFILE = prof


def test_half_product_library():
    """lib/libpcm_hip_f16.so (what --mixed_precision=fp16 and bench.py --precision fp16 load): no hooks either, a few hook-free cases"""
    from pcm_amd import capi, precision
    precision.set_precision("fp16")
    try:
        assert capi.lib().path == capi.F16_LIB and capi.lib().act_dtype == 1 and not hasattr(capi.lib().dll, "pcm_debug_gemm_big_mode")
        K.case_groupnorm("cuda", 2, 1024, 640, 32, 0, eps=1e-6)
        K.case_wgrad_plain("cuda", 4096, 320, 320)
        K.case_attention("cuda", 1, 8, 1024, 1024, 40, prescaled=True)
        K.case_attention("cuda", 1, 8, 1024, 1024, 40)
    finally:
        torch.cuda.synchronize()
        precision.set_precision("bf16")
