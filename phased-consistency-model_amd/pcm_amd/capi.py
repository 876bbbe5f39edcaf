"""ctypes binding of libpcm_hip.so (include/pcm_hip.h).

The product path has exactly one implementation per op — the HIP library.  ``lib()`` raises
``RuntimeError`` when it is missing; there is no torch-op or CPU fallback.  Tests may construct
``Lib(path)`` on another build of the SAME sources (tests/emu) to check index math on the host.
"""
import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "lib", "libpcm_hip.so")
F16_LIB = os.path.join(HERE, "lib", "libpcm_hip_f16.so")      # the same sources compiled with -DPCM_ACT_F16 (pcm_amd/precision.py)
# TOOLS builds (-DPCM_TOOLS): the same kernels plus the pcm_debug_* hooks / environment switches / A-B kernel variants that tools/ and
# the hook-using tests need (csrc/pcm_common.h); never loaded by the product path
TOOLS_LIB = os.path.join(HERE, "lib", "libpcm_hip_tools.so")
TOOLS_F16_LIB = os.path.join(HERE, "lib", "libpcm_hip_tools_f16.so")

PCM_BF16, PCM_F32 = 0, 1
REDUCE_WS_BYTES = 32768     # include/pcm_hip.h PCM_REDUCE_WS_BYTES
ACT_NONE, ACT_SILU, ACT_LEAKY, ACT_GEGLU = 0, 1, 2, 3
SEG_PLAIN, SEG_CONV3X3 = 0, 1
SRC_DIRECT, SRC_UPSAMPLE2, SRC_ZEROINS2 = 0, 1, 2

vp = C.c_void_p


class GemmSeg(C.Structure):
    _fields_ = [("a", vp), ("w", vp), ("K", C.c_int), ("lda", C.c_int), ("mode", C.c_int),
                ("Hs", C.c_int), ("Ws", C.c_int), ("C", C.c_int), ("stride", C.c_int),
                ("src_mode", C.c_int)]


class GemmEpi(C.Structure):
    _fields_ = [("M", C.c_int), ("N", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int), ("bias", vp),
                ("rowvec", vp), ("rows_per_batch", C.c_int), ("residual", vp), ("ldr", C.c_int),
                ("out", vp), ("ldo", C.c_int), ("out_dtype", C.c_int), ("act", C.c_int),
                ("alpha", C.c_float), ("workspace", vp), ("workspace_bytes", C.c_size_t),
                ("pre_out", vp), ("pre_rows", C.c_int), ("ldp", C.c_int),
                ("out2", vp), ("ldo2", C.c_int), ("chstats", vp), ("stats_rows", C.c_int)]      # abi 5


class WgradArgs(C.Structure):
    _fields_ = [("big", vp), ("ldb", C.c_int), ("G", C.c_int), ("mode", C.c_int), ("Hs", C.c_int),
                ("Ws", C.c_int), ("C", C.c_int), ("stride", C.c_int), ("src_mode", C.c_int),
                ("Ho", C.c_int), ("Wo", C.c_int), ("small_", vp), ("lds_", C.c_int), ("M", C.c_int),
                ("out", vp), ("g_stride", C.c_long), ("r_stride", C.c_long), ("out_conv", C.c_int),
                ("alpha", C.c_float), ("workspace", vp), ("workspace_bytes", C.c_size_t)]


class PackDesc(C.Structure):
    _fields_ = [("src_off", C.c_long), ("dst_copy_off", C.c_long), ("dst_t_off", C.c_long), ("R", C.c_int), ("Cc", C.c_int),
                ("lds", C.c_int), ("ldc", C.c_int), ("ldt", C.c_int), ("scale", C.c_float)]


i32, i64, f32 = C.c_int, C.c_long, C.c_float
_PROTOS = {
    "pcm_gemm_bf16": [C.POINTER(GemmSeg), i32, C.POINTER(GemmEpi), vp],
    "pcm_lora_wgrad_bf16": [C.POINTER(WgradArgs), vp],
    "pcm_lora_wgrad_multi_bf16": [C.POINTER(WgradArgs), C.c_int, vp],
    "pcm_conv3x3_wgrad_bf16": [vp, vp, vp, i32, i32, i32, i32, i32, f32, vp],
    "pcm_groupnorm_stats": [vp, vp, i32, i32, i32, i32, vp],
    "pcm_groupnorm_stats_acc": [vp, vp, i32, i32, i32, i32, vp],
    "pcm_groupnorm_stats_ws": [vp, vp, i32, i32, i32, i32, vp, C.c_size_t, vp],
    "pcm_groupnorm_bwd_stats_ws": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp, C.c_size_t, vp],
    "pcm_groupnorm_apply": [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp],
    "pcm_groupnorm_apply_chstats": [vp, vp, i32, vp, i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp],
    "pcm_groupnorm_bwd_stats": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp],
    "pcm_groupnorm_bwd_stats_acc": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp],
    "pcm_groupnorm_bwd_apply": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp],
    "pcm_groupnorm_bwd_apply_res": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp],
    "pcm_groupnorm_param_grad": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp],
    "pcm_rowdot_fwd": [vp, vp, vp, vp, i64, i32, vp],
    "pcm_rowdot_bwd": [vp, vp, vp, vp, vp, vp, i64, i32, vp],
    "pcm_rowdot_bwd_ws": [vp, vp, vp, vp, vp, vp, i64, i32, vp, C.c_size_t, vp],
    "pcm_groupnorm_param_grad_ws": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp, C.c_size_t, vp],
    "pcm_mod_grad_ws": [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, C.c_size_t, vp],
    "pcm_hinge_loss_ordered": [vp, vp, i32, f32, vp, vp, vp, f32, i64, vp],
    "pcm_noise_travel": [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "pcm_scale_add_rows": [vp, vp, vp, vp, i32, i32, vp],
    "pcm_hinge_loss": [vp, vp, i32, f32, vp, vp, vp, f32, i64, vp],
    "pcm_layernorm_fwd": [vp, vp, vp, vp, vp, vp, i32, i32, f32, vp],
    "pcm_layernorm_bwd": [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "pcm_geglu_fwd": [vp, vp, i32, i32, vp],
    "pcm_geglu_bwd": [vp, vp, vp, i32, i32, vp],
    "pcm_geglu_bwd_interleaved": [vp, i32, vp, vp, i32, i32, vp],
    "pcm_attn_fwd": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp],
    "pcm_attn_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp],
    "pcm_attn_fwd_prescaled": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "pcm_attn_bwd_prescaled": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "pcm_attn_fwd_ws": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, C.c_size_t, vp],
    "pcm_attn_bwd_ws": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, C.c_size_t, vp],
    "pcm_upsample2x_nhwc": [vp, vp, i32, i32, i32, i32, vp],
    "pcm_pool2x_sum_nhwc": [vp, vp, i32, i32, i32, i32, vp],
    "pcm_concat_channels": [vp, i32, vp, i32, vp, i64, vp],
    "pcm_split_channels": [vp, vp, i32, vp, i32, i64, i32, vp],
    "pcm_add_bf16": [vp, vp, vp, i64, vp],
    "pcm_colsum_bf16": [vp, vp, i32, i32, i32, vp],
    "pcm_colsum_bf16_ws": [vp, vp, i32, i32, i32, vp, C.c_size_t, vp],
    "pcm_silu_bf16": [vp, vp, i64, vp],
    "pcm_silu_bwd_bf16": [vp, vp, vp, i64, vp],
    "pcm_conv_in_fwd": [vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "pcm_conv_in_fwd2": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "pcm_conv_out_fwd": [vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "pcm_conv_out_bwd": [vp, vp, vp, i32, i32, i32, i32, vp],
    "pcm_timestep_embedding": [vp, vp, i32, i32, vp],
    "pcm_add_noise": [vp, vp, vp, vp, vp, i32, i32, vp],
    "pcm_phase_jump": [vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, i32, i32, vp],
    "pcm_sampler_ddim_step": [vp, vp, vp, f32, f32, f32, vp, C.c_long, vp],
    "pcm_layernorm_mod_fwd": [vp, vp, vp, vp, vp, vp, i32, i32, f32, i32, vp],
    "pcm_layernorm_mod_bwd": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "pcm_rowgate_fma": [vp, vp, vp, vp, i32, i32, i32, vp],
    "pcm_gelu_tanh_fwd": [vp, vp, C.c_long, vp],
    "pcm_gelu_tanh_bwd": [vp, vp, vp, C.c_long, vp],
    "pcm_patchify2x2": [vp, vp, i32, i32, i32, i32, i32, vp],
    "pcm_unpatchify2x2": [vp, vp, i32, i32, i32, i32, i32, vp],
    "pcm_mod_grad": [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "pcm_timestep_embedding_f32": [vp, vp, i32, i32, vp],
    "pcm_fm_add_noise": [vp, vp, vp, vp, vp, i32, i32, vp],
    "pcm_fm_phase_jump": [vp, i32, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, i32, i32, vp],
    "pcm_fm_cfg_euler_step": [vp, vp, vp, vp, f32, vp, vp, vp, vp, i32, i32, vp],
    "pcm_fm_noise_travel": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "pcm_fm_sampler_step": [vp, vp, f32, vp, f32, f32, vp, vp, C.c_long, vp],
    "pcm_cfg_ddim_step": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "pcm_consistency_loss": [vp, vp, vp, i32, f32, vp, vp, f32, i32, i32, vp],
    "pcm_consistency_loss_ws": [vp, vp, vp, i32, f32, vp, vp, f32, i32, i32, vp, C.c_size_t, vp],
    "pcm_sumsq_f32": [vp, vp, i64, vp],
    "pcm_sumsq_f32_ws": [vp, vp, i64, vp, C.c_size_t, vp],
    "pcm_adamw_clip_step": [vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, f32, i32, f32, i64, vp, vp, vp],
    "pcm_adamw_clip_step_scaled": [vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, f32, f32, i64, vp, vp, vp, vp],
    "pcm_loss_scale_update": [vp, vp, vp, vp, f32, f32, i32, vp],
    "pcm_scale_f32_dev": [vp, vp, i64, vp],
    "pcm_ema_update": [vp, vp, f32, i64, vp],
    "pcm_ema_update_gated": [vp, vp, f32, i64, vp, vp],
    "pcm_pack_linear": [vp, vp, vp, i32, i32, f32, vp],
    "pcm_pack_conv3x3": [vp, vp, vp, i32, i32, f32, i32, vp],
    "pcm_pack_segmented": [vp, vp, vp, vp, i32, i32, vp],
    "pcm_cast_f32_bf16": [vp, vp, i64, vp],
    "pcm_cast_bf16_f32": [vp, vp, i64, vp],
}


class PcmError(RuntimeError):
    pass


def ptr(t):
    """Raw device/host pointer of a tensor (or None)."""
    if t is None:
        return None
    return vp(t.data_ptr())


class Lib:
    """Loaded C-ABI library with checked calls: ``lib.call('pcm_x', args...)``."""

    def __init__(self, path=DEFAULT_LIB):
        if not os.path.exists(path):
            raise RuntimeError(
                f"pcm_amd: HIP library not found at {path}. Build it with "
                "`python __graft_entry__.py` (or pcm_amd/build.py). There is no fallback path.")
        self.path = path
        self.dll = C.CDLL(path)
        self.dll.pcm_last_error.restype = C.c_char_p
        self.dll.pcm_abi_version.restype = C.c_int
        self.dll.pcm_act_dtype.restype = C.c_int
        self.act_dtype = int(self.dll.pcm_act_dtype())       # 0: bfloat16 build, 1: IEEE-half build (include/pcm_hip.h PCM_FMT_*)
        # identity of the sources the library was built from (abi 5).  One of THIS tree's own libraries (pcm_amd/lib/) must carry this tree's
        # source id: a snapshot whose binaries are older than its sources fails here instead of running stale kernels (build() rebuilds by
        # content hash, so this only fires when build() was skipped).  Other paths (emulator builds, A/B copies under tools/) are not checked.
        self.build_id = "unknown"
        if hasattr(self.dll, "pcm_build_id"):
            self.dll.pcm_build_id.restype = C.c_char_p
            self.build_id = self.dll.pcm_build_id().decode()
        if os.path.dirname(os.path.abspath(path)) == os.path.join(HERE, "lib") and os.environ.get("PCM_ALLOW_STALE_LIB") != "1":
            from . import build as _build
            want = _build.source_id()
            if not self.build_id.startswith(want + "-"):
                raise RuntimeError(f"pcm_amd: {path} was built from other sources (library {self.build_id}, tree {want}): "
                                   "run `python __graft_entry__.py build` (PCM_ALLOW_STALE_LIB=1 loads it anyway)")
        self.dll.pcm_gemm_workspace_bytes.restype = C.c_size_t
        self.dll.pcm_gemm_workspace_bytes.argtypes = [C.POINTER(GemmSeg), C.c_int, C.POINTER(GemmEpi)]
        self.dll.pcm_gemm_plan_code.restype = C.c_int
        self.dll.pcm_gemm_plan_code.argtypes = [C.POINTER(GemmSeg), C.c_int, C.POINTER(GemmEpi)]
        self.dll.pcm_gemm_emits_chstats.restype = C.c_int
        self.dll.pcm_gemm_emits_chstats.argtypes = [C.POINTER(GemmSeg), C.c_int, C.POINTER(GemmEpi)]
        self.dll.pcm_attn_workspace_bytes.restype = C.c_size_t
        self.dll.pcm_attn_workspace_bytes.argtypes = [C.c_int] * 6
        self.dll.pcm_groupnorm_workspace_bytes.restype = C.c_size_t
        self.dll.pcm_groupnorm_workspace_bytes.argtypes = [C.c_int] * 4
        self.dll.pcm_lora_wgrad_workspace_bytes.restype = C.c_size_t
        self.dll.pcm_lora_wgrad_workspace_bytes.argtypes = [C.POINTER(WgradArgs)]
        self.dll.pcm_colsum_workspace_bytes.restype = C.c_size_t
        self.dll.pcm_colsum_workspace_bytes.argtypes = [C.c_int] * 3
        for name, argt in (("pcm_rowdot_bwd_workspace_bytes", [C.c_long, C.c_int]), ("pcm_groupnorm_param_grad_workspace_bytes", [C.c_int] * 4),
                           ("pcm_mod_grad_workspace_bytes", [C.c_int] * 3)):        # abi 5
            f = getattr(self.dll, name, None)
            if f is not None:
                f.restype, f.argtypes = C.c_size_t, argt
        self.fn = {}
        for name, argt in _PROTOS.items():
            f = getattr(self.dll, name, None)
            if f is None:
                continue  # checked by tests/test_capi_symbols.py; calling a missing symbol raises below
            f.argtypes = argt
            f.restype = C.c_int
            self.fn[name] = f

    def call(self, name, *args):
        f = self.fn.get(name)
        if f is None:
            raise PcmError(f"{name}: symbol not exported by {self.path}")
        rc = f(*args)
        if rc != 0:
            raise PcmError(f"{name} failed (rc={rc}): {self.dll.pcm_last_error().decode()}")

    @staticmethod
    def stream():
        """torch's current HIP stream as a void* (NULL on CPU tensors / host-emulation tests)."""
        if torch.cuda.is_available():
            return vp(torch.cuda.current_stream().cuda_stream)
        return vp(0)


_LIB = None
_LIB_PATH = DEFAULT_LIB      # what lib() loads: precision.set_precision("fp16") points it at F16_LIB


def lib():
    global _LIB
    if _LIB is None:
        _LIB = Lib(_LIB_PATH)
    return _LIB


def set_lib(l):
    """Tests only: point the op layer at another build of the same C ABI (tests/emu)."""
    global _LIB
    _LIB = l


_TOOLS = {}


def tools_lib(f16=False):
    """the TOOLS build (pcm_debug_* hooks, A/B variants) of the library for tools/ and hook-using tests: ``capi.set_lib(capi.tools_lib())``"""
    if f16 not in _TOOLS:
        _TOOLS[f16] = Lib(TOOLS_F16_LIB if f16 else TOOLS_LIB)
    return _TOOLS[f16]
