"""Tensor-level wrappers over the C ABI (one HIP implementation per op; no fallbacks).
All activations are channels-last bf16 ``[B, H*W, C]`` / ``[M, C]`` torch tensors used purely as
device-memory handles."""
import ctypes as C

import torch

from . import capi
from .capi import GemmEpi, GemmSeg, WgradArgs, ptr

BF16 = torch.bfloat16


def _chk(t, dtype=None):
    assert t.is_contiguous(), "pcm_amd.ops: tensor must be contiguous"
    if dtype is not None:
        assert t.dtype == dtype, f"expected {dtype}, got {t.dtype}"
    return t


class Seg:
    """One K-segment of pcm_gemm_bf16."""

    def __init__(self, a, w, conv=None, lda=None):
        """plain: a [M, K] (row stride lda), w [N, K].  conv: a NHWC [B,Hs,Ws,C], w [N, 9*C],
        conv = dict(Hs, Ws, stride=1, src_mode=SRC_DIRECT)."""
        self.a, self.w, self.conv, self.lda = a, w, conv, lda

    def fill(self, s: GemmSeg):
        s.a, s.w = ptr(self.a), ptr(self.w)
        s.K = self.w.shape[-1]
        if self.conv is None:
            s.mode = capi.SEG_PLAIN
            s.lda = self.lda if self.lda is not None else self.a.shape[-1]
            s.Hs = s.Ws = s.C = 0
            s.stride, s.src_mode = 1, 0
        else:
            s.mode = capi.SEG_CONV3X3
            s.lda = 0
            s.Hs, s.Ws = self.conv["Hs"], self.conv["Ws"]
            s.C = self.a.shape[-1]
            s.stride = self.conv.get("stride", 1)
            s.src_mode = self.conv.get("src_mode", capi.SRC_DIRECT)


def gemm(segs, M, N, out, bias=None, rowvec=None, rows_per_batch=0, residual=None, act=capi.ACT_NONE,
         alpha=1.0, Ho=0, Wo=0, ldo=None, ldr=None):
    arr = (GemmSeg * len(segs))()
    for i, s in enumerate(segs):
        s.fill(arr[i])
    e = GemmEpi()
    e.M, e.N, e.Ho, e.Wo = M, N, Ho, Wo
    e.bias, e.rowvec, e.rows_per_batch = ptr(bias), ptr(rowvec), rows_per_batch
    e.residual = ptr(residual)
    e.ldr = (ldr if ldr is not None else (residual.shape[-1] if residual is not None else 0))
    e.out = ptr(out)
    e.ldo = ldo if ldo is not None else out.shape[-1]
    e.out_dtype = capi.PCM_F32 if out.dtype == torch.float32 else capi.PCM_BF16
    e.act, e.alpha = act, alpha
    capi.lib().call("pcm_gemm_bf16", arr, len(segs), C.byref(e), capi.Lib.stream())
    return out
