"""A/B of the two K orders of the stride-1 3x3 implicit GEMM in gemm8p (tap-outer vs chunk-outer) on the SD1.5 bs-16 step's conv shapes;
inputs are evicted from L2 / Infinity Cache between timed launches (a 512 MB touch), as in the step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import ops, capi
capi.set_lib(capi.tools_lib())      # the TOOLS build of the library: the pcm_debug_* hooks used below are not in the product build
dll = capi.lib().dll
flush = torch.zeros(128 * 1024 * 1024, device="cuda")
def bench(fn, n=6, cold=True):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        if cold: flush.add_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    ts.sort(); return ts[len(ts) // 2]
shapes = [(131072, 320, 320, 64, True), (131072, 320, 640, 64, True), (131072, 320, 960, 64, True), (131072, 640, 640, 64, True), (65536, 320, 320, 64, True),
          (32768, 640, 640, 32, True), (32768, 640, 1920, 32, True), (32768, 1280, 1280, 32, True), (8192, 1280, 1280, 16, True), (8192, 1280, 2560, 16, True),
          (2048, 1280, 1280, 8, True), (16384, 640, 640, 32, True)]
for (M, N, Ci, Hs, lora) in shapes:
    B = M // (Hs * Hs)
    x = torch.randn(B, Hs, Hs, Ci, device="cuda").bfloat16(); w = (torch.randn(N, 9 * Ci, device="cuda") * 0.02).bfloat16()
    segs = [ops.Seg(x, w, conv=dict(Hs=Hs, Ws=Hs))]
    if lora:
        t = torch.randn(M, 64, device="cuda").bfloat16(); bl = (torch.randn(N, 64, device="cuda") * 0.05).bfloat16()
        segs.append(ops.Seg(t, bl))
    out0 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); out1 = torch.empty_like(out0)
    fl = 2.0 * M * N * (9 * Ci + (64 if lora else 0))
    res = []
    for (md, co), o in (((0, 0), out0), ((1, 0), out1), ((1, 1), out1)):
        dll.pcm_debug_gemm_conv_md(md); dll.pcm_debug_gemm_conv_order(co)
        f = lambda: ops.gemm(segs, M, N, o, Ho=Hs, Wo=Hs)
        res.append((bench(f, cold=True), bench(f, cold=False)))
    dll.pcm_debug_gemm_conv_md(-1); dll.pcm_debug_gemm_conv_order(-1)
    print("M=%6d N=%4d Ci=%4d %dx%d | re-key tap-outer %7.1f us %6.0f TF/s (warm %7.1f) | mask+delta tap-outer %7.1f us %6.0f TF/s (warm %7.1f) x%.3f | chunk-outer %7.1f us %6.0f TF/s x%.3f" % (
        M, N, Ci, Hs, Hs, res[0][0] * 1e3, fl / res[0][0] / 1e9, res[0][1] * 1e3, res[1][0] * 1e3, fl / res[1][0] / 1e9, res[1][1] * 1e3, res[0][0] / res[1][0],
        res[2][0] * 1e3, fl / res[2][0] / 1e9, res[0][0] / res[2][0]), flush=True)
