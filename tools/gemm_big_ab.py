"""A/B of the 4-wave GEMM tiles (PCM_GEMM_BIG mode 0) against the 256-row phased kernel (mode 2) on the SD1.5 step's
contraction shapes (tuning tool; prints one line per shape)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import ops, capi
capi.set_lib(capi.tools_lib())      # the TOOLS build of the library: the pcm_debug_* hooks used below are not in the product build
dll = capi.lib().dll
def bench(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
# (M, N, Ks, kind, Hs) from the bs16 step table (gpurun_out/gemm_table3.txt)
shapes = [(65536, 320, (320, 64), "lin", 0), (131072, 320, (320,), "lin", 0), (16384, 640, (640, 64), "lin", 0), (4096, 1280, (1280, 64), "lin", 0),
          (65536, 2560, (320, 64), "lin", 0), (65536, 320, (1280, 64), "lin", 0), (16384, 5120, (640, 64), "lin", 0), (4096, 10240, (1280, 64), "lin", 0),
          (32768, 640, (640,), "lin", 0), (8192, 1280, (1280,), "lin", 0), (4096, 1280, (5120, 64), "lin", 0), (16384, 640, (2560, 64), "lin", 0),
          (65536, 320, (2880, 64), "conv", 64), (131072, 320, (2880,), "conv", 64), (16384, 640, (5760, 64), "conv", 32), (4096, 1280, (11520, 64), "conv", 16),
          (8192, 1280, (11520,), "conv", 16), (1024, 1280, (11520, 64), "conv", 8), (65536, 320, (5760, 64), "conv", 64), (32768, 640, (5760,), "conv", 32),
          (4096, 1280, (23040, 64), "conv", 16), (65536, 320, (8640, 64), "conv", 64), (65536, 640, (5760, 64), "conv", 64), (16384, 1280, (11520, 64), "conv", 32),
          (8192, 8192, (8192,), "lin", 0)]
if os.environ.get("AB_SET") == "lin2b":   # the short-K Linear shapes of the 2B-sample forward passes (round-2 table, gpurun_out/b/gemm_table.txt)
    shapes = [(131072, 2560, (320, 64), "lin", 0), (131072, 2560, (320,), "lin", 0), (131072, 320, (320, 64), "lin", 0), (131072, 320, (320,), "lin", 0),
              (32768, 5120, (640, 64), "lin", 0), (8192, 1280, (1280, 64), "lin", 0), (32768, 640, (640, 64), "lin", 0), (8192, 10240, (1280, 64), "lin", 0),
              (131072, 960, (320, 192), "lin", 0), (16384, 640, (640, 64), "lin", 0), (131072, 320, (1280, 64), "lin", 0), (65536, 320, (320, 64), "lin", 0),
              (4096, 1280, (1280, 64), "lin", 0), (8192, 1280, (5120, 64), "lin", 0), (32768, 640, (2560, 64), "lin", 0), (32768, 1920, (640, 192), "lin", 0),
              (65536, 1280, (320, 64), "lin", 0), (65536, 320, (2560, 64), "lin", 0), (8192, 3840, (1280, 192), "lin", 0), (16384, 2560, (640, 64), "lin", 0)]
only = os.environ.get("AB_ONLY")
for (M, N, Ks, kind, Hs) in shapes:
    segs = []
    if kind == "conv":
        Ci = Ks[0] // 9; B = M // (Hs * Hs)
        x = torch.randn(B, Hs, Hs, Ci, device="cuda").bfloat16(); w = (torch.randn(N, Ks[0], device="cuda") * 0.02).bfloat16()
        segs.append(ops.Seg(x, w, conv=dict(Hs=Hs, Ws=Hs)))
    else:
        x = torch.randn(M, Ks[0], device="cuda").bfloat16(); w = (torch.randn(N, Ks[0], device="cuda") * 0.05).bfloat16()
        segs.append(ops.Seg(x, w))
    if len(Ks) > 1:
        t = torch.randn(M, Ks[1], device="cuda").bfloat16(); bl = (torch.randn(N, Ks[1], device="cuda") * 0.05).bfloat16()
        segs.append(ops.Seg(t, bl))
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * sum(Ks)
    res = []
    for mode in (0, 2):
        dll.pcm_debug_gemm_big_mode(mode)
        ms = bench(lambda: ops.gemm(segs, M, N, out, Ho=Hs, Wo=Hs) if kind == "conv" else ops.gemm(segs, M, N, out))
        res.append((ms, fl / ms / 1e9, dll.pcm_debug_last_gemm_plan()))
    dll.pcm_debug_gemm_big_mode(1)
    print("%-40s small %8.1f us %7.1f TF/s | big(plan %d) %8.1f us %7.1f TF/s | x%.2f" % (str((M, N, Ks, kind)), res[0][0] * 1e3, res[0][1], res[1][2], res[1][0] * 1e3, res[1][1], res[0][0] / res[1][0]), flush=True)
