"""Where does the time of a gemm8p launch go?  Runs shapes of the bs-16 step under the PCM_ABLATE build with parts of the kernel
switched off (results are garbage by construction, only the timing is read).  masks: 1 no global stores, 2 no epilogue, 4 no MFMA,
8 no LDS-DMA after the prologue."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import ops, capi
capi.set_lib(capi.tools_lib())      # the TOOLS build of the library: the pcm_debug_* hooks used below are not in the product build
capi.set_lib(capi.Lib(os.path.join(ROOT, "tools", "probes", "libpcm_ablate.so")))
dll = capi.lib().dll
def bench(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
shapes = [(131072, 2560, (320, 64), "lin", 0), (131072, 320, (320, 64), "lin", 0), (32768, 5120, (640, 64), "lin", 0), (32768, 640, (640, 64), "lin", 0),
          (8192, 1280, (1280, 64), "lin", 0), (131072, 320, (2880, 64), "conv", 64), (32768, 1280, (11520, 64), "conv", 32)]
dll.pcm_debug_gemm_big_mode(2)
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
    row = []
    for mask in (0, 1, 2, 4, 8, 12, 2 | 4, 2 | 8, 2 | 4 | 8):
        dll.pcm_debug_gemm_ablate(mask)
        ms = bench(lambda: ops.gemm(segs, M, N, out, Ho=Hs, Wo=Hs) if kind == "conv" else ops.gemm(segs, M, N, out))
        row.append("m%-2d %7.1f" % (mask, ms * 1e3))
    dll.pcm_debug_gemm_ablate(0)
    tiles = ((M + 255) // 256) * ((N + 319) // 320)
    print("%-36s tiles %5d (%.1f rounds) us: %s" % (str((M, N, Ks, kind)), tiles, tiles / 256, " | ".join(row)), flush=True)
