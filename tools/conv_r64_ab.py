"""conv LoRA down-projection t = conv3x3(x, A) (N = 64) on the shapes of the bs-16 step: halo-window kernel (conv_r64.hip) against the
generic implicit-GEMM tile (pcm_debug_conv_r64(0)); interleaved timing, min of 3 rounds of 10 launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import ops, capi
capi.set_lib(capi.tools_lib())      # the TOOLS build of the library: the pcm_debug_* hooks used below are not in the product build
dll = capi.lib().dll
def bench(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (B, H, C) in [(32, 64, 320), (32, 64, 640), (32, 64, 960), (32, 32, 640), (32, 32, 1280), (32, 32, 1920), (32, 16, 1280), (32, 16, 2560), (16, 64, 320), (16, 32, 640)]:
    x = torch.randn(B, H, H, C, device="cuda").bfloat16(); w = (torch.randn(64, 9 * C, device="cuda") * 0.02).bfloat16()
    M = B * H * H
    out = torch.empty(M, 64, device="cuda", dtype=torch.bfloat16)
    f = lambda: ops.gemm([ops.Seg(x, w, conv=dict(Hs=H, Ws=H))], M, 64, out, Ho=H, Wo=H)
    t = [1e9, 1e9]
    for _ in range(3):
        for mode in (0, 1):
            dll.pcm_debug_conv_r64(mode)
            t[mode] = min(t[mode], bench(f))
    dll.pcm_debug_conv_r64(1)
    fl = 2.0 * M * 64 * 9 * C
    print("B=%2d %3dx%-3d C=%4d: generic %7.1f us %6.0f TF/s | halo kernel %7.1f us %6.0f TF/s (x%.2f), x read %.0f MB" % (B, H, H, C, t[0], fl / t[0] / 1e6, t[1], fl / t[1] / 1e6, t[0] / t[1], M * C * 2 / 1e6), flush=True)
