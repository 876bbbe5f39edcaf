import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import ops, capi
capi.set_lib(capi.tools_lib())      # the TOOLS build of the library: the pcm_debug_* hooks used below are not in the product build
def bench(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
B = 16
CFGS = [(512, 4), (1024, 2), (2048, 2), (4096, 1)]
def sweep(fn):
    out = []
    for tgt, mc in CFGS:
        capi.lib().dll.pcm_debug_wgrad_grid(tgt, mc)
        out.append("%d/%d %.1f" % (tgt, mc, bench(fn) * 1e3))
    capi.lib().dll.pcm_debug_wgrad_grid(0, 0)
    return "  ".join(out)
for (M, N) in [(B*4096, 320), (B*4096, 2560), (B*1024, 640), (B*256, 1280), (B*256, 10240), (B*77, 320), (16, 1280)]:
    dy = torch.randn(M, N, device="cuda").bfloat16(); t = torch.randn(M, 64, device="cuda").bfloat16()
    out = torch.zeros(N, 64, device="cuda")
    ms = bench(lambda: ops.lora_wgrad(dy, t, out, 0.125, M, g_stride=64, r_stride=1))
    print("plain M=%6d G=%5d  %8.3f ms  %7.1f GB/s(big once) | us: %s" % (M, N, ms, M*N*2/ms/1e6, sweep(lambda: ops.lora_wgrad(dy, t, out, 0.125, M, g_stride=64, r_stride=1))))
for (H, C) in [(64, 320), (32, 640), (16, 1280), (8, 1280), (16, 2560), (64, 960)]:
    x = torch.randn(B, H*H, C, device="cuda").bfloat16(); M = B*H*H
    u = torch.randn(M, 64, device="cuda").bfloat16(); out = torch.zeros(64, 3, 3, C, device="cuda")
    ms = bench(lambda: ops.lora_wgrad(x, u, out, 1.0, M, conv=dict(Hs=H, Ws=H, Ho=H, Wo=H), g_stride=1, r_stride=9 * C))
    print("conv  H=%3d C=%5d  %8.3f ms  %7.1f TF/s | us: %s" % (H, C, ms, 2.0*M*9*C*64/ms/1e9, sweep(lambda: ops.lora_wgrad(x, u, out, 1.0, M, conv=dict(Hs=H, Ws=H, Ho=H, Wo=H), g_stride=1, r_stride=9 * C))))
