import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import ops, capi
from pcm_amd.capi import ptr
def bench(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
L = capi.lib()
for (B, HW, C) in [(16, 4096, 320), (32, 4096, 320), (16, 4096, 960), (16, 1024, 640), (16, 256, 1280), (16, 64, 2560)]:
    x = torch.randn(B, HW, C, device="cuda").bfloat16(); g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    stats = torch.empty(B, 32, 2, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    st = capi.Lib.stream()
    t1 = bench(lambda: L.call("pcm_groupnorm_stats", ptr(x), ptr(stats), B, HW, C, 32, st))
    t2 = bench(lambda: L.call("pcm_groupnorm_apply", ptr(x), ptr(stats), ptr(g), ptr(b), ptr(y), B, HW, C, 32, 1e-5, 1, st))
    mb = x.numel() * 2 / 1e6
    print("B=%2d HW=%4d C=%4d (%.0f MB): stats %7.1f us %6.2f TB/s | apply %7.1f us %6.2f TB/s" % (B, HW, C, mb, t1*1e3, mb/t1/1e3, t2*1e3, 2*mb/t2/1e3))
