"""How many attention workgroups does a CU really run side by side?  fwd at d=40, L=4096, H=8 with B = 1..8 -> 256*B workgroups on 256 CUs:
the time stays flat while the extra workgroups fit next to the first one and steps up when a new round starts."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import ops, capi
capi.lib()
def bench(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for d, H, L in ((40, 8, 4096), (64, 8, 4096)):
    for B in (1, 2, 3, 4, 5, 6, 8, 16, 32):
        qkv = torch.randn(B, L, 3 * H * d, device="cuda").bfloat16()
        q, k, v = qkv[:, :, :H * d], qkv[:, :, H * d:2 * H * d], qkv[:, :, 2 * H * d:]
        t = bench(lambda: ops.attn_fwd(q, k, v, H, d))
        fl = 4.0 * B * H * L * L * d
        print("d=%d B=%2d: %5d workgroups  %8.1f us  %6.1f TFLOP/s  (%.1f us per 256 workgroups)" % (d, B, B * H * L // 128, t, fl / t / 1e6, t / B), flush=True)
