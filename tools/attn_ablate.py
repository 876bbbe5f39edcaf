"""Timing ablations of attn_fwd_kernel (PCM_ABLATE build): which part of a key tile costs what at the SD1.5 level-0 shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import ops, capi
capi.set_lib(capi.tools_lib())      # the TOOLS build of the library: the pcm_debug_* hooks used below are not in the product build
capi.set_lib(capi.Lib(os.path.join(ROOT, "tools", "probes", "libpcm_ablate.so")))
dll = capi.lib().dll
def bench(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for (B, L, Lk, d, H) in [(32, 4096, 4096, 40, 8), (32, 1024, 1024, 80, 8), (4, 4096, 4096, 64, 10)]:
    qkv = torch.randn(B, L, 3 * H * d, device="cuda").bfloat16()
    q, k, v = qkv[:, :, :H * d], qkv[:, :, H * d:2 * H * d], qkv[:, :, 2 * H * d:]
    fl = 4.0 * B * H * L * Lk * d
    row = []
    for m in (0, 1, 2, 4, 8, 16, 3, 12, 15, 31):
        dll.pcm_debug_attn_ablate(m)
        t = bench(lambda: ops.attn_fwd(q, k, v, H, d))
        row.append("m%-2d %7.1f us" % (m, t * 1e3))
    dll.pcm_debug_attn_ablate(0)
    print("B=%d L=%d d=%d H=%d (%.0f TF/s at m0 rate basis %.3f TFLOP): %s" % (B, L, d, H, 0, fl / 1e12, " | ".join(row)), flush=True)
