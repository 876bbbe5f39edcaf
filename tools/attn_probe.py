"""GPU throughput probe of the attention kernels at the SD1.5 shapes (not a test)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import ops

def bench(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

for (B, L, Lk, d) in [(16, 4096, 4096, 40), (32, 4096, 4096, 40), (16, 1024, 1024, 80), (16, 4096, 77, 40), (16, 256, 256, 160)]:
    H = 8
    q = torch.randn(B, L, H * d, device="cuda").bfloat16(); k = torch.randn(B, Lk, H * d, device="cuda").bfloat16()
    v = torch.randn(B, Lk, H * d, device="cuda").bfloat16(); do = torch.randn(B, L, H * d, device="cuda").bfloat16()
    o, lse = ops.attn_fwd(q, k, v, H, d)
    fl = 4.0 * B * H * L * Lk * d
    tf = bench(lambda: ops.attn_fwd(q, k, v, H, d))
    tb = bench(lambda: ops.attn_bwd(q, k, v, o, do, lse, H, d))
    print("B=%2d L=%4d Lk=%4d d=%3d  fwd %8.3f ms %7.1f TF/s   bwd %8.3f ms %7.1f TF/s (2.5x fwd flops)" % (B, L, Lk, d, tf, fl / tf / 1e9, tb, 2.5 * fl / tb / 1e9))
