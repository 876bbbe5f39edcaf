"""A/B of the XCD-aware block map of the first forward-attention kernel (pcm_debug_attn_xcd_remap) on the SD1.5 level-0 / level-1 shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import ops, capi
capi.set_lib(capi.tools_lib())      # the TOOLS build of the library: the pcm_debug_* hooks used below are not in the product build
dll = capi.lib().dll
dll.pcm_debug_attn_fwd_variant(0)
def bench(fn, n=8):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for (B, L, Lk, d, H) in [(32, 4096, 4096, 40, 8), (16, 4096, 4096, 40, 8), (32, 1024, 1024, 80, 8), (32, 4096, 77, 40, 8), (4, 4096, 4096, 64, 10)]:
    q = torch.randn(B, L, H * d, device="cuda").bfloat16(); k = torch.randn(B, Lk, H * d, device="cuda").bfloat16()
    v = torch.randn(B, Lk, H * d, device="cuda").bfloat16()
    fl = 4.0 * B * H * L * Lk * d
    best, outs = [1e9, 1e9], []
    for on in (0, 1):
        dll.pcm_debug_attn_xcd_remap(on)
        outs.append(ops.attn_fwd(q, k, v, H, d)[0].float())
    for _ in range(4):
        for on in (0, 1):
            dll.pcm_debug_attn_xcd_remap(on)
            best[on] = min(best[on], bench(lambda: ops.attn_fwd(q, k, v, H, d)))
    print("B=%2d H=%2d L=%4d Lk=%4d d=%3d | plain map %7.3f ms %5.0f TF/s | XCD-aware %7.3f ms %5.0f TF/s (x%.3f) | max diff %.1e" % (
        B, H, L, Lk, d, best[0], fl / best[0] / 1e9, best[1], fl / best[1] / 1e9, best[0] / best[1], float((outs[1] - outs[0]).abs().max())), flush=True)
dll.pcm_debug_attn_xcd_remap(0); dll.pcm_debug_attn_fwd_variant(-1)
