"""A/B of the forward attention variants of ONE library build on the step's shapes: pcm_debug_attn_fwd_variant 0 = the dependent-chain
kernel of attention.hip, 1 = software-pipelined (attention_fwd.hip) with VGPR accumulators, 2 = pipelined / AccVGPR / 2 waves per SIMD,
3 = pipelined / AccVGPR / 1 wave per SIMD.  Timed interleaved (min over rounds); outputs compared with variant 0.
usage: attn_fwd_variants.py [variants, default 0,1,2,3]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import ops, capi
capi.set_lib(capi.tools_lib())      # the TOOLS build of the library: the pcm_debug_* hooks used below are not in the product build
variants = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3".split(","))]
dll = capi.lib().dll
def bench(fn, n=6):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for (B, L, Lk, d, H) in [(32, 4096, 4096, 40, 8), (16, 4096, 4096, 40, 8), (32, 1024, 1024, 80, 8), (32, 4096, 77, 40, 8), (32, 1024, 77, 80, 8),
                         (4, 4096, 4096, 64, 10), (4, 1024, 1024, 64, 20), (2, 4250, 4250, 64, 24)]:
    q = torch.randn(B, L, H * d, device="cuda").bfloat16(); k = torch.randn(B, Lk, H * d, device="cuda").bfloat16()
    v = torch.randn(B, Lk, H * d, device="cuda").bfloat16()
    fl = 4.0 * B * H * L * Lk * d
    best, outs = {}, {}
    for var in variants:
        dll.pcm_debug_attn_fwd_variant(var)
        o, lse = ops.attn_fwd(q, k, v, H, d)
        outs[var] = (o.float(), lse.clone())
        best[var] = 1e9
    for _ in range(4):
        for var in variants:
            dll.pcm_debug_attn_fwd_variant(var)
            best[var] = min(best[var], bench(lambda: ops.attn_fwd(q, k, v, H, d)))
    ref = outs[variants[0]]
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
    print("B=%2d H=%2d L=%4d Lk=%4d d=%3d | " % (B, H, L, Lk, d) + " | ".join(
        "v%d %7.3f ms %5.0f TF/s (x%.2f) do %.0e dlse %.0e" % (var, best[var], fl / best[var] / 1e9, best[variants[0]] / best[var], rel(outs[var][0], ref[0]),
                                                             float((outs[var][1] - ref[1]).abs().max())) for var in variants), flush=True)
dll.pcm_debug_attn_fwd_variant(-1)
