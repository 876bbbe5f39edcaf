"""A/B two builds of the kernel library on the attention shapes of the step (SD1.5 d = 40 / 80 / 160, SDXL / SD3 d = 64).
usage: attn_ab_libs.py <libA.so> <libB.so>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import ops, capi
libs = [capi.Lib(os.path.abspath(p)) for p in sys.argv[1:3]]
def bench(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for (B, L, Lk, d, H) in [(32, 4096, 4096, 40, 8), (16, 4096, 4096, 40, 8), (32, 1024, 1024, 80, 8), (32, 4096, 77, 40, 8), (32, 256, 256, 160, 8), (4, 4096, 4096, 64, 10), (2, 4250, 4250, 64, 24)]:
    q = torch.randn(B, L, H * d, device="cuda").bfloat16(); k = torch.randn(B, Lk, H * d, device="cuda").bfloat16()
    v = torch.randn(B, Lk, H * d, device="cuda").bfloat16(); do = torch.randn(B, L, H * d, device="cuda").bfloat16()
    fl = 4.0 * B * H * L * Lk * d
    # A and B are timed INTERLEAVED (A, B, A, B, ...; min of the rounds): timed one after the other the library measured second
    # came out 3-13 % faster on the long forward shapes whichever library it was (clock ramp after the operand setup)
    res, outs = [[1e9, 1e9], [1e9, 1e9]], []
    for l in libs:
        capi.set_lib(l)
        o, lse = ops.attn_fwd(q, k, v, H, d)
        g = ops.attn_bwd(q, k, v, o, do, lse, H, d)
        outs.append((o.float(), [x.float() for x in g], o, lse))
    for rnd_ in range(4):
        for i, l in enumerate(libs):
            capi.set_lib(l)
            o, lse = outs[i][2], outs[i][3]
            res[i][0] = min(res[i][0], bench(lambda: ops.attn_fwd(q, k, v, H, d)))
            res[i][1] = min(res[i][1], bench(lambda: ops.attn_bwd(q, k, v, o, do, lse, H, d)))
    def rel(a, b): return float((a - b).norm() / (b.norm() + 1e-30))
    print("B=%2d H=%2d L=%4d Lk=%4d d=%3d | fwd A %7.3f ms %6.0f TF/s  B %7.3f ms %6.0f TF/s (x%.3f) | bwd A %7.3f ms %6.0f  B %7.3f ms %6.0f (x%.3f) | rel diff o %.1e dq %.1e dk %.1e dv %.1e" % (
        B, H, L, Lk, d, res[0][0], fl / res[0][0] / 1e9, res[1][0], fl / res[1][0] / 1e9, res[1][0] / res[0][0],
        res[0][1], 2.5 * fl / res[0][1] / 1e9, res[1][1], 2.5 * fl / res[1][1] / 1e9, res[1][1] / res[0][1],
        rel(outs[1][0], outs[0][0]), rel(outs[1][1][0], outs[0][1][0]), rel(outs[1][1][1], outs[0][1][1]), rel(outs[1][1][2], outs[0][1][2])), flush=True)
