"""A/B two builds of the kernel library on the rank-64 LoRA-down projections (N = 64) of the four configs.
usage: n64_ab_libs.py <libA.so> <libB.so>   (3 operand sets rotated, interleaved rounds, median)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch  # noqa: E402
from pcm_amd import capi, ops  # noqa: E402

libs = [capi.Lib(os.path.abspath(p)) for p in sys.argv[1:3]]
shapes = [(8192, 6144), (8192, 5120), (4096, 10240), (16384, 5120), (16384, 2560), (8192, 1280), (4096, 1280), (2048, 1280), (8192, 1536), (16384, 1536), (16384, 640),
          (8192, 640), (32768, 640), (2464, 768), (616, 768), (616, 2048), (4096, 3840), (8192, 3840), (2048, 5120), (4096, 5120)]
ROUNDS, REP = 5, 8


def timed(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(REP):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / REP * 1e3


tot = [0.0, 0.0]
for (M, K) in shapes:
    sets = [(torch.randn(M, K, device="cuda").bfloat16(), (torch.randn(64, K, device="cuda") * 0.05).bfloat16()) for _ in range(3)]
    outs = [torch.empty(M, 64, device="cuda", dtype=torch.bfloat16) for _ in libs]
    res = [[], []]
    for r in range(ROUNDS):
        for li, lib in enumerate(libs):
            capi.set_lib(lib)
            res[li].append(timed(lambda i: ops.gemm([ops.Seg(sets[i % 3][0], sets[i % 3][1])], M, 64, outs[li])))
    for li, lib in enumerate(libs):
        capi.set_lib(lib)
        ops.gemm([ops.Seg(sets[0][0], sets[0][1])], M, 64, outs[li])
    torch.cuda.synchronize()
    t = [sorted(v)[len(v) // 2] for v in res]
    nb = M * K * 2 + M * 128
    tot[0] += t[0]; tot[1] += t[1]
    print("(%6d, 64, %5d)  A %6.1f us %5.2f TB/s | B %6.1f us %5.2f TB/s (x%.3f)  max|A-B| %.3g" % (M, K, t[0], nb / t[0] / 1e6, t[1], nb / t[1] / 1e6, t[1] / t[0],
          float((outs[0].float() - outs[1].float()).abs().max())), flush=True)
capi.set_lib(None)
print("sum: A %.1f us | B %.1f us (x%.3f)" % (tot[0], tot[1], tot[1] / tot[0]))
