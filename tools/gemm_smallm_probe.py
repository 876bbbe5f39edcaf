"""Times of the batch-row projections (M <= 16) of the four configs through pcm_gemm_bf16; run once with PCM_GEMM_SMALLM=0 (the generic
tiles / rank-64 kernel they used before) and once with the default (gemm_smallm.hip) and compare.  3 operand sets rotated, median of 5 rounds."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch  # noqa: E402
from pcm_amd import capi, ops  # noqa: E402
capi.set_lib(capi.tools_lib())      # the TOOLS build of the library: the pcm_debug_* hooks used below are not in the product build

shapes = [(32, 1280, (320,), 1), (32, 1280, (1280,), 0), (32, 1280, (1280, 64), 0), (32, 640, (1280, 64), 0), (32, 320, (1280, 64), 0), (32, 64, (1280,), 0), (16, 64, (1280,), 0),
          (16, 1280, (1280,), 0), (8, 1280, (2816,), 1), (8, 1280, (1280, 64), 0), (2, 9216, (1536,), 0), (4, 9216, (1536,), 0), (2, 3072, (1536,), 0), (4, 1536, (1536,), 1)]
dll = capi.lib().dll
tot = 0.0
for (M, N, Ks, act) in shapes:
    sets = []
    for _ in range(3):
        segs = [ops.Seg(torch.randn(M, K, device="cuda").bfloat16(), (torch.randn(N, K, device="cuda") * 0.05).bfloat16()) for K in Ks]
        sets.append((segs, torch.empty(M, N, device="cuda", dtype=torch.bfloat16)))
    bias = torch.randn(N, device="cuda")
    call = lambda i: ops.gemm(sets[i % 3][0], M, N, sets[i % 3][1], bias=bias, act=capi.ACT_SILU if act else capi.ACT_NONE)   # noqa: E731
    call(0)
    plan = dll.pcm_debug_last_gemm_plan()
    res = []
    for r in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(12):
            call(i)
        e.record()
        torch.cuda.synchronize()
        res.append(s.elapsed_time(e) / 12 * 1e3)
    t = sorted(res)[2]
    tot += t
    nb = sum(N * K * 2 + M * K * 2 for K in Ks) + M * N * 2
    print("(%3d, %5d, %-12s) plan %5d  %6.1f us  %5.2f TB/s" % (M, N, Ks, plan, t, nb / t / 1e6), flush=True)
print("sum %.1f us" % tot)
