"""A/B of the short-K kernel (gemm4w.hip, two workgroups per CU; pcm_debug_gemm_big_mode 3) against the 256-row phased tile (gemm8p.hip,
mode 2) and the planner's own choice (mode 1) on the plain-segment launches of the bs-16 step (profiles/r03_f_gemm_shapes_bs16.txt).
Interleaved rounds in one process, median per arm; cold-ish operands (a 512 MB scratch write between launches would change nothing for
HBM-bound shapes, so the loop simply rotates over 3 operand sets larger than the L2).  Prints one line per shape.
    AB_EPI=res|geglu|none   epilogue flavour (default by shape: N = 2560/5120/10240 -> fused GEGLU, N == K -> residual)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch  # noqa: E402
from pcm_amd import capi, ops  # noqa: E402
capi.set_lib(capi.tools_lib())      # the TOOLS build of the library: the pcm_debug_* hooks used below are not in the product build

dll = capi.lib().dll
shapes = [(131072, 2560, (320, 64)), (131072, 2560, (320,)), (131072, 320, (320, 64)), (131072, 320, (320,)), (32768, 5120, (640, 64)), (32768, 5120, (640,)),
          (8192, 1280, (1280, 64)), (8192, 1280, (1280,)), (32768, 640, (640, 64)), (32768, 640, (640,)), (8192, 10240, (1280, 64)), (8192, 10240, (1280,)),
          (131072, 960, (320, 192)), (131072, 960, (320,)), (16384, 640, (640, 64)), (131072, 320, (1280, 64)), (131072, 320, (1280,)), (65536, 320, (320, 64)),
          (4096, 1280, (1280, 64)), (65536, 1280, (320, 64)), (32768, 1920, (640, 192)), (32768, 1920, (640,)), (8192, 3840, (1280, 192)), (8192, 3840, (1280,)),
          (16384, 2560, (640, 64)), (4096, 5120, (1280, 64)), (65536, 640, (320, 64)), (131072, 320, (640,)), (65536, 320, (960, 192)),
          (8192, 1280, (2560, 64)), (32768, 640, (2560, 64)), (65536, 320, (2560, 64)), (8192, 1280, (5120, 64)), (16384, 640, (5120, 64))]
only = os.environ.get("AB_ONLY")
ROUNDS, REP = 5, 6


def timed(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(REP):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / REP


tot = {1: 0.0, 2: 0.0, 3: 0.0}
for (M, N, Ks) in shapes:
    if only and only not in str((M, N, Ks)):
        continue
    epi = os.environ.get("AB_EPI") or ("geglu" if N in (2560, 5120, 10240) else ("res" if N == Ks[0] or Ks[0] >= 1280 else "none"))
    sets = []
    for r in range(3):
        x = torch.randn(M, Ks[0], device="cuda").bfloat16()
        w = (torch.randn(N, Ks[0], device="cuda") * 0.05).bfloat16()
        segs = [ops.Seg(x, w)]
        if len(Ks) > 1:
            segs.append(ops.Seg(torch.randn(M, Ks[1], device="cuda").bfloat16(), (torch.randn(N, Ks[1], device="cuda") * 0.05).bfloat16()))
        kw = {}
        if epi == "geglu":
            out = torch.empty(M, N // 2, device="cuda", dtype=torch.bfloat16)
            kw = dict(act=capi.ACT_GEGLU, ldo=N // 2, bias=torch.randn(N, device="cuda"))
            if len(Ks) > 1:     # the grad-requiring half keeps the interleaved pre-activation (online half of the 2B pass)
                kw["pre_out"] = torch.empty(M // 2, N, device="cuda", dtype=torch.bfloat16)
        else:
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            if epi == "res":
                kw = dict(residual=torch.randn(M, N, device="cuda").bfloat16(), bias=torch.randn(N, device="cuda"))
        sets.append((segs, out, kw))
    fl = 2.0 * M * N * sum(Ks)
    nbytes = M * sum(Ks) * 2 + N * sum(Ks) * 2 + sets[0][1].numel() * 2 + (M * N * 2 if epi == "res" else 0) + (M // 2 * N * 2 if "pre_out" in sets[0][2] else 0)
    res, plans = {1: [], 2: [], 3: []}, {}
    for rnd in range(ROUNDS + 1):
        for mode in (2, 3, 1):
            dll.pcm_debug_gemm_big_mode(mode)
            ms = timed(lambda i: ops.gemm(sets[i % 3][0], M, N, sets[i % 3][1], **sets[i % 3][2]))
            plans[mode] = dll.pcm_debug_last_gemm_plan()
            if rnd:
                res[mode].append(ms)
    dll.pcm_debug_gemm_big_mode(1)
    med = {m: sorted(v)[len(v) // 2] for m, v in res.items()}
    for m in med:
        tot[m] += med[m]
    print("%-30s %-5s | 8p(plan %5d) %7.1f us %6.0f TF %5.2f TB/s | 4w(plan %5d) %7.1f us %6.0f TF %5.2f TB/s | x%.2f | planner(plan %5d) %7.1f us" %
          (str((M, N, Ks)), epi, plans[2], med[2] * 1e3, fl / med[2] / 1e9, nbytes / med[2] / 1e9, plans[3], med[3] * 1e3, fl / med[3] / 1e9, nbytes / med[3] / 1e9,
           med[2] / med[3], plans[1], med[1] * 1e3), flush=True)
print("sum of medians: 8p %.3f ms | 4w %.3f ms | planner %.3f ms" % (tot[2], tot[3], tot[1]))
