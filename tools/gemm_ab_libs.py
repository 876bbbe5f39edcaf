"""A/B two builds of the kernel library on GEMM shapes of the bs-16 step (short-K projections with and without bias / residual /
LoRA segment, and two 3x3 convs).  usage: gemm_ab_libs.py <libA.so> <libB.so>   (cold operands: a 400 MB eviction pass between launches
is NOT used here -- back-to-back launches like in the step; 10 timed launches each, interleaved A/B/A/B to cancel clock drift)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import ops, capi
capi.set_lib(capi.tools_lib())      # the TOOLS build of the library: the pcm_debug_* hooks used below are not in the product build
libs = [capi.Lib(os.path.abspath(p)) for p in sys.argv[1:] if p.endswith(".so")]
# third column: library B with its persistent tile loop switched off (when it has the hook) -- isolates the epilogue changes
cfgs = [(l, None) for l in libs]
if False and hasattr(libs[1].dll, "pcm_debug_gemm8p_persist"):
    cfgs.append((libs[1], 0))
def bench(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
shapes = [(8500, 1536, (1536, 64), "lin", 0, "b"), (8500, 6144, (1536, 64), "lin", 0, "b"), (8500, 1536, (6144, 64), "lin", 0, "b"), (65536, 640, (640, 64), "lin", 0, "br"), (16384, 5120, (1280, 64), "lin", 0, "bg"),
          (131072, 2560, (320, 64), "lin", 0, "bg"), (131072, 2560, (320,), "lin", 0, "bg"), (32768, 5120, (640,), "lin", 0, "bg"), (8192, 10240, (1280,), "lin", 0, "bg"),
          (131072, 320, (320,), "lin", 0, "br"), (131072, 320, (320,), "lin", 0, "b"), (131072, 960, (320,), "lin", 0, ""), (8192, 1280, (1280,), "lin", 0, "br"),
          (131072, 2560, (320, 64), "lin", 0, "b"), (131072, 2560, (320,), "lin", 0, "b"), (131072, 320, (320, 64), "lin", 0, ""), (131072, 320, (320, 64), "lin", 0, "br"),
          (131072, 320, (1280, 64), "lin", 0, "br"), (131072, 960, (320, 192), "lin", 0, ""), (32768, 5120, (640, 64), "lin", 0, "b"), (32768, 640, (640, 64), "lin", 0, "br"),
          (32768, 640, (640,), "lin", 0, ""), (8192, 1280, (1280, 64), "lin", 0, "br"), (65536, 320, (320, 64), "lin", 0, "br"), (16384, 640, (640, 64), "lin", 0, ""),
          (131072, 320, (2880, 64), "conv", 64, "bv"), (131072, 320, (2880, 64), "conv", 64, "br"), (32768, 1280, (11520, 64), "conv", 32, "br"), (8192, 1280, (11520, 64), "conv", 16, "bv")]
tot = [0.0] * len(cfgs)
for (M, N, Ks, kind, Hs, opt) in shapes:
    segs = []
    if kind == "conv":
        Ci = Ks[0] // 9; B = M // (Hs * Hs)
        x = torch.randn(B, Hs, Hs, Ci, device="cuda").bfloat16(); w = (torch.randn(N, Ks[0], device="cuda") * 0.02).bfloat16()
        segs.append(ops.Seg(x, w, conv=dict(Hs=Hs, Ws=Hs)))
    else:
        B = 32
        x = torch.randn(M, Ks[0], device="cuda").bfloat16(); w = (torch.randn(N, Ks[0], device="cuda") * 0.05).bfloat16()
        segs.append(ops.Seg(x, w))
    if len(Ks) > 1:
        t = torch.randn(M, Ks[1], device="cuda").bfloat16(); bl = (torch.randn(N, Ks[1], device="cuda") * 0.05).bfloat16()
        segs.append(ops.Seg(t, bl))
    kw = {}
    if "b" in opt: kw["bias"] = torch.randn(N, device="cuda")
    if "r" in opt: kw["residual"] = torch.randn(M, N, device="cuda").bfloat16()
    if "v" in opt: kw["rowvec"] = torch.randn(B, N, device="cuda").bfloat16(); kw["rows_per_batch"] = M // B
    if kind == "conv": kw.update(Ho=Hs, Wo=Hs)
    No = N
    if "g" in opt:          # fused GEGLU epilogue (interleaved value / gate columns): N/2 outputs
        kw.update(act=capi.ACT_GEGLU, ldo=N // 2); No = N // 2
    outs, ts = [], [[] for _ in cfgs]
    for rep in range(3):
        for i, (l, persist) in enumerate(cfgs):
            capi.set_lib(l)
            if persist is not None: l.dll.pcm_debug_gemm8p_persist(persist)
            out = torch.empty(M, No, device="cuda", dtype=torch.bfloat16)
            ts[i].append(bench(lambda: ops.gemm(segs, M, N, out, **kw)))
            if rep == 0: outs.append(out.float())
    mins = [min(t) for t in ts]
    same = float((outs[0] - outs[1]).abs().max())
    for i, v in enumerate(mins): tot[i] += v
    print("%-34s %-3s  A %7.1f us | B %7.1f us (x%.3f)%s  max|A-B| %.3g" % (str((M, N, Ks, kind)), opt, mins[0], mins[1], mins[1] / mins[0],
          "".join(" | %s %7.1f us (x%.3f)" % ("CDEF"[i - 2], mins[i], mins[i] / mins[0]) for i in range(2, len(mins))), same), flush=True)
print("sum: " + "  ".join("%.1f us (x%.3f)" % (t, t / tot[0]) for t in tot))
