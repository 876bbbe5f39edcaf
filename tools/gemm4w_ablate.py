"""Where does the time of a short-K launch go?  The -DPCM_ABLATE build (tools/probes/build_ablate.py) of gemm4w.hip / gemm8p.hip with parts of
the kernel switched off (results are garbage by construction, only the timing is read), the start stagger of gemm4w's odd workgroup slot,
and the cycle stamps of one mid-grid gemm4w tile.  masks: 1 no global stores, 2 no epilogue, 4 no MFMA, 8 no LDS-DMA after the prologue;
sN = start stagger of N x ~0.85 us."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch  # noqa: E402
from pcm_amd import capi, ops  # noqa: E402
capi.set_lib(capi.tools_lib())      # the TOOLS build of the library: the pcm_debug_* hooks used below are not in the product build

capi.set_lib(capi.Lib(os.path.join(ROOT, "tools", "probes", "libpcm_ablate.so")))
dll = capi.lib().dll
REP = 6


def timed(fn):
    fn(0); fn(1)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(REP):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / REP * 1e3


shapes = [(131072, 320, (320,), "none"), (131072, 320, (320,), "res"), (131072, 320, (320, 64), "res"), (131072, 960, (320,), "none"),
          (131072, 2560, (320,), "geglu"), (32768, 640, (640,), "res"), (8192, 1280, (1280,), "res"), (32768, 5120, (640,), "geglu")]
for (M, N, Ks, epi) in shapes:
    sets = []
    for r in range(3):
        segs = [ops.Seg(torch.randn(M, Ks[0], device="cuda").bfloat16(), (torch.randn(N, Ks[0], device="cuda") * 0.05).bfloat16())]
        if len(Ks) > 1:
            segs.append(ops.Seg(torch.randn(M, Ks[1], device="cuda").bfloat16(), (torch.randn(N, Ks[1], device="cuda") * 0.05).bfloat16()))
        kw = {}
        if epi == "geglu":
            out = torch.empty(M, N // 2, device="cuda", dtype=torch.bfloat16)
            kw = dict(act=capi.ACT_GEGLU, ldo=N // 2, bias=torch.randn(N, device="cuda"))
        else:
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            if epi == "res":
                kw = dict(residual=torch.randn(M, N, device="cuda").bfloat16(), bias=torch.randn(N, device="cuda"))
        sets.append((segs, out, kw))
    run = lambda i: ops.gemm(sets[i % 3][0], M, N, sets[i % 3][1], **sets[i % 3][2])   # noqa: E731
    line = []
    for mode, name in ((2, "8p"), (3, "4w")):
        dll.pcm_debug_gemm_big_mode(mode)
        row = []
        for mask in (0, 1, 2, 4, 8, 2 | 4, 2 | 8, 4 | 8):
            dll.pcm_debug_gemm_ablate(mask)
            row.append("m%-2d %6.1f" % (mask, timed(run)))
        if mode == 3:
            for st in (1, 2, 4, 8, 16, 32):
                dll.pcm_debug_gemm_ablate((st + 1) << 8)
                row.append("s%-2d %6.1f" % (st, timed(run)))
        dll.pcm_debug_gemm_ablate(0)
        line.append("%s: %s" % (name, " | ".join(row)))
    print("%-34s %-5s\n   %s\n   %s" % (str((M, N, Ks)), epi, line[0], line[1]), flush=True)
    # stamps of one mid-grid gemm4w tile (mode 3 still set), without and with a stagger
    dll.pcm_debug_gemm_big_mode(3)
    for st in (0, 8):
        dll.pcm_debug_gemm_ablate(((st + 1) << 8) if st else (1 << 8))
        run(0)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 32)()
        dll.pcm_debug_gemm4w_stamps(buf)
        w0 = [buf[i] for i in range(8)]
        d = [w0[i] - w0[0] for i in range(1, 7)]
        print("   stamps (stagger %d; cycles from entry, wave 0): K0 landed %d | K loop done %d | passes %d %d %d | end %d | hwid %#x tg %d" %
              (st, d[0], d[1], d[2], d[3], d[4], d[5], w0[7], (w0[7] >> 16) & 15), flush=True)
    dll.pcm_debug_gemm_ablate(0)
    dll.pcm_debug_gemm_big_mode(1)
