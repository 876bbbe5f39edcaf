"""Cycle timeline of one mid-grid tile of pcm_gemm8p_kernel<3> (PCM_ABLATE build with s_memtime stamps).
stamps per K-tile: 0 start | 1 ph1 reads+barrier | 2 ph1 MFMA+barrier | 3 ph2 in | 4 ph2 out | 5 ph3 in | 6 ph3 out | 7 ph4 reads+DMA+vmcnt | 8 ph4 in | 9 ph4 out"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import numpy as np, torch
from pcm_amd import ops, capi
capi.set_lib(capi.tools_lib())      # the TOOLS build of the library: the pcm_debug_* hooks used below are not in the product build
capi.set_lib(capi.Lib(os.path.join(ROOT, "tools", "probes", "libpcm_ablate.so")))
dll = capi.lib().dll
dll.pcm_debug_gemm_big_mode(2)
for (M, N, Ks, kind, Hs) in [(32768, 1280, (11520,), "conv", 32), (8192, 8192, (8192,), "lin", 0)]:
    segs = []
    if kind == "conv":
        Ci = Ks[0] // 9; B = M // (Hs * Hs)
        x = torch.randn(B, Hs, Hs, Ci, device="cuda").bfloat16(); w = (torch.randn(N, Ks[0], device="cuda") * 0.02).bfloat16()
        segs.append(ops.Seg(x, w, conv=dict(Hs=Hs, Ws=Hs)))
    else:
        x = torch.randn(M, Ks[0], device="cuda").bfloat16(); w = (torch.randn(N, Ks[0], device="cuda") * 0.05).bfloat16()
        segs.append(ops.Seg(x, w))
    if len(Ks) > 1:
        t = torch.randn(M, Ks[1], device="cuda").bfloat16(); bl = (torch.randn(N, Ks[1], device="cuda") * 0.05).bfloat16()
        segs.append(ops.Seg(t, bl))
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for mask in (0, 4):
        dll.pcm_debug_gemm_ablate(mask)
        for _ in range(3):
            ops.gemm(segs, M, N, out, Ho=Hs, Wo=Hs) if kind == "conv" else ops.gemm(segs, M, N, out)
        torch.cuda.synchronize()
        st = np.zeros((8, 16, 12), dtype=np.uint64)
        dll.pcm_debug_gemm8p_stamps(st.ctypes.data_as(ctypes.c_void_p))
        st = st.astype(np.int64)
        nt = min(16, sum(k // 64 for k in Ks))
        print("%s %s: per-K-tile cycle deltas, median over K-tiles 1..%d" % ((M, N, Ks, kind), "FULL" if mask == 0 else "no-MFMA", nt - 2))
        names = ["p1 rd+bar", "p1 mfma+bar", "p2 rd+bar", "p2 mfma+bar", "p3 rd+bar", "p3 mfma+bar", "p4 rd+dma+vm", "p4 bar", "p4 mfma+bar"]
        for w_ in (0, 4):
            tt = st[w_, 1:nt - 1]
            dl = [np.median(tt[:, i + 1] - tt[:, i]) for i in range(9)]
            tot = np.median(tt[1:, 0] - tt[:-1, 0]) if nt > 3 else -1
            print("          phase 3 detail: reads issued %4d | B1 DMA issued (+advance) %4d | lgkmcnt + barrier %4d" % (
                np.median(tt[:, 10] - tt[:, 4]), np.median(tt[:, 11] - tt[:, 10]), np.median(tt[:, 5] - tt[:, 11])))
            print("  wave %d: " % w_ + " | ".join("%s %4d" % (n, x) for n, x in zip(names, dl)) + " | K-tile %5d" % tot)
    dll.pcm_debug_gemm_ablate(0)
