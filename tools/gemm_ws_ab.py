"""Weights-stationary kernel (csrc/gemm_ws.hip) against the phased tile on the short-K projections of the 64x64 level, per shape, interleaved:
    python tools/gemm_ws_ab.py        (TOOLS build: pcm_debug_gemm_ws switches the kernel: 0 phased tile, 1 four-wave form, 2 eight-wave form)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch  # noqa: E402

from pcm_amd import capi, ops  # noqa: E402

capi.set_lib(capi.tools_lib())
dll = capi.lib().dll
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)


def rnd(*s, scale=1.0):
    return (torch.randn(*s, generator=g, device=dev) * scale).to(ops.BF16)


def bench(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("%-34s %-10s %9s %9s %7s %9s %7s   %s" % ("(M, N, K segments)", "epilogue", "8p us", "ws4 us", "x", "ws8 us", "x", "ws8 GB/s (algorithmic bytes)"))
for M in (131072, 65536, 32768):
    for N, lora, res, bias in ((320, False, False, False), (320, False, True, True), (320, True, False, False), (320, True, True, True), (960, False, False, False)):
        K = 320
        x, w = rnd(M, K), rnd(N, K, scale=0.05)
        segs = [ops.Seg(x, w)]
        nbytes = M * K * 2 + N * K * 2 + M * N * 2
        if lora:
            t, bl = rnd(M, 64), rnd(N, 64, scale=0.05)
            segs.append(ops.Seg(t, bl))
            nbytes += M * 64 * 2 + N * 64 * 2
        b = torch.randn(N, generator=g, device=dev) if bias else None
        r = rnd(M, N) if res else None
        if res:
            nbytes += M * N * 2
        out = torch.empty(M, N, dtype=ops.BF16, device=dev)
        # evict between shapes is not needed: every call streams >= 100 MB
        ts = {}
        for rnd_ in range(2):
            for on in (0, 1, 2):
                if on == 2 and lora:
                    ts.setdefault(on, []).append(float("nan"))      # (the eight-wave form exists for K = 320 only)
                    continue
                dll.pcm_debug_gemm_ws(on)
                ts.setdefault(on, []).append(bench(lambda: ops.gemm(segs, M, N, out, bias=b, residual=r)))
        dll.pcm_debug_gemm_ws(-1)
        t8, tw, tw8 = min(ts[0]), min(ts[1]), min(ts[2])
        print("%-34s %-10s %9.1f %9.1f %7.3f %9.1f %7.3f   %.0f" % (str((M, N, (K, 64) if lora else (K,))), ("bias+res" if res else "none"), t8, tw, tw / t8, tw8, tw8 / t8,
                                                                nbytes / tw8 / 1e3), flush=True)
