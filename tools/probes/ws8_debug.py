"""which epilogue operand breaks the eight-wave weights-stationary form on hardware: bias only / residual only / both, error pattern by row and column"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch  # noqa: E402
from pcm_amd import capi, ops  # noqa: E402
capi.set_lib(capi.tools_lib())
dll = capi.lib().dll
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g, device=dev) * scale).to(ops.BF16)   # noqa: E731
for M in (16384, 16384 + 72, 131072):
    for bias, res in ((True, False), (False, True), (True, True)):
        K, N = 320, 320
        x, w = rnd(M, K), rnd(N, K, scale=0.1)
        b = torch.randn(N, generator=g, device=dev) if bias else None
        r = rnd(M, N) if res else None
        outs = {}
        for on in (0, 2):
            dll.pcm_debug_gemm_ws(on)
            o = torch.zeros(M, N, dtype=ops.BF16, device=dev)
            for _ in range(3):
                ops.gemm([ops.Seg(x, w)], M, N, o, bias=b, residual=r)
            torch.cuda.synchronize()
            outs[on] = o.float()
        dll.pcm_debug_gemm_ws(-1)
        d = (outs[2] - outs[0]).abs()
        bad = d > 0
        rows, cols = bad.any(1).nonzero().flatten(), bad.any(0).nonzero().flatten()
        print("M %6d bias %d res %d: max diff %.3f, bad elements %d, bad rows %d (first %s), bad cols %d (first %s)" % (
            M, bias, res, float(d.max()), int(bad.sum()), rows.numel(), rows[:8].tolist(), cols.numel(), cols[:12].tolist()), flush=True)
        if bad.any():
            r0 = int(rows[0])
            print("   row %d (step %d, row-in-step %d): cols %s" % (r0, r0 // 64, r0 % 64, bad[r0].nonzero().flatten()[:20].tolist()))
            rm = (rows % 64).bincount(minlength=64)
            print("   bad rows by row-in-step:", rm.tolist())
