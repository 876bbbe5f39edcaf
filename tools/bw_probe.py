"""Reference streaming rates on this GPU at the tensor sizes of the UNet hot path (tuning tool): torch copy / add / sum vs
our layer-norm, group-norm and rank-64 projection kernels on the same bytes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import ops, capi
capi.lib()
def bench(fn, n=20, flush=None):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        if flush is not None: flush.add_(1.0)          # 512 MB touch: evicts L2 + infinity cache between timed calls
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]
flush = torch.zeros(128 * 1024 * 1024, device="cuda")
for (M, C) in [(65536, 320), (131072, 320), (16384, 640), (65536, 1280)]:
    x = torch.randn(M, C, device="cuda").bfloat16(); y = torch.empty_like(x); z = torch.randn(M, C, device="cuda").bfloat16()
    mb = M * C * 2 / 1e6
    for name, fl in (("warm", None), ("cold", flush)):
        t_copy = bench(lambda: y.copy_(x), flush=fl)
        t_add = bench(lambda: torch.add(x, z, out=y), flush=fl)
        t_sum = bench(lambda: x.sum(), flush=fl)
        g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        t_ln = bench(lambda: ops.layernorm_fwd(x.view(1, M, C), g, b), flush=fl)
        w = (torch.randn(64, C, device="cuda") * 0.05).bfloat16(); o = torch.empty(M, 64, device="cuda", dtype=torch.bfloat16)
        t_n64 = bench(lambda: ops.gemm([ops.Seg(x, w)], M, 64, o), flush=fl)
        print("%-14s %5.0f MB %s | copy %6.1f us %5.2f TB/s | add %6.1f us %5.2f TB/s | sum %6.1f us %5.2f TB/s | ln_fwd %6.1f us %5.2f TB/s | n64 %6.1f us %5.2f TB/s" % (
            str((M, C)), mb, name, t_copy * 1e3, 2 * mb / t_copy / 1e3, t_add * 1e3, 3 * mb / t_add / 1e3, t_sum * 1e3, mb / t_sum / 1e3,
            t_ln * 1e3, 2 * mb / t_ln / 1e3, t_n64 * 1e3, (mb + M * 128 / 1e6) / t_n64 / 1e3), flush=True)
