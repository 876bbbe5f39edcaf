"""GEGLU / concat / split streaming rates at the UNet's largest shapes (tuning tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import ops, capi
capi.lib()


def bench(fn, n=12):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


for M, C4 in [(131072, 1280), (65536, 1280), (32768, 2560), (8192, 5120)]:
    hg = torch.randn(M, 2 * C4, device="cuda").bfloat16()
    do = torch.randn(M, C4, device="cuda").bfloat16()
    t_f = bench(lambda: ops.geglu_fwd(hg))
    t_b = bench(lambda: ops.geglu_bwd(hg, do))
    mb = M * C4 * 2 / 1e6
    print("geglu M=%6d C4=%5d | fwd %7.1f us %5.2f TB/s | bwd %7.1f us %5.2f TB/s" % (M, C4, t_f * 1e3, 3 * mb / t_f / 1e3, t_b * 1e3, 5 * mb / t_b / 1e3), flush=True)
for rows, Ca, Cb in [(131072, 320, 320), (131072, 640, 320), (32768, 1280, 640), (65536, 320, 320)]:
    a = torch.randn(rows, Ca, device="cuda").bfloat16(); b = torch.randn(rows, Cb, device="cuda").bfloat16()
    cat = ops.concat_channels(a, b)
    t_c = bench(lambda: ops.concat_channels(a, b))
    t_s = bench(lambda: ops.split_channels(cat, Ca))
    mb = rows * (Ca + Cb) * 2 / 1e6
    print("concat rows=%6d %4d+%4d | cat %7.1f us %5.2f TB/s | split %7.1f us %5.2f TB/s" % (rows, Ca, Cb, t_c * 1e3, 2 * mb / t_c / 1e3, t_s * 1e3, 2 * mb / t_s / 1e3), flush=True)
