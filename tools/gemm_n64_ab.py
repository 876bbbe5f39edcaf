"""A/B of block tiles for the rank-64 LoRA-down contractions (N = 64): HBM/latency-bound skinny GEMMs (tuning tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import ops, capi
capi.set_lib(capi.tools_lib())      # the TOOLS build of the library: the pcm_debug_* hooks used below are not in the product build
dll = capi.lib().dll
def bench(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
shapes = [(65536, 320, 0), (16384, 640, 0), (4096, 1280, 0), (1024, 1280, 0), (65536, 1280, 0), (65536, 2880, 64), (16384, 5760, 32), (4096, 11520, 16), (131072, 320, 0)]
for (M, K, Hs) in shapes:
    if Hs:
        Ci = K // 9; B = M // (Hs * Hs)
        x = torch.randn(B, Hs, Hs, Ci, device="cuda").bfloat16(); w = (torch.randn(64, K, device="cuda") * 0.02).bfloat16()
        seg = [ops.Seg(x, w, conv=dict(Hs=Hs, Ws=Hs))]
    else:
        x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(64, K, device="cuda") * 0.05).bfloat16()
        seg = [ops.Seg(x, w)]
    out = torch.empty(M, 64, device="cuda", dtype=torch.bfloat16)
    row = []
    for tile in [(0, 0), (256, 64), (128, 64), (64, 64)]:
        dll.pcm_debug_force_gemm_tile(*tile)
        ms = bench(lambda: ops.gemm(seg, M, 64, out, Ho=Hs, Wo=Hs) if Hs else ops.gemm(seg, M, 64, out))
        row.append("%s %6.1f us" % (tile, ms * 1e3))
    dll.pcm_debug_force_gemm_tile(0, 0)
    print("%-22s floor %5.1f us | %s" % (str((M, K, Hs)), (M * K * 2 if not Hs else M * K * 2 / 9) / 6.3e6, " | ".join(row)), flush=True)
