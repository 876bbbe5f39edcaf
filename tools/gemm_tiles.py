"""A/B the block-tile configurations of pcm_gemm_bf16 on representative shapes (tuning tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import ops, capi
capi.set_lib(capi.tools_lib())      # the TOOLS build of the library: the pcm_debug_* hooks used below are not in the product build
L = capi.lib()
def bench(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
B = 16
shapes = [("lin", B*4096, 640, 640), ("lin", B*1024, 1280, 1280), ("lin", B*4096, 2560, 320), ("lin", 8192, 8192, 8192),
          ("conv", 32, 640, 640), ("conv", 16, 1280, 1280), ("conv", 64, 320, 640), ("conv32", 16, 1280, 1280)]
for tile in [(128, 128), (128, 128), (256, 128), (256, 64), (128, 64), (0, 0)]:
    L.dll.pcm_debug_force_gemm_tile(*tile)
    row = []
    for sh in shapes:
        if sh[0] == "lin":
            _, M, N, K = sh
            x = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16()
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            ms = bench(lambda: ops.gemm([ops.Seg(x, w)], M, N, out)); fl = 2.0*M*N*K
        else:
            _, Hs, Ci, Co = sh
            Bb = 32 if sh[0] == "conv32" else B
            x = torch.randn(Bb, Hs, Hs, Ci, device="cuda").bfloat16(); w = (torch.randn(Co, 9*Ci, device="cuda")*0.02).bfloat16()
            M = Bb*Hs*Hs; out = torch.empty(M, Co, device="cuda", dtype=torch.bfloat16)
            ms = bench(lambda: ops.gemm([ops.Seg(x, w, conv=dict(Hs=Hs, Ws=Hs))], M, Co, out, Ho=Hs, Wo=Hs)); fl = 2.0*M*Co*9*Ci
        row.append("%6.0f" % (fl/ms/1e9))
    print("tile %-10s TF/s: %s" % (str(tile), " ".join(row)))
print("shapes:", shapes)
