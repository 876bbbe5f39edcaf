"""Quick GPU throughput probe of pcm_gemm_bf16 on the SD1.5 bs16 layer shapes (not a test)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import capi, ops

def bench(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

res = []
B = 16
for (name, M, N, K) in [("lin320", B*4096, 320, 320), ("ff320", B*4096, 2560, 320), ("ffo320", B*4096, 320, 1280),
                        ("lin640", B*1024, 640, 640), ("lin1280", B*256, 1280, 1280), ("ff1280", B*256, 10240, 1280),
                        ("big", 8192, 8192, 8192)]:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ms = bench(lambda: ops.gemm([ops.Seg(x, w)], M, N, out))
    res.append((name, M, N, K, ms, 2.0*M*N*K/ms/1e9))
for (name, Hs, Ci, Co) in [("c320@64", 64, 320, 320), ("c640@32", 32, 640, 640), ("c1280@16", 16, 1280, 1280),
                           ("c1280@8", 8, 1280, 1280), ("c2560-1280@16", 16, 2560, 1280), ("c960-320@64", 64, 960, 320)]:
    x = torch.randn(B, Hs, Hs, Ci, device="cuda").bfloat16(); w = (torch.randn(Co, 9*Ci, device="cuda")*0.02).bfloat16()
    M = B*Hs*Hs
    out = torch.empty(M, Co, device="cuda", dtype=torch.bfloat16)
    ms = bench(lambda: ops.gemm([ops.Seg(x, w, conv=dict(Hs=Hs, Ws=Hs))], M, Co, out, Ho=Hs, Wo=Hs))
    res.append((name, M, Co, 9*Ci, ms, 2.0*M*Co*9*Ci/ms/1e9))
for r in res:
    print("%-16s M=%6d N=%5d K=%5d  %8.3f ms  %8.1f TFLOP/s" % r)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "gemm_probe.json"), "w"))
print("---- under-filled / split-K shapes")
for (name, Hs, Ci, Co) in [("c1280@8 b16", 8, 1280, 1280), ("c2560-1280@8", 8, 2560, 1280), ("cA 1280->64 @8", 8, 1280, 64), ("cA 1280->64 @16", 16, 1280, 64), ("cA 2560->64 @16", 16, 2560, 64)]:
    x = torch.randn(B, Hs, Hs, Ci, device="cuda").bfloat16(); w = (torch.randn(Co, 9*Ci, device="cuda")*0.02).bfloat16()
    M = B*Hs*Hs
    out = torch.empty(M, Co, device="cuda", dtype=torch.bfloat16)
    ms = bench(lambda: ops.gemm([ops.Seg(x, w, conv=dict(Hs=Hs, Ws=Hs))], M, Co, out, Ho=Hs, Wo=Hs))
    print("%-16s M=%6d N=%5d K=%5d  %8.3f ms  %8.1f TFLOP/s" % (name, M, Co, 9*Ci, ms, 2.0*M*Co*9*Ci/ms/1e9))
