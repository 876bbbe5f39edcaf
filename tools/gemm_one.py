import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import ops
B = 16
for (Hs, Ci, Co) in [(64, 320, 320), (16, 1280, 1280)]:
    x = torch.randn(B, Hs, Hs, Ci, device="cuda").bfloat16(); w = (torch.randn(Co, 9*Ci, device="cuda")*0.02).bfloat16()
    M = B*Hs*Hs
    out = torch.empty(M, Co, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): ops.gemm([ops.Seg(x, w, conv=dict(Hs=Hs, Ws=Hs))], M, Co, out, Ho=Hs, Wo=Hs)
M, N, K = 65536, 320, 320
x = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16(); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3): ops.gemm([ops.Seg(x, w)], M, N, out)
torch.cuda.synchronize()
