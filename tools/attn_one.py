import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import ops
B, L, H, d = 16, 4096, 8, 40
q = torch.randn(B, L, H * d, device="cuda").bfloat16(); k = torch.randn(B, L, H * d, device="cuda").bfloat16()
v = torch.randn(B, L, H * d, device="cuda").bfloat16(); do = torch.randn(B, L, H * d, device="cuda").bfloat16()
for _ in range(2):
    o, lse = ops.attn_fwd(q, k, v, H, d)
    ops.attn_bwd(q, k, v, o, do, lse, H, d)
torch.cuda.synchronize()
