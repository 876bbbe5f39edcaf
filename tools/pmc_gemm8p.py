"""Workload for the PMC passes (FETCH_SIZE / WRITE_SIZE) on the dominant kernel: the 64x64-resolution 3x3 conv of the step
(M = 131072 pixels, 320 -> 320 channels, LoRA segment) through pcm_gemm8p_kernel<3,false,false> (the shipped instantiation), 3 launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import ops, capi
capi.set_lib(capi.tools_lib())      # the TOOLS build of the library: the pcm_debug_* hooks used below are not in the product build
capi.lib()
B, Hs, Ci, Co = 32, 64, 320, 320
M = B * Hs * Hs
x = torch.randn(B, Hs, Hs, Ci, device="cuda").bfloat16(); w = (torch.randn(Co, 9 * Ci, device="cuda") * 0.02).bfloat16()
t = torch.randn(M, 64, device="cuda").bfloat16(); bl = (torch.randn(Co, 64, device="cuda") * 0.05).bfloat16()
out = torch.empty(M, Co, device="cuda", dtype=torch.bfloat16)
flush = torch.zeros(96 * 1024 * 1024, device="cuda")
for _ in range(3):
    flush.add_(1.0)     # evict the operands from L2 / infinity cache between launches
    ops.gemm([ops.Seg(x, w, conv=dict(Hs=Hs, Ws=Hs)), ops.Seg(t, bl)], M, Co, out, Ho=Hs, Wo=Hs)
torch.cuda.synchronize()
print("plan", capi.lib().dll.pcm_debug_last_gemm_plan(), "algorithmic bytes per launch: read %.1f MB (x %.1f + t %.1f + w %.1f), write %.1f MB" % (
    (x.numel() + t.numel() + w.numel() + bl.numel()) * 2 / 1e6, x.numel() * 2 / 1e6, t.numel() * 2 / 1e6, (w.numel() + bl.numel()) * 2 / 1e6, out.numel() * 2 / 1e6))
