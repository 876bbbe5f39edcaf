"""Workload for a rocprofv3 --pmc pass over the two forward-attention kernels (first kernel vs software-pipelined VGPR form) at the SD1.5
level-0 shape (d = 40) and the SDXL shape (d = 64): tools/pmc_table.py turns the counters into MFMA-busy / VALU-busy / wait fractions."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import capi, ops
capi.set_lib(capi.tools_lib())      # the TOOLS build of the library: the pcm_debug_* hooks used below are not in the product build
dll = capi.lib().dll
for (B, L, H, d) in [(16, 4096, 8, 40), (4, 4096, 10, 64)]:
    q = torch.randn(B, L, H * d, device="cuda").bfloat16(); k = torch.randn(B, L, H * d, device="cuda").bfloat16()
    v = torch.randn(B, L, H * d, device="cuda").bfloat16()
    for var in (0, 1):
        dll.pcm_debug_attn_fwd_variant(var)
        for _ in range(3):
            ops.attn_fwd(q, k, v, H, d)
torch.cuda.synchronize()
dll.pcm_debug_attn_fwd_variant(-1)
