#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/f; mkdir -p $O
tools/probes/trbench > $O/trbench.txt 2>&1
python - > $O/wgrad_blocks.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, "phased-consistency-model_amd")
import torch
from pcm_amd import ops, capi
dll = capi.lib().dll
def bench(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
B = 16
for (H, C) in [(64, 320), (32, 640), (16, 1280), (8, 1280)]:
    x = torch.randn(B, H*H, C, device="cuda").bfloat16(); M = B*H*H
    u = torch.randn(M, 64, device="cuda").bfloat16(); out = torch.zeros(64, 3, 3, C, device="cuda")
    f = lambda: ops.lora_wgrad(x, u, out, 1.0, M, conv=dict(Hs=H, Ws=H, Ho=H, Wo=H), g_stride=1, r_stride=9 * C)
    row = []
    for nb in (32, 64, 128, 256, 512, 1024, 2048):
        dll.pcm_debug_wgrad_tr_blocks(nb); row.append("%d:%.1f" % (nb, bench(f)))
    dll.pcm_debug_wgrad_tr_blocks(512); dll.pcm_debug_wgrad_tr(0); row.append("old:%.1f" % bench(f)); dll.pcm_debug_wgrad_tr(1)
    print("conv H=%d C=%d us by target blocks: %s" % (H, C, "  ".join(row)), flush=True)
for (M, G) in [(65536, 320), (65536, 2560), (16384, 640), (4096, 1280)]:
    dy = torch.randn(M, G, device="cuda").bfloat16(); t = torch.randn(M, 64, device="cuda").bfloat16(); out = torch.zeros(G, 64, device="cuda")
    f = lambda: ops.lora_wgrad(dy, t, out, 0.125, M, g_stride=64, r_stride=1)
    row = []
    for nb in (64, 128, 256, 512, 1024, 2048):
        dll.pcm_debug_wgrad_tr_blocks(nb); row.append("%d:%.1f" % (nb, bench(f)))
    dll.pcm_debug_wgrad_tr_blocks(512); dll.pcm_debug_wgrad_tr(0); row.append("old:%.1f" % bench(f)); dll.pcm_debug_wgrad_tr(1)
    print("plain M=%d G=%d us by target blocks: %s" % (M, G, "  ".join(row)), flush=True)
PY
cat $O/trbench.txt $O/wgrad_blocks.txt
