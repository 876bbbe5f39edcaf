"""Cycle timeline of one mid-grid workgroup of attn_fwd_kernel (PCM_ABLATE build with s_memtime stamps): where a key tile's time goes.
stamps: 0 loop top | 1 after barrier A | 2 after LDS restage | 3 after barrier B | 4 after QK^T + softmax issue | 5 after the PV MFMAs issue"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import numpy as np, torch
from pcm_amd import ops, capi
capi.set_lib(capi.tools_lib())      # the TOOLS build of the library: the pcm_debug_* hooks used below are not in the product build
capi.set_lib(capi.Lib(os.path.join(ROOT, "tools", "probes", "libpcm_ablate.so")))
dll = capi.lib().dll
for (B, L, d, H) in [(32, 4096, 40, 8), (32, 1024, 80, 8)]:
    qkv = torch.randn(B, L, 3 * H * d, device="cuda").bfloat16()
    q, k, v = qkv[:, :, :H * d], qkv[:, :, H * d:2 * H * d], qkv[:, :, 2 * H * d:]
    for _ in range(3): ops.attn_fwd(q, k, v, H, d)
    torch.cuda.synchronize()
    st = np.zeros((4, 32, 8), dtype=np.uint64)
    dll.pcm_debug_attn_stamps(st.ctypes.data_as(ctypes.c_void_p))
    st = st.astype(np.int64)
    nt = min(32, L // 64)
    print("B=%d L=%d d=%d: per-tile cycle deltas (median over tiles 2..%d), per wave" % (B, L, d, nt - 1))
    names = ["barrier A", "LDS restage (vmcnt wait + ds_write)", "barrier B", "QK^T + softmax", "PV", "loop back"]
    for w in range(4):
        t = st[w, 2:nt]
        dl = [np.median(t[:, i + 1] - t[:, i]) for i in range(5)] + [np.median(t[1:, 0] - t[:-1, 5])]
        tot = np.median(t[1:, 0] - t[:-1, 0])
        print("  wave %d: " % w + " | ".join("%s %5d" % (n, x) for n, x in zip(names, dl)) + " | tile total %5d" % tot)
