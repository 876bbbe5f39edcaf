"""GroupNorm streaming rates at the UNet's (batch, pixels, channels) shapes (tuning tool): statistics / apply passes,
forward and backward, against a torch copy of the same bytes.  `warm` = back-to-back calls (operands may sit in the
infinity cache, as they do right after the producing GEMM); `cold` = a 512 MB touch between timed calls."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import capi
capi.set_lib(capi.tools_lib())      # the TOOLS build of the library: the pcm_debug_* hooks used below are not in the product build
from pcm_amd.ops import ptr
L = capi.lib()
S = capi.Lib.stream


def bench(fn, n=12, flush=None):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        if flush is not None:
            flush.add_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


flush = torch.zeros(128 * 1024 * 1024, device="cuda")
shapes = [(32, 4096, 320), (32, 4096, 640), (32, 4096, 960), (32, 1024, 640), (32, 1024, 1280), (32, 1024, 1920), (32, 256, 1280),
          (32, 256, 2560), (32, 64, 1280), (16, 4096, 320), (16, 4096, 960)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
print("%-18s %6s %5s | %-16s | %-16s | %-16s | %-16s | %-16s" % ("(B,HW,C)", "MB", "", "copy", "stats", "apply", "bwd_stats", "bwd_apply"))
for (B, HW, C) in shapes:
    x = torch.randn(B, HW, C, device="cuda").bfloat16()
    dy = torch.randn(B, HW, C, device="cuda").bfloat16()
    y = torch.empty_like(x)
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    st = torch.zeros(B, 32, 2, dtype=torch.float64, device="cuda")
    bst = torch.zeros(B, 32, 2, dtype=torch.float64, device="cuda")
    L.dll.pcm_debug_gn_target(4096)
    nws = L.dll.pcm_groupnorm_workspace_bytes(B, HW, C, 32)      # sized for the largest target probed below
    L.dll.pcm_debug_gn_target(0)
    ws = torch.empty(nws // 8, dtype=torch.float64, device="cuda")
    L.call("pcm_groupnorm_stats", ptr(x), ptr(st), B, HW, C, 32, S())
    mb = B * HW * C * 2 / 1e6
    for name, fl in (("warm", None), ("cold", flush)):
        t_copy = bench(lambda: y.copy_(x), flush=fl)
        t_st = bench(lambda: L.call("pcm_groupnorm_stats_ws", ptr(x), ptr(bst), B, HW, C, 32, ptr(ws), nws, S()), flush=fl)
        extra = []
        if name == "warm":
            t_at = bench(lambda: L.call("pcm_groupnorm_stats_acc", ptr(x), ptr(bst), B, HW, C, 32, S()), flush=fl)
            extra.append("atomic512 %.1f" % (t_at * 1e3))
            for tgt in (512, 1024, 4096):
                L.dll.pcm_debug_gn_target(tgt)
                t_x = bench(lambda: L.call("pcm_groupnorm_stats_ws", ptr(x), ptr(bst), B, HW, C, 32, ptr(ws), nws, S()), flush=fl)
                extra.append("ws%d %.1f" % (tgt, t_x * 1e3))
            L.dll.pcm_debug_gn_target(0)
        t_ap = bench(lambda: L.call("pcm_groupnorm_apply", ptr(x), ptr(st), ptr(g), ptr(b), ptr(y), B, HW, C, 32, 1e-5, capi.ACT_SILU, S()), flush=fl)
        t_bs = bench(lambda: L.call("pcm_groupnorm_bwd_stats_ws", ptr(x), ptr(dy), ptr(st), ptr(g), ptr(b), ptr(bst), B, HW, C, 32, 1e-5, capi.ACT_SILU, ptr(ws), nws, S()), flush=fl)
        t_ba = bench(lambda: L.call("pcm_groupnorm_bwd_apply", ptr(x), ptr(dy), ptr(st), ptr(bst), ptr(g), ptr(b), ptr(y), B, HW, C, 32, 1e-5, capi.ACT_SILU, S()), flush=fl)
        f = lambda t, k: "%6.1f us %5.2f TB/s" % (t * 1e3, k * mb / t / 1e3)
        print("%-18s %6.0f %5s | %s | %s | %s | %s | %s" % (str((B, HW, C)), mb, name, f(t_copy, 2), f(t_st, 1), f(t_ap, 2), f(t_bs, 2), f(t_ba, 3)), " ".join(extra), flush=True)
