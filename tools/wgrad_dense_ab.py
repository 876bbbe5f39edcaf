"""A/B of the dense conv3x3 weight gradient (csrc/wgrad_dense.hip, pcm_conv3x3_wgrad_bf16) against the Cout/64 rank-64 launches it
replaces (wgrad_tr.hip / wgrad.hip through pcm_lora_wgrad_multi_bf16) on the discriminator-head geometries of the C3 discriminator step
(bs 8 per GPU -> 16 samples [fake; real]).  Interleaved rounds in one process, median per arm, 3 operand sets rotated (larger than the L2).
The M split of the dense kernel is fixed per process (PCM_WGRAD_DENSE_MSPLIT, 0 / unset = the shipped rule): run once per value."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch  # noqa: E402
from pcm_amd import ops  # noqa: E402

B = int(os.environ.get("AB_B", "16"))
geos = [(320, 32, 32), (640, 16, 16), (1280, 8, 8), (1280, 16, 16), (1280, 32, 32), (640, 64, 64), (320, 64, 64)]   # discriminator_sd15.py:377 taps
only = os.environ.get("AB_ONLY")
ROUNDS, REP = 5, 4


def timed(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(REP):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / REP * 1e3


tot = [0.0, 0.0]
for (C, H, W) in geos:
    if only and only not in str((C, H, W)):
        continue
    M = B * H * W
    sets = [(torch.randn(B, H, W, C, device="cuda").bfloat16(), torch.randn(M, C, device="cuda").bfloat16()) for _ in range(3)]
    g_old = torch.zeros(C, 9 * C, device="cuda")
    g_new = torch.zeros(C, 9 * C, device="cuda")
    wg = dict(Hs=H, Ws=W, Ho=H, Wo=W)

    def old(i):
        x, dy = sets[i % 3]
        with ops.wgrad_batch():
            for co in range(0, C, 64):
                ops.lora_wgrad(x, dy[:, co:co + 64], g_old[co:co + 64], 1.0, M, conv=wg, g_stride=1, r_stride=9 * C, lds=C)

    def new(i):
        x, dy = sets[i % 3]
        ops.conv3x3_wgrad(x, dy, g_new, B, H, W)

    old(0); new(0)
    torch.cuda.synchronize()
    err = float((g_old - g_new).abs().max()) / max(float(g_old.abs().max()), 1e-9)
    res = [[], []]
    for r in range(ROUNDS):
        res[0].append(timed(old))
        res[1].append(timed(new))
    t = [sorted(v)[len(v) // 2] for v in res]
    fl = 2.0 * M * 9 * C * C
    tot[0] += t[0]; tot[1] += t[1]
    print("C %4d %3dx%-3d M %6d  %6.1f GF   rank-64 x %2d %8.1f us %6.0f TF/s | dense %8.1f us %6.0f TF/s (x%.3f)   max|a-b|/max|a| %.1e"
          % (C, H, W, M, fl / 1e9, C // 64, t[0], fl / t[0] / 1e6, t[1], fl / t[1] / 1e6, t[1] / t[0], err), flush=True)
print("sum: rank-64 %.1f us | dense %.1f us (x%.3f)   [x 8 (4 heads x 2 convs) per discriminator step: %.2f -> %.2f ms; the 8x8 tap counts 3 times]"
      % (tot[0], tot[1], tot[1] / tot[0], tot[0] * 8e-3, tot[1] * 8e-3))
