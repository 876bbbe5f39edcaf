"""Where do the small-M launches of the deep UNet levels go?  (C3 at bs 8: every GEMM of the 16x16 / 8x8 levels has M <= 4096;
profiles/r04_m_gemm_shapes_c3.txt.)  Per shape: the planner's choice against forced plans -- the 256-row phased tile with its K split
(pcm_debug_gemm_big_mode 2), the 4-wave tiles 128x128 / 128x64 / 64x64 with K splits 1..8 (pcm_debug_force_gemm_tile(bm | split << 16, bn)).
3 operand sets rotated, median of 5 rounds, times include the split-K finalize launch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch  # noqa: E402
from pcm_amd import capi, ops  # noqa: E402
capi.set_lib(capi.tools_lib())      # the TOOLS build of the library: the pcm_debug_* hooks used below are not in the product build

dll = capi.lib().dll
lin = [(2048, 1280, 1280), (4096, 1280, 1280), (1024, 1280, 1280), (8192, 640, 640), (16384, 640, 640), (32768, 320, 320), (2048, 1280, 5120),
       (8192, 640, 2560), (4096, 1280, 5120), (2048, 10240, 1280), (8192, 1280, 1280), (8192, 3840, 1280), (616, 1280, 2048), (308, 1536, 1536), (2, 9216, 1536), (616, 1536, 1536),
       (16384, 1536, 1536), (8192, 1536, 1536), (8192, 1536, 6144), (2464, 1280, 768)]
conv = [(8, 8, 8, 1280), (16, 8, 8, 1280), (32, 8, 8, 1280), (16, 16, 16, 1280), (16, 16, 16, 640)]      # (B, H, W, C): C -> C 3x3
if os.environ.get("AB_SHAPES"):          # "M,N,K;M,N,K;..." replaces the linear list (and drops the convs)
    lin = [tuple(int(v) for v in t.split(",")) for t in os.environ["AB_SHAPES"].split(";")]
    conv = []
only = os.environ.get("AB_ONLY")
ROUNDS, REP = 5, 6


def timed(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(REP):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / REP * 1e3


def sweep(name, fl, call):
    arms = [("planner", 1, (0, 0)), ("8p", 2, (0, 0)), ("4w", 3, (0, 0))]
    for (bm, bn) in ((128, 128), (128, 64), (64, 64), (256, 256), (256, 320)):      # (+ (256, 192) with tools/probes/patches/r04_gemm8p_256x192_tile.patch applied)
        for sp in (1, 2, 3, 4, 6, 8):
            arms.append(("%dx%d/%d" % (bm, bn, sp), 0, (bm | (sp << 16), bn)))
    res = {a[0]: [] for a in arms}
    plans = {}
    for r in range(ROUNDS):
        for nm, mode, tile in arms:
            dll.pcm_debug_gemm_big_mode(mode if mode else 1)
            dll.pcm_debug_force_gemm_tile(*tile)
            call(0)
            plans[nm] = dll.pcm_debug_last_gemm_plan()
            res[nm].append(timed(call))
    dll.pcm_debug_gemm_big_mode(1); dll.pcm_debug_force_gemm_tile(0, 0)
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    best = min(med, key=med.get)
    rows = sorted(med.items(), key=lambda kv: kv[1])[:5]
    print("%-34s planner(plan %5d) %6.1f us %5.0f TF/s | 8p(plan %5d) %6.1f | best %-11s %6.1f us (x%.2f) | next: %s"
          % (name, plans["planner"], med["planner"], fl / med["planner"] / 1e6, plans["8p"], med["8p"], best, med[best], med[best] / med["planner"],
             "  ".join("%s %.1f" % kv for kv in rows[1:4])), flush=True)


for (M, N, K) in lin:
    if only and only not in str((M, N, K)):
        continue
    sets = [(torch.randn(M, K, device="cuda").bfloat16(), (torch.randn(N, K, device="cuda") * 0.05).bfloat16(), torch.empty(M, N, device="cuda", dtype=torch.bfloat16),
             torch.randn(M, N, device="cuda").bfloat16()) for _ in range(3)]
    bias = torch.randn(N, device="cuda")

    def call(i):
        x, w, o, r = sets[i % 3]
        ops.gemm([ops.Seg(x, w)], M, N, o, bias=bias, residual=r)
    sweep("lin  %s" % ((M, N, K),), 2.0 * M * N * K, call)
for (B, H, W, C) in conv:
    if only and only not in str((B, H, W, C)):
        continue
    M = B * H * W
    sets = [(torch.randn(B, H, W, C, device="cuda").bfloat16(), (torch.randn(C, 9 * C, device="cuda") * 0.02).bfloat16(), torch.empty(M, C, device="cuda", dtype=torch.bfloat16))
            for _ in range(3)]
    bias = torch.randn(C, device="cuda")

    def call(i):
        x, w, o = sets[i % 3]
        ops.gemm([ops.Seg(x, w, conv=dict(Hs=H, Ws=W))], M, C, o, bias=bias, Ho=H, Wo=W)
    sweep("conv %s M %d" % ((B, H, W, C), M), 2.0 * M * C * 9 * C, call)
