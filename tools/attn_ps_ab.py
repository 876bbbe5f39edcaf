"""A/B of the pre-scaled-query attention kernels (csrc/attention_ps.hip: pcm_attn_fwd_prescaled / pcm_attn_bwd_prescaled) against the
kernels they replace (attention.hip / attention_fwd.hip) on the step's shapes: interleaved timing (min over rounds) of forward and
backward, and the error of both against an fp32 torch evaluation of the SAME 16-bit operands (q' = rn(q * d^-1/2 * log2 e) for the new ones).
usage: attn_ps_ab.py [rounds]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import ops, capi
capi.set_lib(capi.tools_lib())      # the TOOLS build of the library: pcm_debug_attn_ps_dma selects the staging of the new kernels
dll = capi.lib().dll
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
# optional third arm: the pre-scaled kernels of ANOTHER build of the library (a tools/probes/build_variant.py variant), e.g. a different
# launch bound / staging of one kernel:  attn_ps_ab.py 3 tools/probes/libpcm_<name>.so
ALT = capi.Lib(os.path.abspath(sys.argv[2])) if len(sys.argv) > 2 else None
MAIN = capi.lib()


def with_alt(fn):
    capi.set_lib(ALT)
    try:
        return fn()
    finally:
        capi.set_lib(MAIN)


def bench(fn, n=6):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def ref_fp32(q, k, v, dO, H, d, factor):
    B, Lq, Lk = q.shape[0], q.shape[1], k.shape[1]
    qr, kr, vr = [t.float().requires_grad_(True) for t in (q, k, v)]
    hd = lambda t, L: t.view(B, L, H, d).transpose(1, 2)   # noqa: E731
    outs = []
    for b0 in range(B):      # per image: the explicit score matrix of 8 heads x 4096^2 is 0.5 GB in fp32
        s = hd(qr, Lq)[b0] @ hd(kr, Lk)[b0].transpose(-1, -2) * factor
        outs.append((torch.softmax(s, -1) @ hd(vr, Lk)[b0]).transpose(0, 1).reshape(Lq, H * d))
    o = torch.stack(outs)
    o.backward(dO.float())
    return o.detach(), qr.grad, kr.grad, vr.grad


rel = lambda a, b: float((a.float() - b).norm() / (b.norm() + 1e-30))   # noqa: E731
for (B, L, Lk, d, H) in [(32, 4096, 4096, 40, 8), (16, 4096, 4096, 40, 8), (32, 1024, 1024, 80, 8), (32, 256, 256, 160, 8), (32, 4096, 77, 40, 8),
                         (32, 1024, 77, 80, 8), (4, 4096, 4096, 64, 10), (4, 1024, 1024, 64, 20)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    q, k, v, dO = [torch.randn(B, n, H * d, device="cuda", generator=g).to(ops.BF16) for n in (L, Lk, Lk, L)]
    qs = (q.float() * ops.attn_q_scale(d)).to(ops.BF16)
    fl = 4.0 * B * H * L * Lk * d
    t = {}
    o0, l0 = ops.attn_fwd(q, k, v, H, d)
    o1, l1 = ops.attn_fwd(qs, k, v, H, d, prescaled=True)
    g0 = ops.attn_bwd(q, k, v, o0, dO, l0, H, d)
    g1 = ops.attn_bwd(qs, k, v, o1, dO, l1, H, d, prescaled=True)
    for key in ("f0", "f1", "b0", "b1", "f2", "b2"):
        t[key] = 1e9
    for _ in range(rounds):
        t["f0"] = min(t["f0"], bench(lambda: ops.attn_fwd(q, k, v, H, d)))
        t["b0"] = min(t["b0"], bench(lambda: ops.attn_bwd(q, k, v, o0, dO, l0, H, d)))
        dll.pcm_debug_attn_ps_dma(1)
        t["f1"] = min(t["f1"], bench(lambda: ops.attn_fwd(qs, k, v, H, d, prescaled=True)))
        t["b1"] = min(t["b1"], bench(lambda: ops.attn_bwd(qs, k, v, o1, dO, l1, H, d, prescaled=True)))
        dll.pcm_debug_attn_ps_dma(0)            # register staging of the same kernels
        t["f2"] = min(t["f2"], bench(lambda: ops.attn_fwd(qs, k, v, H, d, prescaled=True)))
        t["b2"] = min(t["b2"], bench(lambda: ops.attn_bwd(qs, k, v, o1, dO, l1, H, d, prescaled=True)))
        dll.pcm_debug_attn_ps_dma(1)
        if ALT is not None:
            t["f3"] = min(t.get("f3", 1e9), with_alt(lambda: bench(lambda: ops.attn_fwd(qs, k, v, H, d, prescaled=True))))
            t["b3"] = min(t.get("b3", 1e9), with_alt(lambda: bench(lambda: ops.attn_bwd(qs, k, v, o1, dO, l1, H, d, prescaled=True))))
            ga = with_alt(lambda: ops.attn_bwd(qs, k, v, o1, dO, l1, H, d, prescaled=True))
            t["alt_err"] = max(rel(a, b.float()) for a, b in zip(ga, g1))            # the alternate library's gradients against the main library's
    nb = min(B, 2)       # accuracy on the first images only (fp32 reference on the GPU)
    sl = lambda x: x[:nb]    # noqa: E731
    r0 = ref_fp32(sl(q), sl(k), sl(v), sl(dO), H, d, d ** -0.5)
    r1 = ref_fp32(sl(qs), sl(k), sl(v), sl(dO), H, d, 0.6931471805599453)
    e0 = [rel(sl(a), b) for a, b in zip((o0,) + tuple(g0), r0)]
    e1 = [rel(sl(a), b) for a, b in zip((o1,) + tuple(g1), r1)]
    print("B=%2d H=%2d L=%4d Lk=%4d d=%3d | fwd %7.3f -> %7.3f ms (x%.3f, %4.0f -> %4.0f TF/s; register staging %7.3f) | bwd %7.3f -> %7.3f ms (x%.3f; register staging %7.3f) | rel-L2 o/dq/dk/dv old %s new %s"
          % (B, H, L, Lk, d, t["f0"], t["f1"], t["f0"] / t["f1"], fl / t["f0"] / 1e9, fl / t["f1"] / 1e9, t["f2"], t["b0"], t["b1"], t["b0"] / t["b1"], t["b2"],
             " ".join("%.1e" % x for x in e0), " ".join("%.1e" % x for x in e1)) +
          ((" | alt lib fwd %7.3f bwd %7.3f (gradients vs main lib %.1e)" % (t["f3"], t["b3"], t["alt_err"])) if ALT is not None else ""), flush=True)
