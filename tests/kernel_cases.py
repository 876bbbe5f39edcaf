"""Kernel parity cases shared by the host-emulation (CPU) and the GPU test files.  Every case
compares the C-ABI op against plain torch fp32 evaluated on the SAME 16-bit-rounded inputs (ops.BF16: bfloat16, or float16 when the half build is selected)."""
import math

import pytest
import torch
import torch.nn.functional as F

from pcm_amd import capi, ops


def rnd(*shape, seed=0, scale=1.0, dev="cpu", dtype=None, shift=0.0):
    dtype = ops.BF16 if dtype is None else dtype        # the library's 16-bit dtype (float16 under precision.set_precision("fp16"))
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale + shift).to(dtype).to(dev)


def close(a, b, rtol, atol, what=""):
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), f"{what}: max err {err.max().item():.3e} (max ref {b.abs().max().item():.3e})"


def rel_l2(a, b, tol, what=""):
    a, b = a.float().cpu(), b.float().cpu()
    r = float((a - b).norm() / b.norm().clamp_min(1e-30))
    assert r <= tol, f"{what}: rel-L2 {r:.3e} > {tol:.1e} (ref rms {float(b.pow(2).mean().sqrt()):.3e})"
    return r


def case_groupnorm(dev, B, HW, C, G, act, eps=1e-5):
    x = rnd(B, HW, C, seed=1, dev=dev, shift=0.3)
    gamma = rnd(C, seed=2, dev=dev, dtype=torch.float32, shift=1.0, scale=0.2)
    beta = rnd(C, seed=3, dev=dev, dtype=torch.float32, scale=0.2)
    dy = rnd(B, HW, C, seed=4, dev=dev)
    y, stats = ops.groupnorm_fwd(x, gamma, beta, G, eps, act)
    xr = x.float().cpu().permute(0, 2, 1).requires_grad_(True)  # [B, C, HW]
    z = F.group_norm(xr, G, gamma.cpu(), beta.cpu(), eps)
    ref = F.silu(z) if act else z
    close(y.permute(0, 2, 1), ref.detach(), 1e-2, 1e-2, "gn fwd")
    ref.backward(dy.float().cpu().permute(0, 2, 1))
    dx = ops.groupnorm_bwd(x, dy, stats, gamma, beta, G, eps, act)
    close(dx.permute(0, 2, 1), xr.grad, 1e-2, 2e-2, "gn bwd")
    # the atomic form of the two reductions (pre-zeroed arena slices; the runner uses it from batch 16 up) gives the same numbers
    arena = ops.StatArena(x.device, slots=2, per_slot=B * G * 2)
    y2, stats2 = ops.groupnorm_fwd(x, gamma, beta, G, eps, act, arena=arena)
    dx2 = ops.groupnorm_bwd(x, dy, stats2, gamma, beta, G, eps, act, arena=arena)
    assert arena.used == 2
    close(stats2, stats, 1e-5, 1e-3, "gn stats atomic vs partials")
    close(y2, y, 1e-6, 1e-2, "gn fwd atomic vs partials")
    close(dx2, dx, 1e-6, 2e-2, "gn bwd atomic vs partials")
    # round 6: the skip-path gradient joins in the apply pass (pcm_groupnorm_bwd_apply_res): dx + dres with ONE rounding of the sum
    dres = rnd(B, HW, C, seed=5, dev=dev)
    dx3 = ops.groupnorm_bwd(x, dy, stats, gamma, beta, G, eps, act, dres=dres)
    close(dx3.permute(0, 2, 1), xr.grad + dres.float().cpu().permute(0, 2, 1), 1e-2, 2e-2, "gn bwd + residual gradient")


def case_layernorm(dev, M, C):
    x = rnd(M, C, seed=1, dev=dev, shift=0.2)
    gamma = rnd(C, seed=2, dev=dev, dtype=torch.float32, shift=1.0, scale=0.2)
    beta = rnd(C, seed=3, dev=dev, dtype=torch.float32, scale=0.2)
    dy = rnd(M, C, seed=4, dev=dev)
    dres = rnd(M, C, seed=5, dev=dev)
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta)
    xr = x.float().cpu().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gamma.cpu(), beta.cpu(), 1e-5)
    close(y, ref.detach(), 1e-2, 1e-2, "ln fwd")
    ref.backward(dy.float().cpu())
    dx = ops.layernorm_bwd(x, dy, gamma, mean, rstd, dres)
    close(dx, xr.grad + dres.float().cpu(), 1e-2, 2e-2, "ln bwd")


def case_elementwise(dev):
    B, H, W, C = 2, 3, 5, 64
    x = rnd(B, H * W, C, seed=1, dev=dev)
    up = ops.upsample2x(x, B, H, W)
    ref = F.interpolate(x.float().cpu().view(B, H, W, C).permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest")
    assert torch.equal(up.float().cpu().view(B, 2 * H, 2 * W, C).permute(0, 3, 1, 2), ref)
    dy = rnd(B, 4 * H * W, C, seed=2, dev=dev)
    dx = ops.pool2x_sum(dy, B, H, W)
    refp = F.avg_pool2d(dy.float().cpu().view(B, 2 * H, 2 * W, C).permute(0, 3, 1, 2), 2) * 4
    close(dx.view(B, H, W, C).permute(0, 3, 1, 2), refp, 1e-2, 1e-2, "pool")
    a, b = rnd(7, 64, seed=3, dev=dev), rnd(7, 128, seed=4, dev=dev)
    cat = ops.concat_channels(a, b)
    assert torch.equal(cat.cpu(), torch.cat([a.cpu(), b.cpu()], -1))
    acc = rnd(7, 64, seed=5, dev=dev)
    acc0 = acc.clone()
    a2, b2 = ops.split_channels(cat, 64, a_out=acc, accumulate_a=True)
    assert torch.equal(b2.cpu(), b.cpu())
    close(a2, acc0.float().cpu() + a.float().cpu(), 1e-2, 1e-2, "split acc")
    close(ops.add(a, acc0), a.float().cpu() + acc0.float().cpu(), 1e-2, 1e-2, "add")
    close(ops.silu(a), F.silu(a.float().cpu()), 1e-2, 1e-2, "silu")
    ar = a.float().cpu().requires_grad_(True)
    F.silu(ar).backward(acc0.float().cpu())
    close(ops.silu_bwd(a, acc0), ar.grad, 1e-2, 1e-2, "silu bwd")
    hg = rnd(5, 128, seed=6, dev=dev)
    hr = hg.float().cpu().requires_grad_(True)
    h, g = hr.chunk(2, -1)
    refg = h * F.gelu(g)
    close(ops.geglu_fwd(hg), refg.detach(), 1e-2, 1e-2, "geglu")
    do = rnd(5, 64, seed=7, dev=dev)
    refg.backward(do.float().cpu())
    close(ops.geglu_bwd(hg, do), hr.grad, 1e-2, 2e-2, "geglu bwd")
    # larger grids: several trips of the batched row/column walker per thread, ragged tail, odd column counts
    a, b = rnd(301, 72, seed=13, dev=dev), rnd(301, 40, seed=14, dev=dev)
    cat = ops.concat_channels(a, b)
    assert torch.equal(cat.cpu(), torch.cat([a.cpu(), b.cpu()], -1))
    acc = rnd(301, 72, seed=15, dev=dev)
    acc0 = acc.clone()
    a2, b2 = ops.split_channels(cat, 72, a_out=acc, accumulate_a=True)
    assert torch.equal(b2.cpu(), b.cpu())
    close(a2, acc0.float().cpu() + a.float().cpu(), 1e-2, 1e-2, "split acc (large)")
    a3, b3 = ops.split_channels(cat, 72)
    assert torch.equal(a3.cpu(), a.cpu()) and torch.equal(b3.cpu(), b.cpu())
    hg = rnd(203, 2 * 104, seed=16, dev=dev)
    hr = hg.float().cpu().requires_grad_(True)
    h, g = hr.chunk(2, -1)
    refg = h * F.gelu(g)
    close(ops.geglu_fwd(hg), refg.detach(), 1e-2, 1e-2, "geglu (large)")
    do = rnd(203, 104, seed=17, dev=dev)
    refg.backward(do.float().cpu())
    close(ops.geglu_bwd(hg, do), hr.grad, 1e-2, 2e-2, "geglu bwd (large)")
    xs = rnd(2, 37, 320, seed=8, dev=dev)
    close(ops.colsum(xs), xs.float().cpu().sum(1), 1e-4, 1e-3, "colsum")
    close(ops.cast_f32(ops.cast_bf16(xs.float())), xs, 0, 0, "cast")


def case_edge_convs(dev, B=2, H=6, W=5, C0=64):
    x = rnd(B, 4, H, W, seed=1, dev=dev, dtype=torch.float32)
    w = rnd(C0, 4, 3, 3, seed=2, dev=dev, dtype=torch.float32, scale=0.2)
    bias = rnd(C0, seed=3, dev=dev, dtype=torch.float32)
    y = ops.conv_in_fwd(x, w, bias, C0)
    ref = F.conv2d(x.cpu(), w.cpu(), bias.cpu(), padding=1)
    close(y.view(B, H, W, C0).permute(0, 3, 1, 2), ref, 1e-2, 1e-2, "conv_in")
    xo = rnd(B, H * W, C0, seed=4, dev=dev)
    wo = rnd(4, C0, 3, 3, seed=5, dev=dev, dtype=torch.float32, scale=0.1)
    bo = rnd(4, seed=6, dev=dev, dtype=torch.float32)
    xr = xo.float().cpu().view(B, H, W, C0).permute(0, 3, 1, 2).requires_grad_(True)
    refo = F.conv2d(xr, wo.cpu(), bo.cpu(), padding=1)
    yo = ops.conv_out_fwd(xo, wo, bo, B, H, W)
    close(yo, refo.detach(), 1e-4, 1e-4, "conv_out")
    dy = rnd(B, 4, H, W, seed=7, dev=dev, dtype=torch.float32)
    refo.backward(dy.cpu())
    dx = ops.conv_out_bwd(dy, wo, C0)
    close(dx.view(B, H, W, C0).permute(0, 3, 1, 2), xr.grad, 1e-2, 1e-2, "conv_out bwd")


def case_timestep_embedding(dev):
    from oracle.unet_sd15 import timestep_embedding
    t = torch.tensor([0, 19, 259, 999, 500], dtype=torch.int64, device=dev)
    out = ops.timestep_embedding(t, 320)
    ref = timestep_embedding(t.cpu(), 320)
    close(out, ref, 0, 8e-3, "timestep embedding")  # bf16 output, |v|<=1


def case_pcm_math(dev, golden):
    """Reference-owned math: the fused HIP kernels against fixtures produced by EXECUTING THE
    REFERENCE'S OWN SOURCE in the build container (tests/golden/make_golden.py).  Bit-exact."""
    g = golden
    acp = g["alphas_cumprod"].to(dev)
    acp_prev = g["ddim_alpha_cumprods_prev"].to(dev)
    assert acp_prev.dtype == torch.float64          # the reference's table is float64
    t_prev = g["ddim_timesteps_prev"].to(dev)
    idx, x, eps, noise = [g[k].to(dev) for k in ("index", "x", "eps", "noise")]
    start = g["start_timesteps"].to(dev)
    assert torch.equal(ops.add_noise(x, noise, acp, start).cpu(), g["add_noise_fp32"])
    import numpy as np
    for m in (1, 2, 4, 8):
        edges = torch.from_numpy(np.floor(np.linspace(0, 50, num=m, endpoint=False)).astype(np.int64)).to(dev)
        out, coef, end_t = ops.phase_jump(eps, x, start, idx, acp, acp_prev, t_prev, edges, target_mode=False)
        assert torch.equal(end_t.cpu(), g[f"multiphase_{m}_t"])
        assert torch.equal(out.cpu(), g[f"chain_online_{m}"]), m
        close(coef, g[f"chain_coef_{m}"], 1e-5, 1e-6, "coef")
        outt, _, _ = ops.phase_jump(eps, x.double(), start, idx, acp, acp_prev, t_prev, edges, target_mode=True)
        assert torch.equal(outt.cpu(), g[f"chain_target_{m}"]), m
    xp, xp32 = ops.cfg_ddim_step(eps, g["cfg_eps_u"].to(dev), x, start, idx, g["cfg_w"].to(dev), acp, acp_prev)
    assert torch.equal(xp.cpu(), g["cfg_x_prev"]) and torch.equal(xp32.cpu(), g["cfg_x_prev"].float())
    coef1 = torch.full((idx.shape[0],), 0.5, device=dev)
    for lt in ("huber", "l2"):
        loss, d = ops.consistency_loss(x, eps, coef1, lt == "huber", 0.001)
        ref = float(g[f"loss_{lt}"])
        assert abs(loss.item() - ref) <= 2e-6 * abs(ref), (lt, loss.item(), ref)
        close(d, g[f"loss_{lt}_grad"] * 0.5, 1e-5, 1e-9, "loss grad")


def case_optim(dev):
    from oracle.pcm_step import StepConfig, adamw_step, clip_grad_norm_
    n = 5000
    p = rnd(n, seed=1, dev=dev, dtype=torch.float32)
    cfg = StepConfig(lr=1e-2)
    pr = p.cpu().clone()
    m = torch.zeros_like(p); v = torch.zeros_like(p)
    state = {}
    for step in (1, 2, 3):
        g = rnd(n, seed=10 + step, dev=dev, dtype=torch.float32, scale=0.05 * step)
        gs = ops.sumsq(g)
        assert abs(gs.item() - float((g.double().cpu() ** 2).sum())) < 1e-9 * gs.item() + 1e-12
        ops.adamw_clip_step(p, g, m, v, gs, cfg.max_grad_norm, cfg.lr, cfg.adam_beta1, cfg.adam_beta2,
                            cfg.adam_epsilon, cfg.adam_weight_decay, step)
        gr = [g.cpu().clone()]
        clip_grad_norm_(gr, cfg.max_grad_norm)
        adamw_step([pr], gr, state, step, cfg)
        close(p, pr, 1e-5, 1e-6, f"adamw step {step}")
    t = rnd(100, seed=2, dev=dev, dtype=torch.float32)
    s_ = rnd(100, seed=3, dev=dev, dtype=torch.float32)
    ref = t.cpu() * 0.99 + s_.cpu() * 0.01
    ops.ema_update(t, s_, 0.99)
    close(t, ref, 1e-6, 1e-7, "ema")


def case_pack(dev):
    w = rnd(70, 40, seed=1, dev=dev, dtype=torch.float32)
    nk, kn = ops.pack_linear(w, scale=0.125)
    ref = (w.cpu() * 0.125).to(ops.BF16)
    assert torch.equal(nk.cpu(), ref) and torch.equal(kn.cpu(), ref.T.contiguous())
    wc = rnd(16, 24, 3, 3, seed=2, dev=dev, dtype=torch.float32)
    f, d = ops.pack_conv3x3(wc)
    rb = wc.cpu().to(ops.BF16)
    assert torch.equal(f.cpu(), rb.permute(0, 2, 3, 1).reshape(16, -1))
    assert torch.equal(d.cpu(), rb.flip(2, 3).permute(1, 2, 3, 0).reshape(24, -1))
    f2, d2 = ops.pack_conv3x3(wc.permute(0, 2, 3, 1).contiguous(), khwc=True)
    assert torch.equal(f2.cpu(), f.cpu()) and torch.equal(d2.cpu(), d.cpu())
    # the [N][tap][C] source goes through a 32 x 32 tile transpose: ragged tiles in both directions, with a scale
    wr = rnd(70, 3, 3, 40, seed=3, dev=dev, dtype=torch.float32)
    f3, d3 = ops.pack_conv3x3(wr, khwc=True, scale=0.5)
    rb = (wr.cpu() * 0.5).to(ops.BF16).permute(0, 3, 1, 2)          # [N][C][3][3]
    assert torch.equal(f3.cpu(), rb.permute(0, 2, 3, 1).reshape(70, -1))
    assert torch.equal(d3.cpu(), rb.flip(2, 3).permute(1, 2, 3, 0).reshape(40, -1))


def case_wgrad_plain(dev, M, N, K):
    dy = rnd(M, N, seed=1, dev=dev)
    t = rnd(M, 64, seed=2, dev=dev)
    x = rnd(M, K, seed=3, dev=dev)
    u = rnd(M, 64, seed=4, dev=dev)
    dB = torch.zeros(N, 64, dtype=torch.float32, device=dev)
    ops.lora_wgrad(dy, t, dB, 0.125, M, g_stride=64, r_stride=1)
    close(dB, 0.125 * dy.float().cpu().T @ t.float().cpu(), 2e-3, 2e-3 * M ** 0.5, "dB")
    dA = torch.zeros(64, K, dtype=torch.float32, device=dev)
    ops.lora_wgrad(x, u, dA, 0.125, M, g_stride=1, r_stride=K)
    close(dA, 0.125 * u.float().cpu().T @ x.float().cpu(), 2e-3, 2e-3 * M ** 0.5, "dA")


def case_wgrad_multi(dev):
    """Ten independent weight gradients through ONE ops.wgrad_batch (pcm_lora_wgrad_multi_bf16): both output orientations, ragged M,
    column slices of wider rows, different sizes in one launch, more jobs than one launch holds (8), and a 3x3-view job that the
    multi-launch kernel does not take (it must run on its own inside the same call)."""
    jobs, refs = [], []
    with ops.wgrad_batch():
        for i, (M, G) in enumerate([(300, 192), (64, 64), (1000, 72), (130, 320), (77, 136), (512, 640), (200, 200), (333, 64), (90, 128)]):
            big = rnd(M, G + 16, seed=10 + i, dev=dev)[:, 8:8 + G]            # a column slice: ldb != G
            small = rnd(M, 64, seed=30 + i, dev=dev)
            swap = i % 2 == 0
            out = torch.zeros((64, G) if swap else (G, 64), dtype=torch.float32, device=dev)
            ops.lora_wgrad(big, small, out, 0.5, M, G=G, g_stride=1 if swap else 64, r_stride=G if swap else 1, ldb=G + 16)
            ref = 0.5 * small.float().cpu().T @ big.float().cpu()
            jobs.append(out); refs.append((ref if swap else ref.T, M))
        B, Hs, C = 2, 8, 64
        x = rnd(B, Hs, Hs, C, seed=50, dev=dev); u = rnd(B * Hs * Hs, 64, seed=51, dev=dev)
        A = torch.zeros(64, C, 3, 3, requires_grad=True)
        F.conv2d(x.float().cpu().permute(0, 3, 1, 2), A, None, padding=1).backward(u.float().cpu().view(B, Hs, Hs, 64).permute(0, 3, 1, 2))
        dA = torch.zeros(64, 3, 3, C, dtype=torch.float32, device=dev)
        ops.lora_wgrad(x, u, dA, 1.0, B * Hs * Hs, conv=dict(Hs=Hs, Ws=Hs, Ho=Hs, Wo=Hs), g_stride=1, r_stride=9 * C)
        assert float(dA.abs().max()) == 0.0 and all(float(o.abs().max()) == 0.0 for o in jobs)   # nothing runs before the batch closes
    for k, (o, (ref, M)) in enumerate(zip(jobs, refs)):
        close(o, ref, 2e-3, 2e-3 * M ** 0.5, "multi job %d" % k)
    close(dA, A.grad.permute(0, 2, 3, 1), 2e-3, 2e-3 * (B * Hs * Hs) ** 0.5, "conv job inside the batch")


def case_wgrad_conv(dev, B, Hs, Ws, C, stride, src_mode=0):
    x = rnd(B, Hs, Ws, C, seed=1, dev=dev)
    xn = x.float().cpu().permute(0, 3, 1, 2)
    if src_mode == capi.SRC_UPSAMPLE2:
        xv = F.interpolate(xn, scale_factor=2.0, mode="nearest")
    else:
        xv = xn
    A = torch.zeros(64, C, 3, 3, requires_grad=True)
    y = F.conv2d(xv, A, None, stride=stride, padding=1)
    Ho, Wo = y.shape[2], y.shape[3]
    M = B * Ho * Wo
    u = rnd(M, 64, seed=2, dev=dev)
    y.backward(u.float().cpu().view(B, Ho, Wo, 64).permute(0, 3, 1, 2))
    dA = torch.zeros(64, C, 3, 3, dtype=torch.float32, device=dev)
    ops.lora_wgrad(x, u, dA, 1.0, M, conv=dict(Hs=Hs, Ws=Ws, Ho=Ho, Wo=Wo, stride=stride, src_mode=src_mode), out_conv=True)
    close(dA, A.grad, 2e-3, 2e-3 * M ** 0.5, "dA conv (peft layout)")
    dA2 = torch.zeros(64, 3, 3, C, dtype=torch.float32, device=dev)   # internal [r][kh][kw][ci] layout, coalesced atomics
    ops.lora_wgrad(x, u, dA2, 1.0, M, conv=dict(Hs=Hs, Ws=Ws, Ho=Ho, Wo=Wo, stride=stride, src_mode=src_mode), g_stride=1, r_stride=9 * C)
    close(dA2.permute(0, 3, 1, 2), A.grad, 2e-3, 2e-3 * M ** 0.5, "dA conv (khwc layout)")


def case_wgrad_dense(dev, B, H, W, Cin, Cout, alpha=1.0, prefill=True):
    """pcm_conv3x3_wgrad_bf16 (csrc/wgrad_dense.hip) against autograd of F.conv2d: accumulates into a pre-filled dW in the library's
    [co][kh][kw][ci] layout."""
    x = rnd(B, H, W, Cin, seed=1, dev=dev)
    dy = rnd(B * H * W, Cout, seed=2, dev=dev)
    Wt = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    F.conv2d(x.float().cpu().permute(0, 3, 1, 2), Wt, None, padding=1).backward(dy.float().cpu().view(B, H, W, Cout).permute(0, 3, 1, 2))
    ref = Wt.grad.permute(0, 2, 3, 1) * alpha
    pre = (torch.arange(Cout * 9 * Cin, dtype=torch.float32).view(Cout, 3, 3, Cin) % 7 - 3) * 0.25 if prefill else torch.zeros(Cout, 3, 3, Cin)
    dW = pre.clone().to(dev)
    assert ops.conv3x3_wgrad_ok(H, W, Cin, Cout)
    ops.conv3x3_wgrad(x, dy, dW, B, H, W, alpha)
    M = B * H * W
    close(dW.cpu() - pre, ref, 2e-3, 2e-3 * M ** 0.5 * alpha, "dense conv3x3 wgrad %s" % ((B, H, W, Cin, Cout),))


def case_reproducible_reductions(dev, big=False):
    """The reproducible forms (include/pcm_hip.h abi 4; ops.set_deterministic): every cross-workgroup sum through per-workgroup partials +
    an ordered finalize.  (1) the existing parity cases hold with the switch on, (2) the results agree with the atomic forms to summation
    rounding, (3) accumulation into a PRE-FILLED gradient buffer (the finalize adds, it does not overwrite), (4) two runs are BITWISE equal
    (on the GPU the atomic forms are not: that is what the switch is for)."""
    from pcm_amd.model import BF16  # noqa: F401
    sizes = dict(M=40000, N=640, K=320, B=4, H=32, C=128) if big else dict(M=700, N=200, K=136, B=2, H=8, C=64)
    M, N, Kk, B, Hs, C = (sizes[k] for k in ("M", "N", "K", "B", "H", "C"))

    def run_all():
        out = {}
        dy, t, x, u = rnd(M, N, seed=1, dev=dev), rnd(M, 64, seed=2, dev=dev), rnd(M, Kk, seed=3, dev=dev), rnd(M, 64, seed=4, dev=dev)
        dB = torch.full((N, 64), 0.25, dtype=torch.float32, device=dev)
        dA = torch.full((64, Kk), -0.5, dtype=torch.float32, device=dev)
        with ops.wgrad_batch():
            ops.lora_wgrad(dy, t, dB, 0.125, M, g_stride=64, r_stride=1)
            ops.lora_wgrad(x, u, dA, 0.125, M, g_stride=1, r_stride=Kk)
        out["dB"], out["dA"] = dB, dA
        xc, uc = rnd(B, Hs, Hs, C, seed=5, dev=dev), rnd(B * Hs * Hs, 64, seed=6, dev=dev)
        geo = dict(Hs=Hs, Ws=Hs, Ho=Hs, Wo=Hs)
        dAc = torch.full((64, 3, 3, C), 1.0, dtype=torch.float32, device=dev)
        ops.lora_wgrad(xc, uc, dAc, 1.0, B * Hs * Hs, conv=geo, g_stride=1, r_stride=9 * C)           # transpose-read 3x3 kernel (M >= 4096) or wgrad.hip
        dAp = torch.zeros(64, C, 3, 3, dtype=torch.float32, device=dev)
        ops.lora_wgrad(xc, uc, dAp, 1.0, B * Hs * Hs, conv=geo, out_conv=True)                         # peft layout: register-transposing kernel
        out["dA_conv"], out["dA_conv_peft"] = dAc, dAp
        out["colsum"] = ops.colsum(rnd(B, Hs * Hs, 4 * C, seed=7, dev=dev))
        g = rnd(300001 if big else 5001, seed=8, dev=dev, dtype=torch.float32)
        out["sumsq"] = ops.sumsq(g).clone()
        mp, tg = rnd(B, 4, Hs, Hs, seed=9, dev=dev, dtype=torch.float32), rnd(B, 4, Hs, Hs, seed=10, dev=dev, dtype=torch.float32)
        coef = torch.ones(B, dtype=torch.float32, device=dev)
        loss, d_eps = ops.consistency_loss(mp, tg, coef, True, 1e-3)
        out["loss"], out["d_eps"] = loss.clone(), d_eps
        xg = rnd(B, Hs * Hs, 4 * C, seed=11, dev=dev)
        gam, bet = rnd(4 * C, seed=12, dtype=torch.float32, dev=dev), rnd(4 * C, seed=13, dtype=torch.float32, dev=dev)
        y, st = ops.groupnorm_fwd(xg, gam, bet, 32, 1e-5, 1)
        out["gn_y"], out["gn_stats"] = y, st.clone()
        out["gn_dx"] = ops.groupnorm_bwd(xg, rnd(B, Hs * Hs, 4 * C, seed=14, dev=dev), st, gam, bet, 32, 1e-5, 1)
        # round 6 (abi 5): the discriminator heads' parameter gradients (discriminator_sd15.py:348-434) and the MMDiT modulation gradients
        dgn = rnd(B, Hs * Hs, 4 * C, seed=15, dev=dev)
        dgam, dbet = torch.full((4 * C,), 0.5, dtype=torch.float32, device=dev), torch.full((4 * C,), -0.25, dtype=torch.float32, device=dev)
        ops.groupnorm_param_grad(xg, dgn, st, gam, bet, dgam, dbet, 32, 1e-5, capi.ACT_LEAKY)           # adds into pre-filled buffers
        out["gn_dgamma"], out["gn_dbeta"] = dgam, dbet
        Mr = B * Hs * Hs
        xr, wr = rnd(Mr, 4 * C, seed=16, dev=dev), rnd(4 * C, seed=17, dtype=torch.float32, dev=dev)
        dyr = rnd(Mr, seed=18, dtype=torch.float32, dev=dev)
        dw, db = torch.full((4 * C,), 2.0, dtype=torch.float32, device=dev), torch.full((1,), 3.0, dtype=torch.float32, device=dev)
        out["rowdot_dx"] = ops.rowdot_bwd(xr, wr, dyr, dw, db)
        out["rowdot_dw"], out["rowdot_db"] = dw, db
        mean, rstd = rnd(Mr, seed=19, dtype=torch.float32, dev=dev), rnd(Mr, seed=20, dtype=torch.float32, dev=dev).abs() + 0.5
        out["mod_a"], out["mod_b"] = ops.mod_grad(xg, dgn, B, mean, rstd)
        out["mod_gate"], _ = ops.mod_grad(xg, dgn, B, want_b=False)
        lsum = torch.zeros(1, dtype=torch.float64, device=dev)
        fk, rl = rnd(Mr, seed=21, dtype=torch.float32, dev=dev), rnd(Mr, seed=22, dtype=torch.float32, dev=dev)
        out["hinge_df"], out["hinge_dr"] = ops.hinge_loss(fk, rl, 0, 0.25, lsum)
        ops.hinge_loss(fk, None, 1, 0.25, lsum)
        out["hinge_loss"] = lsum.clone()
        bias = torch.empty(4 * C, dtype=torch.float32, device=dev)
        ops.colsum_into(xr, bias, Mr, 4 * C)
        out["bias_sum"] = bias
        return {k: v.detach().cpu().clone() for k, v in out.items()}

    assert not ops.DETERMINISTIC
    fast = run_all()
    ops.set_deterministic(True)
    try:
        det1, det2 = run_all(), run_all()
        case_wgrad_plain(dev, 333, 200, 136)
        case_wgrad_multi(dev)
        case_wgrad_conv(dev, 2, 6, 5, 64, 2)
        case_optim(dev)
    finally:
        ops.set_deterministic(False)
    for k in fast:
        assert torch.equal(det1[k], det2[k]), "reproducible form of %s differs between two runs" % k
        a, b = det1[k].double(), fast[k].double()
        tol = 1e-12 if a.dtype == torch.float64 and k in ("sumsq", "loss", "gn_stats", "hinge_loss") else 2e-5
        assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max())) + (1e-2 if k in ("gn_y", "gn_dx") else 0.0), \
            (k, float((a - b).abs().max()), float(b.abs().max()))


def case_attention(dev, B, H, Lq, Lk, d, spike=False, spike_at=None, prescaled=False, spike_overflow=False, spike_gain=40):
    """``prescaled``: the pre-scaled-query kernels (csrc/attention_ps.hip): q carries d^-1/2 * log2(e) (rounded ONCE to the 16-bit
    format, as the folded to_q weights deliver it), the reference is the softmax of that q' in base 2, and dq is the gradient with
    respect to q'.  ``spike_overflow``: a late key row so far above the first tile's maximum that exp2 against the first tile's
    reference overflows -> the forward's one-time check must catch it and repeat the workgroup with maximum tracking.  ``spike_gain``: 40 puts
    that key ~250 above the first tile's maximum in the log2 domain (fp32 overflow); 4 .. 16 put it ~25 .. 100 above: finite in fp32 but beyond
    the IEEE-half range of the packed P (65504 = 2^16), which the half build must catch as well."""
    q = rnd(B, Lq, H * d, seed=1, dev=dev)
    k = rnd(B, Lk, H * d, seed=2, dev=dev)
    v = rnd(B, Lk, H * d, seed=3, dev=dev)
    if spike:  # force a large running-max jump late in the key stream (online-softmax rescale path)
        k[:, Lk - 3, :] = k[:, Lk - 3, :] * 6
    for pos in (spike_at or ()):   # ... and in the middle of the stream: the pipelined forward rescales O one tile after the move
        k[:, pos, :] = k[:, pos, :] * 5
    if spike_overflow:
        assert Lk > 70
        k[:, Lk - 5, :] = q[:, 3, :] * spike_gain       # query row 3 of every head: score ~ 40 |q|^2 / sqrt(d) ~ 250 in the log2 domain at gain 40
    dO = rnd(B, Lq, H * d, seed=4, dev=dev)
    if prescaled:
        q = (q.float() * ops.attn_q_scale(d)).to(q.dtype)
    qr, kr, vr = [t.float().cpu().requires_grad_(True) for t in (q, k, v)]

    def heads(t, L):
        return t.view(B, L, H, d).transpose(1, 2)
    s = heads(qr, Lq) @ heads(kr, Lk).transpose(-1, -2) * (0.6931471805599453 if prescaled else d ** -0.5)
    ref = (torch.softmax(s, -1) @ heads(vr, Lk)).transpose(1, 2).reshape(B, Lq, H * d)
    o, lse = ops.attn_fwd(q, k, v, H, d, prescaled=prescaled)
    # Relative bounds: at L=4096 the output rms is ~0.026, so an absolute 2e-2 would accept zeros.  bf16 output rounding alone is
    # 2^-9 = 2e-3 per element; P is rounded to bf16 before the PV MFMA (another ~2e-3, averaged down over the keys).
    rel_l2(o, ref.detach(), 5e-3, "attn fwd")
    close(o, ref.detach(), 1e-2, 1e-2 * float(ref.detach().abs().max()), "attn fwd (max-abs, scaled by |ref|max)")
    ref_lse = torch.logsumexp(s.detach(), -1) * 1.4426950408889634
    close(lse, ref_lse, 1e-3, 2e-2, "lse")
    ref.backward(dO.float().cpu())
    dq, dk, dv = ops.attn_bwd(q, k, v, o, dO, lse, H, d, prescaled=prescaled)
    for name, got, want in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
        # (several x5 / x6 key spikes make the softmax nearly one-hot: dS = P (dP - delta) is then a difference of nearly equal bf16-rounded
        # terms -- those stress cases get 2.5e-2)
        rel_l2(got, want, 2.5e-2 if (spike_at or spike_overflow) else 1.5e-2, name)
        close(got, want, 3e-2, 3e-2 * float(want.abs().max()), name + " (max-abs, scaled by |ref|max)")


def case_lora_repack(dev):
    """The one-launch segmented pack must equal the per-tensor packers."""
    from pcm_amd.model import LoraState
    from pcm_amd.unet_spec import UNetConfig
    cfg = UNetConfig(block_out_channels=(64, 128), layers_per_block=1, cross_attention_dim=64, heads=2)
    lora = LoraState(cfg, 64, 8.0, dev, seed=3, b_std=0.05)
    for m in lora.modules.values():
        if m.kind == "conv3":
            f, d = ops.pack_conv3x3(m.A, khwc=True)
        else:
            f, d = ops.pack_linear(m.A.view(64, m.K))
        bf, bd = ops.pack_linear(m.B.view(m.N, 64), scale=lora.scaling * lora.q_scale.get(m.path, 1.0))    # (to_q: the folded attention scale)
        assert torch.equal(m.A_fwd.cpu(), f.cpu()) and torch.equal(m.A_bwd.cpu(), d.cpu().view_as(m.A_bwd)), m.path
        assert torch.equal(m.Bs_fwd.cpu(), bf.cpu()) and torch.equal(m.Bs_bwd.cpu(), bd.cpu()), m.path


def case_adv_kernels(dev, golden):
    g = golden
    # noise_travel: bit-exact vs the fixture from the reference's own scheduler source
    acp = g["alphas_cumprod"].to(dev)
    out, sr = ops.noise_travel(g["x"].to(dev), g["noise"].to(dev), acp, g["start_timesteps"].to(dev), g["travel_target_t"].to(dev))
    assert torch.equal(out.cpu(), g["noise_travel"])
    r = g["alphas_cumprod"][g["travel_target_t"]] / g["alphas_cumprod"][g["start_timesteps"]]
    close(sr, r ** 0.5, 1e-6, 1e-7, "sqrt_r")
    # hinge losses vs the fixture (reference Discriminator.d_loss / g_loss, 3 heads)
    loss = torch.zeros(1, dtype=torch.float64, device=dev)
    dfs = []
    for i in range(3):
        df, dr = ops.hinge_loss(g[f"hinge_fake_{i}"].to(dev).reshape(-1), g[f"hinge_real_{i}"].to(dev).reshape(-1), 0, 1.0 / 3, loss)
        dfs.append((df, dr))
    assert abs(loss.item() - float(g["hinge_d_loss"])) < 1e-6
    f0 = g["hinge_fake_0"].clone().requires_grad_(True)
    r0 = g["hinge_real_0"].clone().requires_grad_(True)
    ((torch.relu(f0 + 1).mean() + torch.relu(1 - r0).mean()) / 3).backward()
    close(dfs[0][0], f0.grad.reshape(-1), 1e-6, 1e-9, "d_fake")
    close(dfs[0][1], r0.grad.reshape(-1), 1e-6, 1e-9, "d_real")
    loss.zero_()
    for i in range(3):
        ops.hinge_loss(g[f"hinge_fake_{i}"].to(dev).reshape(-1), None, 1, 1.0 / 3, loss)
    assert abs(loss.item() - float(g["hinge_g_loss"])) < 1e-6
    # GroupNorm + LeakyReLU fwd / bwd / param grads
    B, HW, C, G = 2, 40, 64, 32
    x = rnd(B, HW, C, seed=1, dev=dev, shift=0.2)
    gamma = rnd(C, seed=2, dev=dev, dtype=torch.float32, shift=1.0, scale=0.2)
    beta = rnd(C, seed=3, dev=dev, dtype=torch.float32, scale=0.2)
    dy = rnd(B, HW, C, seed=4, dev=dev)
    y, stats = ops.groupnorm_fwd(x, gamma, beta, G, 1e-5, capi.ACT_LEAKY)
    xr = x.float().cpu().permute(0, 2, 1).requires_grad_(True)
    gr, br = gamma.cpu().clone().requires_grad_(True), beta.cpu().clone().requires_grad_(True)
    ref = F.leaky_relu(F.group_norm(xr, G, gr, br, 1e-5), 0.01)
    close(y.permute(0, 2, 1), ref.detach(), 1e-2, 1e-2, "gn leaky fwd")
    ref.backward(dy.float().cpu().permute(0, 2, 1))
    dx = ops.groupnorm_bwd(x, dy, stats, gamma, beta, G, 1e-5, capi.ACT_LEAKY)
    close(dx.permute(0, 2, 1), xr.grad, 1e-2, 2e-2, "gn leaky bwd")
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ops.groupnorm_param_grad(x, dy, stats, gamma, beta, dg, db, G, 1e-5, capi.ACT_LEAKY)
    close(dg, gr.grad, 2e-3, 2e-2, "dgamma")
    close(db, br.grad, 2e-3, 2e-2, "dbeta")
    # rowdot
    M, C = 77, 64
    xx = rnd(M, C, seed=5, dev=dev)
    w = rnd(C, seed=6, dev=dev, dtype=torch.float32, scale=0.2)
    bias = rnd(1, seed=7, dev=dev, dtype=torch.float32)
    o = ops.rowdot_fwd(xx, w, bias)
    close(o, xx.float().cpu() @ w.cpu() + bias.cpu(), 1e-5, 1e-5, "rowdot")
    d = rnd(M, seed=8, dev=dev, dtype=torch.float32)
    dw, dbb = torch.zeros(C, device=dev), torch.zeros(1, device=dev)
    dxx = ops.rowdot_bwd(xx, w, d, dw, dbb)
    close(dxx, d.cpu()[:, None] * w.cpu()[None, :], 1e-2, 1e-2, "rowdot dx")
    close(dw, (d.cpu()[:, None] * xx.float().cpu()).sum(0), 1e-4, 1e-4, "rowdot dw")
    close(dbb, d.cpu().sum().reshape(1), 1e-5, 1e-5, "rowdot db")
    # the discriminator's real widths: 40 column vectors x 6 row lanes per block (block-level LDS reduction before the atomics), many blocks
    B, HW, C, G = 2, 300, 320, 32
    x = rnd(B, HW, C, seed=11, dev=dev, shift=0.2)
    gamma = rnd(C, seed=12, dev=dev, dtype=torch.float32, shift=1.0, scale=0.2)
    beta = rnd(C, seed=13, dev=dev, dtype=torch.float32, scale=0.2)
    dy = rnd(B, HW, C, seed=14, dev=dev)
    y, stats = ops.groupnorm_fwd(x, gamma, beta, G, 1e-5, capi.ACT_LEAKY)
    xr = x.float().cpu().permute(0, 2, 1)
    gr, br = gamma.cpu().clone().requires_grad_(True), beta.cpu().clone().requires_grad_(True)
    F.leaky_relu(F.group_norm(xr, G, gr, br, 1e-5), 0.01).backward(dy.float().cpu().permute(0, 2, 1))
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ops.groupnorm_param_grad(x, dy, stats, gamma, beta, dg, db, G, 1e-5, capi.ACT_LEAKY)
    close(dg, gr.grad, 5e-3, 5e-2, "dgamma (C=320)")
    close(db, br.grad, 5e-3, 5e-2, "dbeta (C=320)")
    M = 1000
    xx = rnd(M, C, seed=15, dev=dev)
    w = rnd(C, seed=16, dev=dev, dtype=torch.float32, scale=0.2)
    d = rnd(M, seed=18, dev=dev, dtype=torch.float32)
    dw, dbb = torch.zeros(C, device=dev), torch.zeros(1, device=dev)
    dxx = ops.rowdot_bwd(xx, w, d, dw, dbb)
    close(dxx, d.cpu()[:, None] * w.cpu()[None, :], 1e-2, 1e-2, "rowdot dx (C=320)")
    close(dw, (d.cpu()[:, None] * xx.float().cpu()).sum(0), 1e-3, 1e-3, "rowdot dw (C=320)")
    close(dbb, d.cpu().sum().reshape(1), 1e-4, 1e-4, "rowdot db (C=320)")
    # narrow rows: C = 32 -> 64 row lanes per block (more than the 32 bias slots the first LDS layout reserved: out-of-bounds write, ASan-visible)
    M, C = 500, 32
    xx = rnd(M, C, seed=25, dev=dev)
    w = rnd(C, seed=26, dev=dev, dtype=torch.float32, scale=0.2)
    d = rnd(M, seed=28, dev=dev, dtype=torch.float32)
    dw, dbb = torch.zeros(C, device=dev), torch.zeros(1, device=dev)
    dxx = ops.rowdot_bwd(xx, w, d, dw, dbb)
    close(dxx, d.cpu()[:, None] * w.cpu()[None, :], 1e-2, 1e-2, "rowdot dx (C=32)")
    close(dw, (d.cpu()[:, None] * xx.float().cpu()).sum(0), 1e-3, 1e-3, "rowdot dw (C=32)")
    close(dbb, d.cpu().sum().reshape(1), 1e-4, 1e-4, "rowdot db (C=32)")


def case_discriminator_heads(dev, dims=(64, 128), hw=(6, 3), B=2, nh=2):
    """Discriminator heads forward / backward (parameter grads and feature grads) vs torch autograd of the
    oracle's restatement of DiscriminatorHead (pinned to the reference source by the golden fixture)."""
    from oracle import pcm_math as M
    from pcm_amd.discriminator import Discriminator
    disc = Discriminator(dims, num_h_per_head=nh, device=dev, seed=4)
    feats, feats_ref = [], []
    for i, (C, h) in enumerate(zip(dims, hw)):
        f = rnd(B, h * h, C, seed=20 + i, dev=dev)
        feats.append((f, h, h))
        feats_ref.append(f.float().cpu().view(B, h, h, C).permute(0, 3, 1, 2).requires_grad_(True))
    logits, tape = disc.forward(feats, save=True)
    sd = {k: v.cpu().clone().requires_grad_(True) for k, v in disc.state_dict().items()}
    ref_logits = []
    for k in range(len(dims)):
        for h in range(nh):
            ref_logits.append(M.discriminator_head(sd, feats_ref[k], prefix=f"heads.{k}.{h}."))
    d_logits, loss = [], 0.0
    for i, (lg, rl) in enumerate(zip(logits, ref_logits)):
        close(lg, rl.detach().permute(0, 2, 3, 1).reshape(-1), 3e-2, 3e-2, f"logit {i}")
        d = rnd(lg.numel(), seed=40 + i, dev=dev, dtype=torch.float32)
        d_logits.append(d)
        loss = loss + (rl.permute(0, 2, 3, 1).reshape(-1) * d.cpu()).sum()
    loss.backward()
    disc.grads.zero_()
    d_feats = disc.backward(d_logits, tape, param_grads=True, feature_grads=True)
    rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / (b.double().norm() + 1e-12))
    cnt = {}
    for k, hd in disc.heads:
        h = cnt.get(k, 0); cnt[k] = h + 1
        for n, t in hd.g.items():
            v = t
            if n in ("conv1.0.weight", "conv2.0.weight"):
                v = v.permute(0, 3, 1, 2)
            elif n == "conv_out.weight":
                v = v.reshape(1, hd.C, 1, 1)
            r = rel(v, sd[f"heads.{k}.{h}.{n}"].grad)
            assert r < 0.1, (k, h, n, r)
    for k, (C, h) in enumerate(zip(dims, hw)):
        r = rel(d_feats[k].view(B, h, h, C).permute(0, 3, 1, 2), feats_ref[k].grad)
        assert r < 0.1, ("d_feat", k, r)


def case_teacher_input_grad(dev):
    """Feature-tap forward (reference modified_forward) and the generator-step backward through the frozen teacher
    down to d sample, with random cotangents on all 9 features, vs the oracle's autograd."""
    from oracle import unet_sd15 as O
    from pcm_amd.model import UNet, UNetWeights
    from pcm_amd.unet_spec import UNetConfig
    kw = dict(block_out_channels=(64, 128, 128, 128), cross_attention_dim=64, heads=2, norm_num_groups=32)
    oc, pc = O.UNetConfig(**kw), UNetConfig(**kw)
    sd = O.init_state_dict(oc, 0)
    W = UNetWeights(pc, sd, dev)
    g = torch.Generator().manual_seed(3)
    B, Hh = 2, 8
    x = torch.randn(B, 4, Hh, Hh, generator=g); t = torch.tensor([100, 700]); ctx = torch.randn(B, 7, 64, generator=g)
    xr = x.clone().requires_grad_(True)
    feats_ref = O.unet_forward(oc, sd, xr, t, ctx, return_features=True)
    teacher = UNet(W, None)
    feats, tape = teacher.forward(x.to(dev), t.to(dev), ctx.to(dev), features=True, save=True)
    assert len(feats) == 9
    d_feats, loss = [], 0.0
    for k, ((f, H, Wd), fr) in enumerate(zip(feats, feats_ref)):
        # the 1x1-resolution mid block normalises 4 values per GroupNorm group: bf16 rounding noise is amplified there (2.0-3.6 % of the
        # feature's range over seeds and GEMM plans, against 0.6-2.4 % for the other eight taps)
        ft = 5e-2 if H * Wd == 1 else 3e-2
        close(f.float().view(B, H, Wd, -1).permute(0, 3, 1, 2), fr.detach(), ft, ft * float(fr.detach().abs().max()), f"feature {k}")
        d = torch.randn(fr.shape, generator=torch.Generator().manual_seed(50 + k))
        loss = loss + (fr * d).sum()
        d_feats.append(d.permute(0, 2, 3, 1).reshape(B, H * Wd, -1).to(ops.BF16).contiguous().to(dev))
    loss.backward()
    d_in = teacher.backward(None, tape, d_feats=d_feats, need_input_grad=True)
    r = float((d_in.cpu() - xr.grad).norm() / xr.grad.norm())
    assert r < 0.08, r


def case_gemm_big(dev, which):
    """The 256-row phased tile (gemm8p.hip) forced through pcm_debug_gemm_big_mode(2) on every contraction flavour
    the UNet uses; returns (max abs err, tolerance) against torch fp32 on the same bf16-rounded operands."""
    import torch.nn.functional as F
    from pcm_amd import capi, ops

    def rnd(*shape, seed=0, scale=1.0):
        g = torch.Generator().manual_seed(seed)
        return (torch.randn(*shape, generator=g) * scale).to(ops.BF16).to(dev)

    dll = capi.lib().dll
    dll.pcm_debug_gemm_big_mode(2)
    try:
        if which == "plain_lora":      # 2 M tiles of 256x320, second K segment, bias + residual
            M, N, K = 512, 320, 320
            x, w, t, bl = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(M, 64, seed=3), rnd(N, 64, seed=4, scale=0.1)
            bias = torch.randn(N, generator=torch.Generator().manual_seed(5)).to(dev)
            res = rnd(M, N, seed=6)
            out = torch.empty(M, N, dtype=ops.BF16, device=dev)
            ops.gemm([ops.Seg(x, w), ops.Seg(t, bl)], M, N, out, bias=bias, residual=res)
            ref = x.float() @ w.float().T + t.float() @ bl.float().T + bias + res.float()
        elif which == "ragged":         # M tail, 2 N tiles, N not a tile multiple, SiLU + alpha, single K tile
            M, N, K = 300, 448, 64
            x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.2)
            out = torch.empty(M, N, dtype=ops.BF16, device=dev)
            ops.gemm([ops.Seg(x, w)], M, N, out, act=capi.ACT_SILU, alpha=0.5)
            ref = F.silu(0.5 * (x.float() @ w.float().T))
        elif which == "two_tiles_k":    # exactly two K tiles (prologue-only pipeline), 256x256 tile
            M, N, K = 256, 256, 128
            x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.2)
            out = torch.empty(M, N, dtype=ops.BF16, device=dev)
            ops.gemm([ops.Seg(x, w)], M, N, out)
            ref = x.float() @ w.float().T
        elif which == "persist":        # more work items than resident blocks (8 in the emulator, 256 on the GPU): multi-item blocks
            M, N, K = (2304, 640, 128) if dev == "cpu" else (70000, 640, 320)
            x, w, t, bl = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(M, 64, seed=3), rnd(N, 64, seed=4, scale=0.1)
            bias = torch.randn(N, generator=torch.Generator().manual_seed(5)).to(dev)
            res = rnd(M, N, seed=6)
            out = torch.empty(M, N, dtype=ops.BF16, device=dev)
            ops.gemm([ops.Seg(x, w), ops.Seg(t, bl)], M, N, out, bias=bias, residual=res)
            ref = x.float() @ w.float().T + t.float() @ bl.float().T + bias + res.float()
        elif which == "persist_splitk":
            M, N, K = (1280, 320, 2048) if dev == "cpu" else (40000, 320, 2048)
            x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
            out = torch.empty(M, N, dtype=ops.BF16, device=dev)
            ops.gemm([ops.Seg(x, w)], M, N, out, act=capi.ACT_SILU)
            ref = F.silu(x.float() @ w.float().T)
        elif which == "splitk":
            M, N, K = 256, 320, 2048
            x, w, t, bl = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05), rnd(M, 64, seed=3), rnd(N, 64, seed=4, scale=0.1)
            bias = torch.randn(N, generator=torch.Generator().manual_seed(5)).to(dev)
            out = torch.empty(M, N, dtype=ops.BF16, device=dev)
            ops.gemm([ops.Seg(x, w), ops.Seg(t, bl)], M, N, out, bias=bias)
            ref = x.float() @ w.float().T + t.float() @ bl.float().T + bias
        else:                            # conv3x3 flavours, with the LoRA branch as a plain or a conv second segment
            stride, src_mode, lora = {"conv": (1, capi.SRC_DIRECT, "plain"), "conv_s2": (2, capi.SRC_DIRECT, None),
                                      "conv_up": (1, capi.SRC_UPSAMPLE2, "plain"), "conv_zi": (1, capi.SRC_ZEROINS2, None),
                                      "conv_conv": (1, capi.SRC_DIRECT, "conv"), "conv_splitk": (1, capi.SRC_DIRECT, "plain"),
                                      "conv_persist": (1, capi.SRC_DIRECT, "plain"), "conv_small_map": (1, capi.SRC_DIRECT, "conv")}[which]
            Hs = {"conv": 16, "conv_s2": 32, "conv_up": 8, "conv_zi": 8, "conv_conv": 16, "conv_splitk": 16, "conv_persist": 16, "conv_small_map": 8}[which]
            B, Ci, Co = (1, 256, 320) if which == "conv_splitk" else (2, 128 if which == "conv" else 64, 320)
            if which == "conv_small_map":      # 8x8 feature maps, five images per 256-row tile and a ragged last tile: the by-shape chunk-outer order
                B, Ci, Co = 5, 128, 320
            if which == "conv_persist":
                B, Ci, Co = (9, 64, 640) if dev == "cpu" else (300, 64, 640)
            x = rnd(B, Hs, Hs, Ci, seed=7)
            w = rnd(Co, Ci, 3, 3, seed=8, scale=0.05)
            xn = x.float().permute(0, 3, 1, 2)
            if src_mode == capi.SRC_UPSAMPLE2:
                xv = F.interpolate(xn, scale_factor=2.0, mode="nearest")
            elif src_mode == capi.SRC_ZEROINS2:
                xv = torch.zeros(B, Ci, 2 * Hs, 2 * Hs, device=dev)
                xv[:, :, ::2, ::2] = xn
            else:
                xv = xn
            ref = F.conv2d(xv, w.float(), None, stride=stride, padding=1)
            Ho = ref.shape[2]
            M = B * Ho * Ho
            segs = [ops.Seg(x, w.permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous(), conv=dict(Hs=Hs, Ws=Hs, stride=stride, src_mode=src_mode))]
            ref = ref.permute(0, 2, 3, 1).reshape(M, Co)
            if lora == "plain":
                t, bl = rnd(M, 64, seed=3), rnd(Co, 64, seed=4, scale=0.1)
                segs.append(ops.Seg(t, bl))
                ref = ref + t.float() @ bl.float().T
            elif lora == "conv":
                d = rnd(B, Hs, Hs, 64, seed=11)
                w2 = rnd(Co, 64, 3, 3, seed=12, scale=0.05)
                segs.append(ops.Seg(d, w2.permute(0, 2, 3, 1).reshape(Co, 576).contiguous(), conv=dict(Hs=Hs, Ws=Hs)))
                ref = ref + F.conv2d(d.float().permute(0, 3, 1, 2), w2.float(), None, padding=1).permute(0, 2, 3, 1).reshape(M, Co)
            temb = rnd(B, Co, seed=9)
            out = torch.empty(M, Co, dtype=ops.BF16, device=dev)
            ops.gemm(segs, M, Co, out, rowvec=temb, rows_per_batch=Ho * Ho, Ho=Ho, Wo=Ho)
            ref = ref + temb.float().repeat_interleave(Ho * Ho, 0)
        plan = dll.pcm_debug_last_gemm_plan()
        assert plan >= 4000, ("big tile not taken", which, plan)
        if which in ("splitk", "conv_splitk", "persist_splitk") and not (which == "persist_splitk" and dev != "cpu"):
            assert plan % 1000 > 1, plan
    finally:
        dll.pcm_debug_gemm_big_mode(1)
    err = (out.float() - ref).abs()
    tol = 2e-2 + 1e-2 * ref.abs()
    return float((err - tol).max()), float(err.max())


def case_gemm_epilogue_fusions(dev, which):
    """abi 5 (include/pcm_hip.h pcm_gemm_epi.out2 / chstats): the contraction's epilogue (a) writes a second copy of the output rows into a
    strided slot (a skip tensor's place in its concat buffer) and (b) accumulates per-(sample, channel) sum / sum-of-squares of the STORED
    values; pcm_groupnorm_apply_chstats then normalises from them without a statistics pass -- equal to the statistics pass + apply on the same
    tensor.  ``which``: the phased tile's own epilogue (plain / residual / rowvec-conv flavours, small feature maps that flush per pass, ragged
    N), the split-K finalize, and a plan that does not emit (the caller falls back)."""
    import torch.nn.functional as F
    from pcm_amd import capi, ops

    def rnd(*shape, seed=0, scale=1.0):
        g = torch.Generator().manual_seed(seed)
        return (torch.randn(*shape, generator=g) * scale).to(ops.BF16).to(dev)

    dll = capi.lib().dll
    dll.pcm_debug_gemm_big_mode(2 if which != "small_tile" else 1)
    try:
        if which in ("plain", "res", "small_maps", "ragged_n", "splitk", "small_tile"):
            B, HW, N, K = {"plain": (2, 256, 320, 128), "res": (3, 256, 320, 128), "small_maps": (10, 64, 320, 128), "ragged_n": (2, 256, 448, 64),
                           "splitk": (1, 256, 320, 2048), "small_tile": (2, 64, 128, 64)}[which]
            M = B * HW
            x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1)
            bias = torch.randn(N, generator=torch.Generator().manual_seed(5)).to(dev)
            res = rnd(M, N, seed=6) if which in ("res", "small_maps", "splitk") else None
            segs, kw = [ops.Seg(x, w)], dict(bias=bias, residual=res)
        else:                      # resnet conv1 flavour: 3x3 conv + bias + per-sample row vector (time embedding)
            B, Hs, Ci, N = 3, 16, 64, 320
            HW, M = Hs * Hs, B * Hs * Hs
            xi, w = rnd(B, Hs, Hs, Ci, seed=7), rnd(N, 9 * Ci, seed=8, scale=0.05)
            temb = rnd(B, N, seed=9)
            bias = torch.randn(N, generator=torch.Generator().manual_seed(5)).to(dev)
            segs, kw = [ops.Seg(xi, w, conv=dict(Hs=Hs, Ws=Hs))], dict(bias=bias, rowvec=temb, rows_per_batch=HW, Ho=Hs, Wo=Hs)
        ref_out = torch.empty(M, N, dtype=ops.BF16, device=dev)
        ops.gemm(segs, M, N, ref_out, **kw)                                   # the same call without the fusions
        arena = ops.ChStatArena(dev, B * N * 2 * 8)
        cs = arena.take(B, N, HW)
        cat = torch.full((M, N + 192), 3.0, dtype=ops.BF16, device=dev)       # the output's slot: columns [128, 128 + N) of a wider buffer
        out = torch.empty(M, N, dtype=ops.BF16, device=dev)
        ops.gemm(segs, M, N, out, out2=cat[:, 128:128 + N], ldo2=N + 192, chstats=cs, **kw)
        plan = dll.pcm_debug_last_gemm_plan()
        assert torch.equal(out.cpu(), ref_out.cpu()), "the fusions must not change the output"
        assert torch.equal(cat[:, 128:128 + N].cpu(), out.cpu()) and bool((cat[:, :128].float() == 3.0).all()) and bool((cat[:, 128 + N:].float() == 3.0).all()), "out2 slot"
        if which == "small_tile":
            assert cs.rows == 0 and plan < 1000, ("a 4-wave tile does not emit statistics: the caller falls back", plan)
            return
        assert cs.rows == HW, ("statistics not emitted", which, plan)
        assert (plan % 1000 > 1) == (which == "splitk"), plan
        o = out.float().cpu().double().view(B, HW, N)
        want = torch.stack([o.sum(1), (o * o).sum(1)], -1)                     # [B, N, 2]
        got = cs.buf.cpu()
        assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max()), (which, float((got - want).abs().max()), float(want.abs().max()))
        # GroupNorm from the per-channel sums == statistics pass + apply (same kernel body downstream of the group sums)
        G = 32
        gamma, beta = torch.randn(N, generator=torch.Generator().manual_seed(3)).to(dev), torch.randn(N, generator=torch.Generator().manual_seed(4)).to(dev)
        xo = out.view(B, HW, N)
        y_ref, st_ref = ops.groupnorm_fwd(xo, gamma, beta, G, 1e-5, capi.ACT_SILU)
        y, st = ops.groupnorm_fwd(xo, gamma, beta, G, 1e-5, capi.ACT_SILU, chstats=cs)
        assert float((st.cpu() - st_ref.cpu()).abs().max()) <= 2e-5 * float(st_ref.abs().max())
        close(y, y_ref.float().cpu(), 1e-2, 1e-2, "GroupNorm from epilogue statistics")
        # ... and as the right-hand part of a channel concatenation [other | out]
        Ca = 128
        other = rnd(B, HW, Ca, seed=21)
        oa = other.float().cpu().double()
        csa = ops.ChStats(torch.stack([oa.sum(1), (oa * oa).sum(1)], -1).to(dev).contiguous(), B, Ca, HW)
        xc = torch.cat([other, xo], -1).contiguous()
        g2, b2 = torch.randn(Ca + N, generator=torch.Generator().manual_seed(13)).to(dev), torch.randn(Ca + N, generator=torch.Generator().manual_seed(14)).to(dev)
        if (Ca + N) % G == 0:
            yc_ref, stc_ref = ops.groupnorm_fwd(xc, g2, b2, G, 1e-5, capi.ACT_SILU)
            yc, stc = ops.groupnorm_fwd(xc, g2, b2, G, 1e-5, capi.ACT_SILU, chstats=csa, chstats2=cs)
            assert float((stc.cpu() - stc_ref.cpu()).abs().max()) <= 2e-5 * float(stc_ref.abs().max())
            close(yc, yc_ref.float().cpu(), 1e-2, 1e-2, "GroupNorm of a concatenation from two producers' statistics")
    finally:
        dll.pcm_debug_gemm_big_mode(1)


def case_gemm_ws(dev, which, form=1):
    """The weights-stationary kernel (csrc/gemm_ws.hip, round 6): the short-K projections of the 64x64 level -- a 320-column weight slice held in
    registers per workgroup, 64-row activation tiles through a two-stage LDS-DMA ring.  Against torch fp32 on the same 16-bit operands AND bit for
    bit against the phased tile (pcm_debug_gemm_ws(0)): same products, same accumulation order, same epilogue operation order."""
    from pcm_amd import capi, ops

    def rnd(*shape, seed=0, scale=1.0):
        g = torch.Generator().manual_seed(seed)
        return (torch.randn(*shape, generator=g) * scale).to(ops.BF16).to(dev)

    M, N, lora, res, bias, strided = {"plain": (16384, 320, False, False, False, False), "lora_res": (16384, 320, True, True, True, False),
                                      "qkv": (16384, 960, False, False, False, False), "tail_strided": (16384 + 72, 320, True, True, True, True),
                                      "res_tail_strided": (16384 + 72, 320, False, True, True, True)}[which]
    K = 320
    xw = rnd(M, K + 64, seed=1) if strided else None            # the activation as a column slice of a wider matrix (row stride K + 64)
    x = xw[:, :K] if strided else rnd(M, K, seed=1)
    w = rnd(N, K, seed=2, scale=0.1)
    segs = [ops.Seg(x, w, lda=(K + 64) if strided else None)]
    ref = x.float().cpu() @ w.float().cpu().T
    if lora:
        t, bl = rnd(M, 64, seed=3), rnd(N, 64, seed=4, scale=0.1)
        segs.append(ops.Seg(t, bl))
        ref = ref + t.float().cpu() @ bl.float().cpu().T
    b = torch.randn(N, generator=torch.Generator().manual_seed(5)).to(dev) if bias else None
    r = rnd(M, N, seed=6) if res else None
    if b is not None:
        ref = ref + b.cpu()
    if r is not None:
        ref = ref + r.float().cpu()
    dll = capi.lib().dll
    outs = []
    for on in (form, 0):
        dll.pcm_debug_gemm_ws(on)
        try:
            ow = torch.full((M, N + (64 if strided else 0)), 5.0, dtype=ops.BF16, device=dev)
            ops.gemm(segs, M, N, ow[:, :N] if strided else ow, bias=b, residual=r, ldo=ow.shape[-1])
            plan = dll.pcm_debug_last_gemm_plan()
        finally:
            dll.pcm_debug_gemm_ws(-1)
        assert (plan // 1000 == 29 + on) == bool(on), (which, on, plan)       # 30000 + K/32: four waves x 80 columns; 31000 + K/32: eight waves x 40
        outs.append(ow)
    got = outs[0][:, :N].float().cpu()
    err = (got - ref).abs()
    assert float((err - (2e-2 + 1e-2 * ref.abs())).max()) <= 0, (which, float(err.max()))
    assert torch.equal(outs[0].cpu(), outs[1].cpu()), (which, "weights-stationary kernel and phased tile differ", float((outs[0].float() - outs[1].float()).abs().max()))
    if strided:
        assert bool((outs[0][:, N:].float() == 5.0).all()), "columns beyond N touched"


GEMM_WS_CASES = ["plain", "lora_res", "qkv", "tail_strided"]          # form 1 (four waves)
GEMM_WS8_CASES = ["plain", "qkv", "res_tail_strided"]                    # form 2 (eight waves; K = 320 only)


GEMM_EPI_FUSION_CASES = ["plain", "res", "small_maps", "ragged_n", "splitk", "conv_rowvec", "small_tile"]


GEMM_BIG_CASES = ["plain_lora", "ragged", "two_tiles_k", "splitk", "conv", "conv_s2", "conv_up", "conv_zi", "conv_conv", "conv_splitk", "persist", "persist_splitk", "conv_persist", "conv_small_map"]


def case_gemm_4w(dev, which):
    """The short-K kernel (gemm4w.hip: 128 x 320 / 128 x 256 tiles, two workgroups per CU) forced through pcm_debug_gemm_big_mode(3) on the
    plain-segment flavours it serves; returns (max abs excess over tolerance, max abs err) against torch fp32 on the same bf16 operands."""
    import torch.nn.functional as F
    from pcm_amd import capi, ops

    def rnd(*shape, seed=0, scale=1.0):
        g = torch.Generator().manual_seed(seed)
        return (torch.randn(*shape, generator=g) * scale).to(ops.BF16).to(dev)

    dll = capi.lib().dll
    dll.pcm_debug_gemm_big_mode(3)
    try:
        if which == "plain_lora":      # 4 M tiles of 128x320, 10 + 2 K-steps over two segments, bias + residual
            M, N, K = 512, 320, 320
            x, w, t, bl = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(M, 64, seed=3), rnd(N, 64, seed=4, scale=0.1)
            bias = torch.randn(N, generator=torch.Generator().manual_seed(5)).to(dev)
            res = rnd(M, N, seed=6)
            out = torch.empty(M, N, dtype=ops.BF16, device=dev)
            ops.gemm([ops.Seg(x, w), ops.Seg(t, bl)], M, N, out, bias=bias, residual=res)
            ref = x.float() @ w.float().T + t.float() @ bl.float().T + bias + res.float()
            fn = 5
        elif which == "ragged":         # M tail, 128x256 tile with an N tail, SiLU + alpha, exactly two K-steps (prologue-only pipeline)
            M, N, K = 300, 448, 64
            x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.2)
            out = torch.empty(M, N, dtype=ops.BF16, device=dev)
            ops.gemm([ops.Seg(x, w)], M, N, out, act=capi.ACT_SILU, alpha=0.5)
            ref = F.silu(0.5 * (x.float() @ w.float().T))
            fn = 4
        elif which == "qkv":            # fused q/k/v: N = 3 x 320, LoRA segment of K = 192, three K-steps ahead of a segment switch
            M, N, K = 256, 960, 128
            x, w, t, bl = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(M, 192, seed=3), rnd(N, 192, seed=4, scale=0.1)
            out = torch.empty(M, N, dtype=ops.BF16, device=dev)
            ops.gemm([ops.Seg(x, w), ops.Seg(t, bl)], M, N, out)
            ref = x.float() @ w.float().T + t.float() @ bl.float().T
            fn = 5
        elif which == "rowvec":         # per-image row vector (time embedding) + SiLU, strided activation rows (lda > K), strided output
            M, N, K = 384, 640, 192
            xa, w = rnd(M, K + 64, seed=1), rnd(N, K, seed=2, scale=0.1)
            temb = rnd(3, N, seed=9)
            outa = torch.full((M, N + 64), 3.0, dtype=ops.BF16, device=dev)
            ops.gemm([ops.Seg(xa, w, lda=K + 64)], M, N, outa, rowvec=temb, rows_per_batch=128, act=capi.ACT_SILU, ldo=N + 64)
            ref = F.silu(xa[:, :K].float() @ w.float().T + temb.float().repeat_interleave(128, 0))
            assert bool((outa[:, N:].float() == 3.0).all())
            out = outa[:, :N]
            fn = 5
        else:                            # "many": more tiles than the emulator keeps resident; odd K-step count
            M, N, K = (1300, 640, 160 + 32 * 0) if dev == "cpu" else (70000, 640, 320)
            K = 192 if dev == "cpu" else K
            x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1)
            res = rnd(M, N, seed=6)
            out = torch.empty(M, N, dtype=ops.BF16, device=dev)
            ops.gemm([ops.Seg(x, w)], M, N, out, residual=res)
            ref = x.float() @ w.float().T + res.float()
            fn = 5
        plan = dll.pcm_debug_last_gemm_plan()
        assert plan == 10000 + 1000 * fn + 1, ("gemm4w not taken", which, plan)
    finally:
        dll.pcm_debug_gemm_big_mode(1)
    err = (out.float() - ref).abs()
    tol = 2e-2 + 1e-2 * ref.abs()
    return float((err - tol).max()), float(err.max())


GEMM_4W_CASES = ["plain_lora", "ragged", "qkv", "rowvec", "many"]


def case_gemm_n64(dev, M, K):
    """Streaming rank-64 projection kernel (gemm_n64.hip) vs torch fp32; returns max abs excess over tolerance."""
    from pcm_amd import capi, ops
    g = torch.Generator().manual_seed(M + K)
    x = (torch.randn(M, K, generator=g)).to(ops.BF16).to(dev)
    w = (torch.randn(64, K, generator=g) * 0.1).to(ops.BF16).to(dev)
    out = torch.empty(M, 64, dtype=ops.BF16, device=dev)
    ops.gemm([ops.Seg(x, w)], M, 64, out)
    assert capi.lib().dll.pcm_debug_last_gemm_plan() == (32 if M <= 16 else 64)     # (batch-row calls, M <= 16: gemm_smallm.hip takes them first)
    ref = x.float() @ w.float().T
    err = (out.float() - ref).abs()
    return float((err - (2e-2 + 1e-2 * ref.abs())).max())


def case_gemm_smallm(dev, M, N, Ks, act=0, bias=True, out_f32=False, alpha=1.0):
    """batch-row projections (M <= 16: time-embedding MLP, time_emb_proj, adaLN modulation) through the weight-streaming kernel
    (gemm_smallm.hip, plan 32) vs torch fp32 on the same 16-bit operands AND vs the generic tile path (big_mode 0); returns the excess."""
    from pcm_amd import capi, ops
    g = torch.Generator().manual_seed(M * 7 + N)
    segs, ref = [], torch.zeros(M, N)
    for K in Ks:
        x = torch.randn(M, K, generator=g).to(ops.BF16).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.05).to(ops.BF16).to(dev)
        segs.append(ops.Seg(x, w))
        ref = ref + x.float().cpu() @ w.float().cpu().T
    b = torch.randn(N, generator=g).to(dev) if bias else None
    ref = alpha * ref + (b.cpu() if bias else 0.0)
    if act:
        ref = F.silu(ref)
    dll = capi.lib().dll
    outs = []
    for mode in (1, 0):
        dll.pcm_debug_gemm_big_mode(mode)
        try:
            out = torch.full((M, N + 8), 7.0, dtype=torch.float32 if out_f32 else ops.BF16, device=dev)     # row stride > N, sentinel in the gap
            ops.gemm(segs, M, N, out, bias=b, act=capi.ACT_SILU if act else capi.ACT_NONE, alpha=alpha, ldo=N + 8)
            assert (dll.pcm_debug_last_gemm_plan() == 32) == (mode == 1), (mode, dll.pcm_debug_last_gemm_plan())
        finally:
            dll.pcm_debug_gemm_big_mode(1)
        assert bool((out[:, N:].float() == 7.0).all()), "wrote past N"
        outs.append(out[:, :N].float().cpu())
    err = (outs[0] - ref).abs()
    tol = (2e-3 if out_f32 else 2e-2) + (2e-3 if out_f32 else 1e-2) * ref.abs()
    assert float((outs[0] - outs[1]).abs().max()) <= float((2 * tol).max()), "small-M kernel vs generic tile"
    return float((err - tol).max())


def case_conv_r64(dev, B, H, W, C, expect_kernel=True):
    """conv LoRA down-projection t = conv3x3(x, A), A: C -> 64 (N = 64 implicit GEMM): the halo-window kernel (conv_r64.hip) where the
    geometry allows it, the generic tile otherwise -- both against torch conv2d on the same bf16 operands."""
    import ctypes
    from pcm_amd import capi, ops
    cnt = capi.lib().dll.pcm_debug_conv_r64_count
    cnt.restype = ctypes.c_long
    x = rnd(B, H, W, C, seed=1, dev=dev)
    A = rnd(64, C, 3, 3, seed=2, dev=dev, scale=0.05)
    wk = A.permute(0, 2, 3, 1).reshape(64, 9 * C).contiguous()          # [r][(kh, kw, ci)]
    M = B * H * W
    ref = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), A.float().cpu(), None, padding=1).permute(0, 2, 3, 1).reshape(M, 64)
    n0 = cnt()
    out = torch.full((M, 64), 7.0, dtype=ops.BF16, device=dev)
    capi.lib().dll.pcm_debug_conv_r64(2)         # 2: take every shape the kernel can run (the product skips launches with < 256 patches)
    try:
        ops.gemm([ops.Seg(x, wk, conv=dict(Hs=H, Ws=W))], M, 64, out, Ho=H, Wo=W)
    finally:
        capi.lib().dll.pcm_debug_conv_r64(1)
    assert (cnt() == n0 + 1) == expect_kernel, (cnt() - n0, expect_kernel)
    close(out, ref, 1e-2, 1e-2 * float(ref.abs().max()), "conv r64 %dx%dx%dx%d" % (B, H, W, C))
    if expect_kernel:    # same numbers from the generic path
        capi.lib().dll.pcm_debug_conv_r64(0)
        try:
            out2 = torch.empty(M, 64, dtype=ops.BF16, device=dev)
            ops.gemm([ops.Seg(x, wk, conv=dict(Hs=H, Ws=W))], M, 64, out2, Ho=H, Wo=W)
        finally:
            capi.lib().dll.pcm_debug_conv_r64(1)
        close(out, out2.float().cpu(), 1e-2, 1e-2 * float(ref.abs().max()), "conv r64 vs generic")


def case_gemm_geglu(dev, M=300, K=128, inner=160, big_mode=1):
    """Projection with the GEGLU epilogue (interleaved value/gate rows) vs torch; returns max abs excess over tolerance."""
    import torch.nn.functional as F
    from pcm_amd import capi, ops
    capi.lib().dll.pcm_debug_gemm_big_mode(big_mode)      # 3: the gemm4w.hip epilogue instead of gemm8p's
    try:
        return _case_gemm_geglu(dev, M, K, inner, big_mode)
    finally:
        capi.lib().dll.pcm_debug_gemm_big_mode(1)


def _case_gemm_geglu(dev, M, K, inner, big_mode):
    import torch.nn.functional as F
    from pcm_amd import capi, ops
    g = torch.Generator().manual_seed(7)
    x = torch.randn(M, K, generator=g).to(ops.BF16).to(dev)
    w = (torch.randn(2 * inner, K, generator=g) * 0.1).to(ops.BF16).to(dev)
    bias = torch.randn(2 * inner, generator=g).to(dev)
    from pcm_amd.model import geglu_perm
    perm = geglu_perm(inner, dev)
    out = torch.empty(M, inner, dtype=ops.BF16, device=dev)
    ops.gemm([ops.Seg(x, w[perm].contiguous())], M, 2 * inner, out, bias=bias[perm].contiguous(), act=capi.ACT_GEGLU, ldo=inner)
    plan = capi.lib().dll.pcm_debug_last_gemm_plan()
    assert (plan >= 14000) if big_mode == 3 else ((4000 <= plan < 10000) if big_mode in (2, 4) else plan >= 4000), plan
    h = x.float() @ w.float().T + bias
    ref = h[:, :inner] * F.gelu(h[:, inner:])
    err = (out.float() - ref).abs()
    excess = float((err - (2e-2 + 1e-2 * ref.abs())).max())
    # second output: the interleaved pre-activation of the first `rows` rows (what the backward of GEGLU needs) -- plus a LoRA segment
    # with interleaved s*B rows; rows beyond `rows` must stay untouched; geglu_bwd_interleaved == geglu_bwd on the de-interleaved tensor
    rows = (M // 2 + 7) // 8 * 8
    t = torch.randn(M, 64, generator=g).to(ops.BF16).to(dev)
    bl = (torch.randn(2 * inner, 64, generator=g) * 0.05).to(ops.BF16).to(dev)
    pre = torch.full((rows + 8, 2 * inner), 7.0, dtype=ops.BF16, device=dev)
    out2 = torch.empty(M, inner, dtype=ops.BF16, device=dev)
    ops.gemm([ops.Seg(x, w[perm].contiguous()), ops.Seg(t, bl[perm].contiguous())], M, 2 * inner, out2, bias=bias[perm].contiguous(),
             act=capi.ACT_GEGLU, ldo=inner, pre_out=pre[:rows])
    h2 = h + t.float() @ bl.float().T
    ref2 = h2[:, :inner] * F.gelu(h2[:, inner:])
    excess = max(excess, float(((out2.float() - ref2).abs() - (2e-2 + 1e-2 * ref2.abs())).max()))
    assert bool((pre[rows:] == 7.0).all()), "rows >= pre_rows were written"
    pre_std = torch.empty(rows, 2 * inner, dtype=ops.BF16, device=dev)
    pre_std[:, perm] = pre[:rows]
    excess = max(excess, float(((pre_std.float() - h2[:rows]).abs() - (2e-2 + 1e-2 * h2[:rows].abs())).max()))
    dout = torch.randn(rows, inner, generator=g).to(ops.BF16).to(dev)
    d_il = ops.geglu_bwd_interleaved(pre[:rows], dout)
    d_std = ops.geglu_bwd(pre_std, dout)
    assert torch.equal(d_il, d_std), "geglu_bwd_interleaved differs from geglu_bwd on the same values"
    return excess


def case_pcm_fm_math(dev, g):
    """flow-matching PCM math + samplers of the SD3 variant: BIT-EXACT against the reference's own source
    (tests/golden/pcm_fm_golden.safetensors, generated by tests/golden/make_golden_sd3.py)."""
    from pcm_amd import fm
    sol = fm.EulerSolver(fm.flow_sigmas(1000, 3.0), 1000, 50, device=dev)
    for k in ("euler_timesteps", "euler_timesteps_prev", "sigmas", "sigmas_prev"):
        assert getattr(sol, k).dtype == g[k].dtype and torch.equal(getattr(sol, k).cpu(), g[k]), k
    idx = g["index"].to(dev)
    t, tp = sol.timesteps(idx)
    assert torch.equal(t.cpu(), g["timesteps"]) and torch.equal(tp.cpu(), g["timesteps_prev"])
    noisy = sol.add_noise(g["x"].to(dev), g["noise"].to(dev), idx)
    assert torch.equal(noisy.cpu(), g["noisy"]), "fm add_noise"
    pred = g["pred"].to(dev)
    for M in (1, 2, 4, 8):
        for tgt, name in ((False, "online"), (True, "target")):
            xp, end = sol.euler_style_multiphase_pred(noisy, pred, idx, M, tgt)
            assert xp.dtype == torch.float64 and torch.equal(xp.cpu(), g[f"{name}_{M}_x"]), (name, M)
            assert torch.equal(end.cpu(), g[f"{name}_{M}_end"]), (name, M)
    xp, xp32 = sol.euler_step(noisy, g["cond"].to(dev), idx, g["uncond"].to(dev), 3)
    assert torch.equal(xp.cpu(), g["euler_step"]) and torch.equal(xp32.cpu(), g["euler_step"].float()), "fm cfg + euler_step"
    xn, _ = sol.euler_step(noisy, g["teacher"].to(dev), idx, None)                       # --not_apply_cfg_solver: teacher used as is
    assert torch.equal(xn.cpu(), g["euler_step"])
    for M in (1, 4):
        ct, end = sol.euler_style_multiphase_pred(xp, pred, idx, M, True)                  # float64 sample
        assert torch.equal(ct.cpu(), g[f"chain_target_{M}_x"]) and torch.equal(end.cpu(), g[f"chain_target_{M}_end"]), M
    # adversarial trainers' re-noising of the phase-edge prediction
    fa64, fa32, ratio = sol.noise_travel(g["online_4_x"].to(dev), g["adv_noise"].to(dev), g["online_4_end"].to(dev), g["adv_index"].to(dev))
    assert torch.equal(fa64.cpu(), g["fake_adv"]) and torch.equal(fa32.cpu(), g["fake_adv"].float()), "fm noise travel"
    sp = g["sigmas_prev"]
    close(ratio, ((1 - sp[g["adv_index"]]) / (1 - sp[g["online_4_end"]])).float(), 1e-6, 0, "fm noise travel ratio")
    assert torch.equal((sol.sigmas_prev[g["adv_index"].to(dev)] * 1000).cpu(), g["timesteps_adv"])
    # huber loss through the step's loss kernel (same expression as the SD1.5 trainer): value parity
    B = idx.shape[0]
    loss = torch.zeros(1, dtype=torch.float64, device=dev)
    d = torch.empty_like(noisy)
    a, b = g["online_4_x"].float().to(dev).contiguous(), g["target_4_x"].float().to(dev).contiguous()
    ones = torch.ones(B, dtype=torch.float32, device=dev)
    capi.lib().call("pcm_consistency_loss", ops.ptr(a), ops.ptr(b), ops.ptr(ones), 1, 0.001, ops.ptr(loss), ops.ptr(d), 1.0, B, a.numel() // B, capi.Lib.stream())
    ref_d = (a - b).cpu() / torch.sqrt((a - b).cpu() ** 2 + 0.001 ** 2) / a.numel()
    close(d, ref_d, 1e-4, 1e-9, "fm huber grad")
    assert abs(float(loss) - float(g["huber_loss"])) <= 2e-6 * abs(float(g["huber_loss"])), (float(loss), float(g["huber_loss"]))
    # samplers
    for stochastic in (False, True):
        kind = "stochastic" if stochastic else "deterministic"
        for steps in (1, 2, 4, 8):
            sm = fm.PCMFMSampler(1000, 3.0, 100, stochastic=stochastic)
            sm.set_timesteps(steps, device=dev)
            assert torch.equal(sm.timesteps.cpu(), g[f"sampler_timesteps_{steps}"]) and torch.equal(sm.sigmas_, g[f"sampler_sigmas_{steps}"])
            lat = g[f"{kind}_{steps}_x0"].to(dev)
            for i, tt in enumerate(sm.timesteps):
                nz = g[f"{kind}_{steps}_noise{i}"].to(dev) if stochastic else None
                lat = sm.step(g[f"{kind}_{steps}_v{i}"].to(dev), tt, lat, noise=nz)
                assert torch.equal(lat.cpu(), g[f"{kind}_{steps}_x{i + 1}"]), (kind, steps, i)
    # fused guidance combine of the sampler step
    sm3 = fm.PCMFMSampler(1000, 3.0, 100)
    sm3.set_timesteps(2, device=dev)
    vc, vu, x3 = g["cond"][:2].to(dev), g["uncond"][:2].to(dev), g["x"][:2].to(dev)
    got = sm3.step(vc, sm3.timesteps[0], x3, model_output_uncond=vu, guidance_scale=1.7)
    sm4 = fm.PCMFMSampler(1000, 3.0, 100)
    sm4.set_timesteps(2, device=dev)
    vcomb = (vu.cpu() + 1.7 * (vc.cpu() - vu.cpu())).to(dev)
    close(got, sm4.step(vcomb, sm4.timesteps[0], x3).cpu(), 1e-6, 1e-6, "fm sampler cfg")
    with pytest.raises(ValueError):
        sm2 = fm.PCMFMSampler(1000, 3.0, 100)
        sm2.set_timesteps(2)
        sm2.step(g["pred"][:1].to(dev), 3, g["x"][:1].to(dev))


def case_mmdit_ops(dev):
    """element-wise pieces of the MMDiT blocks (SD3 variant) against plain torch fp32."""
    B, L, C = 3, 13, 192
    x = rnd(B * L, C, seed=1, dev=dev, shift=0.2)
    dy = rnd(B * L, C, seed=2, dev=dev)
    scale = rnd(B, C, seed=3, dev=dev, dtype=torch.float32, scale=0.3)
    shift = rnd(B, C, seed=4, dev=dev, dtype=torch.float32, scale=0.3)
    gamma = (1.0 + scale).contiguous()
    y, mean, rstd = ops.layernorm_mod_fwd(x, gamma, shift, L, eps=1e-6)
    xr = x.float().cpu().view(B, L, C).requires_grad_(True)
    ref = F.layer_norm(xr, (C,), eps=1e-6) * (1 + scale.cpu()[:, None, :]) + shift.cpu()[:, None, :]
    close(y.view(B, L, C), ref.detach(), 1e-2, 1e-2, "adaLN fwd")
    ref.backward(dy.float().cpu().view(B, L, C))
    dres = rnd(B * L, C, seed=5, dev=dev)
    dx = ops.layernorm_mod_bwd(x, dy, gamma, mean, rstd, L, dres=dres)
    close(dx.view(B, L, C), xr.grad + dres.float().cpu().view(B, L, C), 1e-2, 2e-2, "adaLN bwd")
    gate = rnd(B, C, seed=6, dev=dev, dtype=torch.float32)
    res = rnd(B * L, C, seed=7, dev=dev)
    out = ops.rowgate_fma(dy, gate, L, res=res)
    close(out.view(B, L, C), res.float().cpu().view(B, L, C) + gate.cpu()[:, None, :] * dy.float().cpu().view(B, L, C), 1e-2, 1e-2, "gate fma")
    out = ops.rowgate_fma(dy, gate, L)
    close(out.view(B, L, C), gate.cpu()[:, None, :] * dy.float().cpu().view(B, L, C), 1e-2, 1e-2, "gate mul")
    h = rnd(257, 136, seed=8, dev=dev, scale=1.5)
    hr = h.float().cpu().requires_grad_(True)
    refg = F.gelu(hr, approximate="tanh")
    close(ops.gelu_tanh_fwd(h), refg.detach(), 1e-2, 1e-2, "gelu tanh")
    dh = rnd(257, 136, seed=9, dev=dev)
    refg.backward(dh.float().cpu())
    close(ops.gelu_tanh_bwd(h, dh), hr.grad, 1e-2, 2e-2, "gelu tanh bwd")
    # patchify: order 0 == Conv2d(k=2, s=2) im2col, order 1 == inverse of the unpatchify einsum
    Bi, Ci, H, W = 2, 16, 6, 10
    img = rnd(Bi, Ci, H, W, seed=10, dev=dev, dtype=torch.float32)
    tok0 = ops.patchify2x2(img, 0)
    refu = F.unfold(img.cpu().to(ops.BF16).float(), kernel_size=2, stride=2).transpose(1, 2).reshape(Bi * (H // 2) * (W // 2), 4 * Ci)
    assert torch.equal(tok0.float().cpu(), refu), "patchify (c,p,q)"
    tok1 = ops.patchify2x2(img, 1)
    back = ops.unpatchify2x2(tok1.float(), Bi, Ci, H, W)
    assert torch.equal(back.cpu(), img.cpu().to(ops.BF16).float()), "patchify (p,q,c) / unpatchify roundtrip"
    tk = rnd(Bi * (H // 2) * (W // 2), 4 * Ci, seed=11, dev=dev, dtype=torch.float32)
    hs = tk.cpu().reshape(Bi, H // 2, W // 2, 2, 2, Ci)
    refimg = torch.einsum("nhwpqc->nchpwq", hs).reshape(Bi, Ci, H, W)                     # discriminator_sd3.py:112-131
    assert torch.equal(ops.unpatchify2x2(tk, Bi, Ci, H, W).cpu(), refimg), "unpatchify einsum"
    tk0 = rnd(Bi * (H // 2) * (W // 2), 4 * Ci, seed=12, dev=dev, dtype=torch.float32)
    ref0 = F.fold(tk0.cpu().reshape(Bi, (H // 2) * (W // 2), 4 * Ci).transpose(1, 2), (H, W), kernel_size=2, stride=2)
    assert torch.equal(ops.unpatchify2x2(tk0, Bi, Ci, H, W, order=0).cpu(), ref0), "unpatchify (c,p,q)"
    # modulation-vector gradients
    ga, gb = ops.mod_grad(x, dy, B, mean, rstd)
    xh = ((x.float().cpu().view(B, L, C) - mean.cpu().view(B, L, 1)) * rstd.cpu().view(B, L, 1))
    dyf = dy.float().cpu().view(B, L, C)
    close(ga, (dyf * xh).sum(1), 1e-3, 1e-3, "mod grad scale")
    close(gb, dyf.sum(1), 1e-3, 1e-3, "mod grad shift")
    gg, none = ops.mod_grad(x, dy, B, want_b=False)
    assert none is None
    close(gg, (dyf * x.float().cpu().view(B, L, C)).sum(1), 1e-3, 1e-3, "mod grad gate")
    t = torch.tensor([999.97, 500.25, 3.0, 57.7], device=dev)
    emb = ops.timestep_embedding_f32(t, 256)
    half = 128
    fr = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    arg = t.cpu()[:, None] * fr[None, :]
    close(emb, torch.cat([torch.cos(arg), torch.sin(arg)], -1), 1e-2, 1e-2, "timestep f32")


def case_pcm_math_random_shapes(dev):
    """reference-owned math on shapes / parameters the golden fixture does not hold: ragged per-sample sizes (not multiples of the
    256-thread block), batch 1, every multiphase from 1 to the number of solver steps, other solver step counts and schedule shifts.
    The flow-matching part (no transcendental / sqrt) is BIT-EXACT against the oracle restatement (pinned bit-exactly to the reference's
    source); the DDPM part to 1 ulp."""
    import numpy as np
    from oracle import pcm_fm_math as FM
    from oracle import pcm_math as PM
    from pcm_amd import fm
    g = torch.Generator().manual_seed(2024)
    # ---- flow matching (SD3 variant)
    for B, shape, E, shift in ((1, (16, 5, 7), 20, 3.0), (3, (16, 2, 2), 50, 1.0), (5, (4, 9, 3), 100, 3.0), (2, (16, 16, 16), 50, 2.0)):
        osol = FM.EulerSolver(FM.flow_sigmas(1000, shift), 1000, E)
        sol = fm.EulerSolver(fm.flow_sigmas(1000, shift), 1000, E, device=dev)
        assert torch.equal(sol.sigmas.cpu(), osol.sigmas) and torch.equal(sol.sigmas_prev.cpu(), osol.sigmas_prev)
        x, nz, pr, c, u = (torch.randn(B, *shape, generator=g) for _ in range(5))
        idx = torch.randint(0, E, (B,), generator=g)
        idx[0] = E - 1
        noisy = FM.fm_add_noise(osol, x, nz, idx)
        got = sol.add_noise(x.to(dev), nz.to(dev), idx.to(dev))
        assert torch.equal(got.cpu(), noisy)
        xp = osol.euler_step(noisy, FM.fm_cfg(c, u, 3), idx)
        gp, gp32 = sol.euler_step(got, c.to(dev), idx.to(dev), u.to(dev), 3)
        assert torch.equal(gp.cpu(), xp) and torch.equal(gp32.cpu(), xp.float())
        for M in sorted({1, 2, 3, 7, E}):
            for tgt, smp, gsmp in ((False, noisy, got), (True, xp, gp)):
                a, ea = osol.euler_style_multiphase_pred(smp, pr, idx, M, tgt)
                b, eb = sol.euler_style_multiphase_pred(gsmp, pr.to(dev), idx.to(dev), M, tgt)
                assert torch.equal(b.cpu(), a) and torch.equal(eb.cpu(), ea), (B, shape, E, M, tgt)
        end = ea
        adv = torch.minimum(end + torch.randint(0, max(1, E // 7), (B,), generator=g), torch.full((B,), E - 1))
        n64 = torch.randn(B, *shape, generator=g, dtype=torch.float64)
        ra = FM.fm_noise_travel(osol, a, n64, end, adv)
        rb, rb32, _ = sol.noise_travel(b, n64.to(dev), end.to(dev), adv.to(dev))
        assert torch.equal(rb.cpu(), ra) and torch.equal(rb32.cpu(), ra.float())
    # ---- DDPM side (SD1.5 / SDXL): add_noise and noise_travel on ragged shapes
    acp = PM.sd15_alphas_cumprod()
    for B, shape in ((1, (4, 5, 7)), (3, (4, 3, 3)), (2, (4, 17, 9))):
        x, nz = torch.randn(B, *shape, generator=g), torch.randn(B, *shape, generator=g)
        t = torch.randint(0, 1000, (B,), generator=g)
        t2 = torch.minimum(t + torch.randint(0, 250, (B,), generator=g), torch.full((B,), 999))
        # (these two take square roots: the kernels use the correctly rounded sqrt, torch's CPU sqrt is 1 ulp off on some hosts -- the
        # reason the bit-exact anchor for them is the committed fixture, DESIGN.md section 5 -- so a LIVE oracle is compared to 1 ulp)
        ref = PM.add_noise(acp, x, nz, t)
        assert torch.allclose(ops.add_noise(x.to(dev), nz.to(dev), acp.to(dev), t.to(dev)).cpu(), ref, rtol=3e-7, atol=3e-7)
        got, _ = ops.noise_travel(x.to(dev), nz.to(dev), acp.to(dev), t.to(dev), t2.to(dev))
        assert torch.allclose(got.cpu(), PM.noise_travel(acp, x, nz, t, t2), rtol=3e-7, atol=3e-7)
