"""MMDiT / SD3-step parity cases shared by the host-emulation (CPU) and the GPU test files."""
import torch


def run_case(dev, adv_targets=False):
    """``adv_targets``: the adversarial trainers' 22-entry LoRA list (context stream, adaLN projections, embedders, patch conv)."""
    from oracle import mmdit_sd3 as O
    from pcm_amd.mmdit import MMDiT, MMDiTWeights, sd3_lora_state
    from pcm_amd.mmdit_spec import LORA_TARGETS_SD3_ADV, MMDiTConfig
    kw = dict(sample_size=16, num_layers=3, attention_head_dim=64, num_attention_heads=2, joint_attention_dim=96, caption_projection_dim=128,
              pooled_projection_dim=64, pos_embed_max_size=12)
    oc, pc = O.MMDiTConfig(**kw), MMDiTConfig(**kw)
    sd = O.init_state_dict(oc, 0)
    B, H, Wd, Lc = 2, 8, 12, 7
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, 16, H, Wd, generator=g)
    t = torch.tensor([901.25, 57.5])
    ctx = torch.randn(B, Lc, 96, generator=g)
    pooled = torch.randn(B, 64, generator=g)
    d_out = torch.randn(B, 16, H, Wd, generator=g)
    W = MMDiTWeights(pc, {k: v.to(dev) for k, v in sd.items()}, dev)
    if adv_targets:
        lora = sd3_lora_state(pc, 32, 8.0, dev, seed=1, b_std=0.05, targets=LORA_TARGETS_SD3_ADV, init="kaiming")
        expect = {p for p, _ in O.lora_target_modules(oc, O.LORA_SUFFIXES_ADV)}
        assert "pos_embed.proj" in expect and "transformer_blocks.0.norm1.linear" in expect and not any("add_q_proj" in p for p in expect)
    else:
        lora = sd3_lora_state(pc, 32, 8.0, dev, seed=1, b_std=0.1)
        expect = {p for p, _ in O.lora_target_modules(oc)}
    assert lora.real_rank == 32 and lora.rank == 64 and abs(lora.scaling - 0.25) < 1e-12
    olora = {p: (m.A[:32].detach().cpu().clone().requires_grad_(True), m.B[:, :32].detach().cpu().clone().requires_grad_(True)) for p, m in lora.modules.items()}
    assert set(olora) == expect
    assert all(float(m.A[32:].abs().max()) == 0.0 and float(m.B[:, 32:].abs().max()) == 0.0 for m in lora.modules.values())
    ref_t = O.mmdit_forward(oc, sd, x, t, ctx, pooled)
    ref_s = O.mmdit_forward(oc, sd, x, t, ctx, pooled, olora, 8.0)
    xd, td, cd, pd = x.to(dev), t.to(dev), ctx.to(dev), pooled.to(dev)
    out_t = MMDiT(W, None).forward(xd, td, cd, pd)
    student = MMDiT(W, lora)
    out_s, tape = student.forward(xd, td, cd, pd, save=True)
    from pcm_amd import mmdit as _mm
    assert all(r["fuse"] == _mm.FUSE_QKV for r in tape[:-1])        # the fused q/k/v schedule is the one that ran (unless switched off)
    scale = ref_t.abs().max().item()
    err_t = (out_t.cpu() - ref_t).abs().max().item()
    err_s = (out_s.cpu() - ref_s.detach()).abs().max().item()
    effect = (ref_s - ref_t).abs().max().item()
    print("fwd err teacher %.3e student %.3e (scale %.3e), lora effect %.3e" % (err_t, err_s, scale, effect))
    assert err_t < 0.03 * scale and err_s < 0.03 * scale
    assert effect > 5 * err_s, "LoRA branch not exercised"
    (ref_s * d_out).sum().backward()
    lora.zero_grad()
    student.backward(d_out.to(dev), tape)
    num = den = 0.0
    worst = 0.0
    for p, m in lora.modules.items():
        assert float(m.gA[32:].abs().max()) == 0.0 and float(m.gB[:, 32:].abs().max()) == 0.0, "padded LoRA ranks must get exactly zero gradient"
        for got, ref in ((m.gA[:32].cpu(), olora[p][0].grad), (m.gB[:, :32].cpu(), olora[p][1].grad)):
            ref = ref.view_as(got)
            num += float(((got - ref) ** 2).sum())
            den += float((ref ** 2).sum())
            rel = float((got - ref).norm() / (ref.norm() + 1e-12))
            worst = max(worst, rel)
            assert rel < (0.3 if adv_targets else 0.15), (p, rel)
    print("grad rel err: global %.3e worst module %.3e" % ((num / den) ** 0.5, worst))
    assert (num / den) ** 0.5 < 0.05


def run_step_case(dev, not_apply_cfg_solver=False):
    """one SD3 PCM distillation step (train_pcm_lora_sd3.py:1270-1390) through SD3Distiller vs the oracle step."""
    from oracle import mmdit_sd3 as O
    from oracle import pcm_step_sd3 as OS
    from pcm_amd.mmdit import MMDiTWeights, sd3_lora_state
    from pcm_amd.mmdit_spec import MMDiTConfig
    from pcm_amd.trainer_sd3 import SD3Distiller, SD3StepConfig
    kw = dict(sample_size=16, num_layers=2, attention_head_dim=64, num_attention_heads=2, joint_attention_dim=96, caption_projection_dim=128,
              pooled_projection_dim=64, pos_embed_max_size=12)
    oc, pc = O.MMDiTConfig(**kw), MMDiTConfig(**kw)
    sd = O.init_state_dict(oc, 0)
    W = MMDiTWeights(pc, {k: v.to(dev) for k, v in sd.items()}, dev)
    lora = sd3_lora_state(pc, 32, 8.0, dev, seed=1, b_std=0.1)
    olora = {p: (m.A[:32].detach().cpu().clone().requires_grad_(True), m.B[:, :32].detach().cpu().clone().requires_grad_(True)) for p, m in lora.modules.items()}
    B, H, Wd, Lc = 4, 8, 8, 5
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(B, 16, H, Wd, generator=g)
    noise = torch.randn(B, 16, H, Wd, generator=g)
    pe, upe = torch.randn(B, Lc, 96, generator=g), torch.randn(B, Lc, 96, generator=g)
    pp, upp = torch.randn(B, 64, generator=g), torch.randn(B, 64, generator=g)
    index = torch.tensor([0, 49, 13, 30])
    ref = OS.distill_step_sd3(oc, sd, olora, x0, pe, pp, upe, upp, noise, index, multiphase=4, not_apply_cfg_solver=not_apply_cfg_solver)
    ref["loss"].backward()
    cfg = SD3StepConfig(multiphase=4, not_apply_cfg_solver=not_apply_cfg_solver)
    D = SD3Distiller(W, lora, cfg)
    gsc = float(D.loss_scale_dev.item()) if D.loss_scale_dev is not None else 1.0      # half build: the gradient buffers hold S * grad
    p0 = lora.params.clone()
    out = D.step(*(t.to(dev) for t in (x0, pe, pp, upe, upp, noise, index)))
    assert torch.equal(out["end_index"].cpu(), ref["end_index"])
    assert torch.equal(out["noisy_model_input"].cpu(), ref["noisy_model_input"])                     # reference-owned math: bit-exact
    for k in ("model_output", "cond_teacher_output", "x_prev", "target", "model_pred"):
        a, b = out[k].double().cpu(), ref[k].detach().double()
        err = (a - b).abs().max().item()
        assert err < 0.04 * b.abs().max().item(), (k, err, b.abs().max().item())
    rl = float(ref["loss"].detach())
    assert abs(float(out["loss"]) - rl) < 5e-2 * abs(rl), (float(out["loss"]), rl)
    num = den = 0.0
    for p, m in lora.modules.items():
        for got, r in ((m.gA[:32].cpu() / gsc, olora[p][0].grad), (m.gB[:, :32].cpu() / gsc, olora[p][1].grad)):
            num += float(((got - r) ** 2).sum())
            den += float((r ** 2).sum())
    print("loss %.5f (oracle %.5f), LoRA grad rel err %.3e" % (float(out["loss"]), rl, (num / den) ** 0.5))
    # bf16 compute + fp32 atomics vs the fp32 oracle on a 128-wide model: ~5 % here (the same measure is ~1 % for a plain forward/backward
    # with a smooth cotangent; the huber gradient's sign pattern makes the sums cancel more); the bound leaves room for atomic ordering
    assert (num / den) ** 0.5 < 0.12
    assert not torch.equal(lora.params, p0) and D.step_count == 1                                      # AdamW applied
    assert float(lora.params.view(-1)[:0].numel()) == 0
    for m in lora.modules.values():                                                                    # padded ranks stay exactly zero
        assert float(m.A[32:].abs().max()) == 0.0 and float(m.B[:, 32:].abs().max()) == 0.0


def run_sampler_case(dev):
    """few-step latent sampling with the SD3 student (deterministic + guidance, stochastic) vs the oracle loop."""
    from oracle import mmdit_sd3 as O
    from oracle import pcm_fm_math as FM
    from pcm_amd.mmdit import MMDiT, MMDiTWeights, sd3_lora_state
    from pcm_amd.mmdit_spec import MMDiTConfig
    from pcm_amd.sampler_sd3 import PCMFMLatentSampler
    kw = dict(sample_size=16, num_layers=2, attention_head_dim=64, num_attention_heads=2, joint_attention_dim=96, caption_projection_dim=128,
              pooled_projection_dim=64, pos_embed_max_size=12)
    oc, pc = O.MMDiTConfig(**kw), MMDiTConfig(**kw)
    sd = O.init_state_dict(oc, 0)
    W = MMDiTWeights(pc, {k: v.to(dev) for k, v in sd.items()}, dev, need_bwd=False)
    lora = sd3_lora_state(pc, 32, 8.0, dev, seed=1, b_std=0.1)
    olora = {p: (m.A[:32].detach().cpu().clone(), m.B[:, :32].detach().cpu().clone()) for p, m in lora.modules.items()}
    B, H, Wd, Lc = 2, 8, 8, 5
    g = torch.Generator().manual_seed(21)
    lat = torch.randn(B, 16, H, Wd, generator=g)
    pe, un = torch.randn(B, Lc, 96, generator=g), torch.randn(B, Lc, 96, generator=g)
    pp, unp = torch.randn(B, 64, generator=g), torch.randn(B, 64, generator=g)

    def model_fn(x, t, c, p):
        with torch.no_grad():
            return O.mmdit_forward(oc, sd, x, t, c, p, olora, 8.0)
    m = MMDiT(W, lora)
    to = lambda *ts: (t.to(dev) for t in ts)   # noqa: E731
    for steps, gs in ((1, 1.0), (4, 1.0), (2, 1.5)):
        ref = FM.fm_sample(model_fn, pe, pp, un, unp, lat, steps, gs)
        a, b, c, d, e = to(pe, pp, un, unp, lat)
        got = PCMFMLatentSampler(m).sample(a, b, c, d, steps, gs, latents=e)
        err = (got.cpu() - ref).abs().max().item()
        assert err < 0.05 * ref.abs().max().item(), (steps, gs, err, ref.abs().max().item())
    noises = [torch.randn(B, 16, H, Wd, generator=g) for _ in range(3)]
    ref = FM.fm_sample(model_fn, pe, pp, None, None, lat, 3, 1.0, stochastic=True, noises=noises)
    # same noise through the product sampler: drive the scheduler by hand
    from pcm_amd.fm import PCMFMSampler
    sch = PCMFMSampler(1000, 3.0, 100, stochastic=True)
    sch.set_timesteps(3, device=dev)
    a, b, x = (*to(pe, pp), lat.to(dev))
    for i, t in enumerate(sch.timesteps):
        x = sch.step(m.forward(x, t.expand(B).contiguous(), a, b), t, x, noise=noises[i].to(dev))
    err = (x.cpu() - ref).abs().max().item()
    assert err < 0.05 * ref.abs().max().item(), ("stochastic", err)


def run_adv_case(dev, global_step):
    """one adversarial SD3 step (train_pcm_lora_sd3_adv.py:1330-1520) with the trainers' 22-entry LoRA list: even = discriminator
    update (head gradients vs oracle autograd), odd = generator update (loss_cm + adv_weight * g_loss -> LoRA gradients)."""
    from oracle import mmdit_sd3 as O
    from oracle import pcm_step_sd3 as OS
    from pcm_amd.discriminator import Discriminator
    from pcm_amd.mmdit import MMDiTWeights, sd3_lora_state
    from pcm_amd.mmdit_spec import LORA_TARGETS_SD3_ADV, MMDiTConfig
    from pcm_amd.trainer_sd3 import SD3AdvDistiller, SD3StepConfig
    kw = dict(sample_size=16, num_layers=2, attention_head_dim=64, num_attention_heads=2, joint_attention_dim=96, caption_projection_dim=128,
              pooled_projection_dim=64, pos_embed_max_size=12)
    oc, pc = O.MMDiTConfig(**kw), MMDiTConfig(**kw)
    sd = O.init_state_dict(oc, 0)
    W = MMDiTWeights(pc, {k: v.to(dev) for k, v in sd.items()}, dev)
    lora = sd3_lora_state(pc, 32, 8.0, dev, seed=1, b_std=0.05, targets=LORA_TARGETS_SD3_ADV, init="kaiming")
    olora = {p: (m.A[:32].detach().cpu().clone().requires_grad_(True), m.B[:, :32].detach().cpu().clone().requires_grad_(True)) for p, m in lora.modules.items()}
    disc = Discriminator([128] * 2, num_h_per_head=1, device=dev, seed=4, ksize=1)
    dsd = {k: v.detach().cpu().clone() for k, v in disc.state_dict().items()}
    B, H, Wd, Lc = 4, 8, 8, 5
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(B, 16, H, Wd, generator=g)
    noise = torch.randn(B, 16, H, Wd, generator=g)
    pe, upe = torch.randn(B, Lc, 96, generator=g), torch.randn(B, Lc, 96, generator=g)
    pp, upp = torch.randn(B, 64, generator=g), torch.randn(B, 64, generator=g)
    nf = torch.randn(B, 16, H, Wd, generator=g, dtype=torch.float64)
    nr = torch.randn(B, 16, H, Wd, generator=g, dtype=torch.float64)
    index = torch.tensor([0, 49, 13, 30])
    adv_u = torch.tensor([0.0, 0.99, 0.5, 0.26])
    span = 50 // 4
    off = torch.clamp((adv_u * span).long(), max=span - 1)
    ref = OS.distill_step_sd3_adv(oc, sd, olora, dsd, x0, pe, pp, upe, upp, noise, index, off, nf, nr, global_step, multiphase=4, adv_weight=0.1)
    D = SD3AdvDistiller(W, lora, SD3StepConfig(multiphase=4), disc, adv_weight=0.1, adv_lr=1e-5)
    gsc = float(D.loss_scale_dev.item()) if D.loss_scale_dev is not None else 1.0      # half build: the gradient buffers hold S * grad
    p0, d0 = lora.params.clone(), disc.params.clone()
    out = D.step_adv(global_step, *(t.to(dev) for t in (x0, pe, pp, upe, upp, noise, index, nf, nr, adv_u)))
    assert torch.equal(out["adv_index"].cpu(), ref["adv_index"])
    fa = ref["fake_adv"].detach().float()
    assert (out["fake_adv"].cpu() - fa).abs().max().item() < 0.04 * fa.abs().max().item()
    if global_step % 2 == 0:
        rl = float(ref["d_loss"])
        assert abs(float(out["d_loss"]) - rl) < 3e-2 * abs(rl), (float(out["d_loss"]), rl)
        got = {}
        cnt = {}
        for k, hd in disc.heads:
            h = cnt.get(k, 0)
            cnt[k] = h + 1
            for n, t in hd.g.items():
                v = t.detach().cpu().clone() / gsc
                got[f"heads.{k}.{h}.{n}"] = v
        num = den = 0.0
        for n, gref in ref["head_grads"].items():
            a, b = got[n].reshape(-1), gref.reshape(-1)
            num += float(((a - b) ** 2).sum())
            den += float((b ** 2).sum())
        mine_all = torch.cat([got[n].reshape(-1) for n in ref["head_grads"]])
        ref_all = torch.cat([gr.reshape(-1) for gr in ref["head_grads"].values()])
        cos = float((mine_all * ref_all).sum() / (mine_all.norm() * ref_all.norm()))
        print("d_loss %.5f (oracle %.5f), head grad rel err %.3e cos %.4f" % (float(out["d_loss"]), rl, (num / den) ** 0.5, cos))
        # the hinge cotangent is +-1/n per logit: every head-parameter gradient is a heavily cancelling sum, so bf16 rounding of the
        # summands shows as ~10 % relative error on this tiny config (same criterion as tests/test_emu_adv.py): direction + magnitude
        assert cos > 0.97 and (num / den) ** 0.5 < 0.25, (cos, (num / den) ** 0.5)
        assert torch.equal(lora.params, p0) and not torch.equal(disc.params, d0)          # only the heads moved
        return
    ref["loss"].backward()
    rl, rg = float(ref["loss_cm"].detach()), float(ref["g_loss"].detach()) / 0.1      # out["g_loss"] is the unweighted hinge term
    assert abs(float(out["loss_cm"]) - rl) < 5e-2 * abs(rl) and abs(float(out["g_loss"]) - rg) < 5e-2 * abs(rg), (float(out["loss_cm"]), rl, float(out["g_loss"]), rg)
    num = den = 0.0
    for p, m in lora.modules.items():
        for got, r in ((m.gA[:32].cpu() / gsc, olora[p][0].grad), (m.gB[:, :32].cpu() / gsc, olora[p][1].grad)):
            r = r.view_as(got)
            num += float(((got - r) ** 2).sum())
            den += float((r ** 2).sum())
    print("loss_cm %.5f (%.5f) g_loss %.5f (%.5f), LoRA grad rel err %.3e" % (float(out["loss_cm"]), rl, float(out["g_loss"]), rg, (num / den) ** 0.5))
    assert (num / den) ** 0.5 < 0.15
    assert not torch.equal(lora.params, p0) and torch.equal(disc.params, d0)              # only the student moved


def run_accumulation_case(dev):
    """--gradient_accumulation_steps 2: two micro-batches of 2 (gradient scaled by 1/2 each, one exchange + AdamW at the end) land on the
    same LoRA parameters as one step on the 4 samples."""
    from oracle import mmdit_sd3 as O
    from pcm_amd.mmdit import MMDiTWeights, sd3_lora_state
    from pcm_amd.mmdit_spec import MMDiTConfig
    from pcm_amd.trainer_sd3 import SD3Distiller, SD3StepConfig
    kw = dict(sample_size=16, num_layers=2, attention_head_dim=64, num_attention_heads=2, joint_attention_dim=96, caption_projection_dim=128,
              pooled_projection_dim=64, pos_embed_max_size=12)
    sd = O.init_state_dict(O.MMDiTConfig(**kw), 0)
    pc = MMDiTConfig(**kw)
    W = MMDiTWeights(pc, {k: v.to(dev) for k, v in sd.items()}, dev)
    g = torch.Generator().manual_seed(9)
    G, H, Lc = 4, 8, 5
    ts = [torch.randn(G, 16, H, H, generator=g), torch.randn(G, Lc, 96, generator=g), torch.randn(G, 64, generator=g),
          torch.randn(G, Lc, 96, generator=g), torch.randn(G, 64, generator=g), torch.randn(G, 16, H, H, generator=g), torch.tensor([3, 41, 17, 28])]
    res = []
    for mode in ("single", "accum"):
        lora = sd3_lora_state(pc, 32, 8.0, dev, seed=5, b_std=0.05)
        init = lora.params.clone()
        D = SD3Distiller(W, lora, SD3StepConfig(multiphase=2, learning_rate=1e-3))
        if mode == "single":
            D.step(*(t.to(dev) for t in ts))
        else:
            for i in range(2):
                out = D.step(*(t[2 * i:2 * i + 2].contiguous().to(dev) for t in ts), accum=(i, 2))
                assert (D.step_count == 1) == (i == 1)                      # the optimizer runs with the last micro-batch only
        res.append((lora.params - init).double().cpu())
    u1, u2 = res
    cos = float((u1 * u2).sum() / (u1.norm() * u2.norm()))
    assert cos > 0.98 and 0.9 < float(u2.norm() / u1.norm()) < 1.1, (cos, float(u2.norm() / u1.norm()))


def run_prefetch_case(dev, graphs=False):
    """SD3Distiller.step(..., prefetch=next batch): the frozen teacher's pass of the next batch issued ahead (on a side stream on the GPU) changes
    nothing -- three steps give bitwise the same losses and LoRA parameters as three plain steps.  ``graphs`` (GPU): a third run through
    capture(pipeline=True) / step_graphed(prefetch=...) joins the comparison, reductions in reproducible mode."""
    from oracle import mmdit_sd3 as O
    from pcm_amd import ops
    from pcm_amd.mmdit import MMDiTWeights, sd3_lora_state
    from pcm_amd.mmdit_spec import MMDiTConfig
    from pcm_amd.trainer_sd3 import SD3Distiller, SD3StepConfig
    kw = dict(sample_size=16, num_layers=2, attention_head_dim=64, num_attention_heads=2, joint_attention_dim=96, caption_projection_dim=128,
              pooled_projection_dim=64, pos_embed_max_size=12)
    sd = O.init_state_dict(O.MMDiTConfig(**kw), 0)
    pc = MMDiTConfig(**kw)
    W = MMDiTWeights(pc, {k: v.to(dev) for k, v in sd.items()}, dev)
    g = torch.Generator().manual_seed(9)
    B, H, Lc, n = 2, 8, 5, 3
    batches = [tuple(t.to(dev) for t in (torch.randn(B, 16, H, H, generator=g), torch.randn(B, Lc, 96, generator=g), torch.randn(B, 64, generator=g),
                                         torch.randn(B, Lc, 96, generator=g), torch.randn(B, 64, generator=g), torch.randn(B, 16, H, H, generator=g),
                                         torch.randint(0, 50, (B,), generator=g))) for _ in range(n + 1)]

    def run(mode):
        import os
        return run_(mode)

    def run_(mode):
        lora = sd3_lora_state(pc, 32, 8.0, dev, seed=5, b_std=0.05)
        D = SD3Distiller(W, lora, SD3StepConfig(multiphase=2, learning_rate=1e-3))
        if mode == "graph":
            D.capture(B, H=H, W=H, ctx_len=Lc, pipeline=True)
        losses = []
        for i in range(n):
            nxt = batches[i + 1] if mode != "plain" else None
            if mode != "plain" and i == 1:
                nxt = None                                   # a step that announces nothing: the next one computes its own targets
            out = (D.step_graphed if mode == "graph" else D.step)(*batches[i], prefetch=nxt)
            losses.append(float(out["loss"]))
        return losses, lora.params.clone()

    if graphs:
        ops.set_deterministic(True)
    try:
        runs = [run(m) for m in (("plain", "prefetch", "graph") if graphs else ("plain", "prefetch"))]
    finally:
        if graphs:
            ops.set_deterministic(False)
    for l, p_ in runs[1:]:
        assert l == runs[0][0] and torch.equal(p_, runs[0][1]), (runs[0][0], l)
    assert len(set(runs[0][0])) == n


def run_online_target_modes_case(dev):
    """SD3Distiller.online_target_mode: the online and the target forward as two passes ("serial" / "side": the same launches on one or two streams ->
    bitwise the same step) or as ONE 2B-sample pass whose tape is halved for the backward ("fused": the same arithmetic per sample, contraction plans
    may differ with the row count -> equal to summation rounding)."""
    from oracle import mmdit_sd3 as O
    from pcm_amd.mmdit import MMDiTWeights, sd3_lora_state
    from pcm_amd.mmdit_spec import MMDiTConfig
    from pcm_amd.trainer_sd3 import SD3Distiller, SD3StepConfig
    kw = dict(sample_size=16, num_layers=2, attention_head_dim=64, num_attention_heads=2, joint_attention_dim=96, caption_projection_dim=128,
              pooled_projection_dim=64, pos_embed_max_size=12)
    sd = O.init_state_dict(O.MMDiTConfig(**kw), 0)
    pc = MMDiTConfig(**kw)
    W = MMDiTWeights(pc, {k: v.to(dev) for k, v in sd.items()}, dev)
    g = torch.Generator().manual_seed(11)
    B, H, Lc = 2, 8, 5
    batch = tuple(t.to(dev) for t in (torch.randn(B, 16, H, H, generator=g), torch.randn(B, Lc, 96, generator=g), torch.randn(B, 64, generator=g),
                                      torch.randn(B, Lc, 96, generator=g), torch.randn(B, 64, generator=g), torch.randn(B, 16, H, H, generator=g),
                                      torch.tensor([7, 33])))
    res = {}
    for mode in ("serial", "side", "fused"):
        lora = sd3_lora_state(pc, 32, 8.0, dev, seed=5, b_std=0.05)
        D = SD3Distiller(W, lora, SD3StepConfig(multiphase=2, learning_rate=1e-3))
        D._online_target_mode = mode
        out = D.step(*batch)
        res[mode] = (float(out["loss"]), out["target"].double().cpu().clone(), lora.grads.double().cpu().clone())
    assert res["serial"][0] == res["side"][0] and torch.equal(res["serial"][2], res["side"][2])
    l0, t0, g0 = res["serial"]
    l1, t1, g1 = res["fused"]
    assert abs(l1 - l0) <= 2e-3 * abs(l0), (l0, l1)
    assert float((t1 - t0).norm() / t0.norm()) < 2e-3
    assert float((g1 * g0).sum() / (g1.norm() * g0.norm())) > 0.999 and float(g0.norm()) > 0


def run_wgrad_defer_case(dev):
    """model.WGRAD_DEFER: the LoRA weight-gradient jobs of a backward go out per module (0) or collected across modules into few multi-job
    launches (32, the default; 3: a flush every third module, so that several flushes and the final one all happen in this narrow model).
    Under set_deterministic every job has its ordered finalize, so one training step gives BITWISE the same gradients and parameters."""
    from oracle import mmdit_sd3 as O
    from pcm_amd import model, ops
    from pcm_amd.mmdit import MMDiTWeights, sd3_lora_state
    from pcm_amd.mmdit_spec import MMDiTConfig
    from pcm_amd.trainer_sd3 import SD3Distiller, SD3StepConfig
    kw = dict(sample_size=16, num_layers=2, attention_head_dim=64, num_attention_heads=2, joint_attention_dim=96, caption_projection_dim=128,
              pooled_projection_dim=64, pos_embed_max_size=12)
    sd = O.init_state_dict(O.MMDiTConfig(**kw), 0)
    pc = MMDiTConfig(**kw)
    W = MMDiTWeights(pc, {k: v.to(dev) for k, v in sd.items()}, dev)
    g = torch.Generator().manual_seed(13)
    B, H, Lc = 2, 8, 5
    batch = tuple(t.to(dev) for t in (torch.randn(B, 16, H, H, generator=g), torch.randn(B, Lc, 96, generator=g), torch.randn(B, 64, generator=g),
                                      torch.randn(B, Lc, 96, generator=g), torch.randn(B, 64, generator=g), torch.randn(B, 16, H, H, generator=g),
                                      torch.tensor([5, 40])))
    res, keep = [], model.WGRAD_DEFER
    ops.set_deterministic(True)
    try:
        for n in (0, 3, 32):
            model.WGRAD_DEFER = n
            lora = sd3_lora_state(pc, 32, 8.0, dev, seed=5, b_std=0.05)
            D = SD3Distiller(W, lora, SD3StepConfig(multiphase=2, learning_rate=1e-3))
            out = D.step(*batch)
            res.append((float(out["loss"]), lora.grads.clone(), lora.params.clone()))
    finally:
        model.WGRAD_DEFER = keep
        ops.set_deterministic(False)
    assert float(res[0][1].abs().max()) > 0
    for l, g_, p_ in res[1:]:
        assert l == res[0][0] and torch.equal(g_, res[0][1]) and torch.equal(p_, res[0][2])


def run_property_case(dev, cfg, W, lora, hw, Lc):
    """size-independent properties of the MMDiT path (used at SD3-medium's real size on the GPU, where no fp32 oracle is affordable, and on
    a narrow config on the emulator): finite output, batch independence, B = 0 LoRA == teacher, one distillation step."""
    from pcm_amd.mmdit import MMDiT
    from pcm_amd.trainer_sd3 import SD3Distiller, SD3StepConfig
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)   # noqa: E731
    x, t, c, p = r(2, cfg.in_channels, hw, hw), torch.tensor([900.5, 120.25], device=dev), r(2, Lc, cfg.joint_attention_dim), r(2, cfg.pooled_projection_dim)
    teacher = MMDiT(W, None)
    out = teacher.forward(x, t, c, p)
    assert out.shape == (2, cfg.out_channels, hw, hw) and bool(torch.isfinite(out).all()) and float(out.abs().max()) > 0

    def rel(a, b):
        return float((a - b).norm() / b.norm())
    # batch independence: sample 1 alone gives what it gives inside the batch of 2 (a coupling bug through the token-axis concat / fused
    # q/k/v views gives O(1); bf16 accumulation-order differences between the two GEMM plans give ~1e-2 after 24 blocks)
    solo = teacher.forward(x[1:], t[1:], c[1:], p[1:])
    assert rel(solo[0], out[1]) < 5e-2, rel(solo[0], out[1])
    assert rel(out[0], out[1]) > 0.5                                   # (the two samples really are different)
    stu = MMDiT(W, lora).forward(x, t, c, p)                           # LoRA with B = 0 (peft init): the student is the teacher
    assert rel(stu, out) < 5e-2, rel(stu, out)
    D = SD3Distiller(W, lora, SD3StepConfig(multiphase=2, num_euler_timesteps=100, learning_rate=5e-6, adam_weight_decay=1e-3))
    uc, up, noise = r(2, Lc, cfg.joint_attention_dim), r(2, cfg.pooled_projection_dim), r(2, cfg.in_channels, hw, hw)
    p0 = lora.params.clone()
    res = D.step(x, c, p, uc, up, noise, torch.tensor([7, 93], device=dev))
    assert bool(torch.isfinite(res["loss"]).all()) and float(res["loss"]) > 0
    assert float(res["grad_sumsq"]) > 0 and not torch.equal(lora.params, p0)
    m = lora.modules["transformer_blocks.0.attn.to_q"]
    rr = lora.real_rank
    assert float(m.gB.abs().max()) > 0                                           # dB = s dy^T (x A^T) is non-zero even with B = 0
    assert float(m.gA[rr:].abs().max()) == 0.0 and float(m.A[rr:].abs().max()) == 0.0     # the rank padding stays inert
    return float(res["loss"]), float(res["grad_sumsq"]) ** 0.5
