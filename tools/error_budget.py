"""Where does the bf16 HIP path's deviation from the fp32 oracle come from, and how does it compare with what the REFERENCE's own
mixed-precision run would show?   python tools/error_budget.py [--tiny] [--bf16-oracle]

(1) per-block rel-L2 of the HIP UNet (bf16 MFMA, fp32 accumulate) against the fp32 oracle at the 9 feature taps of the reference's
    modified_forward (discriminator_sd15.py:264-342: after each down block, the mid block, each up block) and at eps;
(2) the same quantities for the ORACLE ITSELF under torch.autocast(bfloat16) -- the arithmetic the reference trains with
    (train_pcm_lora_sd15.py:1034 accelerate mixed precision; :1262 autocast(weight_dtype)) -- against the fp32 oracle;
(3) the loss of one full distillation step (huber) for: fp32 oracle, bf16-autocast oracle, HIP path.
Writes gpurun_out/error_budget[_tiny].json.  TEST/EVIDENCE TOOL: imports oracle/ (allowed for tools and tests only)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "phased-consistency-model_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from oracle import pcm_step as OS  # noqa: E402
from oracle import unet_sd15 as O  # noqa: E402
from pcm_amd import capi  # noqa: E402
from pcm_amd.model import LoraState, UNet, UNetWeights  # noqa: E402
from pcm_amd.trainer import Distiller, StepConfig  # noqa: E402
from pcm_amd.unet_spec import UNetConfig  # noqa: E402


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    tiny = "--tiny" in sys.argv
    capi.lib()
    dev = "cuda"
    if tiny:
        kw = dict(block_out_channels=(64, 128, 128, 128), cross_attention_dim=64, heads=2, norm_num_groups=32)
        oc, pc, hw, ctx_dim = O.UNetConfig(**kw), UNetConfig(**kw), 16, 64
    else:
        oc, pc, hw, ctx_dim = O.UNetConfig.sd15(), UNetConfig.sd15(), 64, 768
    sd = O.init_state_dict(oc, 0)
    W = UNetWeights(pc, sd, dev)
    B = 2
    ocfg = OS.StepConfig(multiphase=2, loss_type="huber", lr=5e-6, adam_weight_decay=1e-3, w_min=4.0, w_max=5.0)
    inp = OS.draw_inputs(B, ocfg, seed=453645634, latent_hw=hw, ctx_dim=ctx_dim)
    inp["index"] = torch.tensor([13, 37])
    lora = LoraState(pc, 64, 8.0, dev, seed=1, b_std=0.02)
    olora = {p: (lora.A_peft(m).detach().cpu().clone(), m.B.detach().cpu().clone()) for p, m in lora.modules.items()}
    x, t, ctx = inp["latents"], torch.tensor([259, 759]), inp["prompt_embeds"]
    rep = {"config": "tiny" if tiny else "sd15", "B": B}
    t0 = time.time()
    with torch.no_grad():
        f32 = O.unet_forward(oc, sd, x, t, ctx, olora, 8.0, return_features=True) + [O.unet_forward(oc, sd, x, t, ctx, olora, 8.0)]
        rep["oracle_fp32_s"] = time.time() - t0
        t0 = time.time()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            b16 = O.unet_forward(oc, sd, x, t, ctx, olora, 8.0, return_features=True) + [O.unet_forward(oc, sd, x, t, ctx, olora, 8.0)]
        rep["oracle_bf16_autocast_s"] = time.time() - t0
    st = UNet(W, lora)
    feats = st.forward(x.to(dev), t.to(dev), ctx.to(dev), features=True)
    eps = st.forward(x.to(dev), t.to(dev), ctx.to(dev))
    names = ["down0", "down1", "down2", "down3", "mid", "up0", "up1", "up2", "up3"][:len(f32) - 1] + ["eps"]
    if len(f32) - 1 != 9:
        names = ["tap%d" % i for i in range(len(f32) - 1)] + ["eps"]
    hip = []
    for (h, Hh, Ww) in feats:
        hip.append(h.float().view(B, Hh, Ww, -1).permute(0, 3, 1, 2))
    hip.append(eps)
    rep["per_block"] = {n: {"hip_vs_fp32": rel(a, r), "ref_bf16_autocast_vs_fp32": rel(b.float(), r)} for n, a, b, r in zip(names, hip, b16, f32)}
    # one full step: loss for the three arithmetics
    ref = OS.distill_step_forward(oc, sd, {k: (a.clone(), b.clone()) for k, (a, b) in olora.items()}, inp, ocfg)
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        ref16 = OS.distill_step_forward(oc, sd, olora, inp, ocfg)
    cfg = StepConfig(multiphase=2, loss_type="huber", learning_rate=5e-6, adam_weight_decay=1e-3, w_min=4.0, w_max=5.0)
    D = Distiller(W, lora, cfg)
    d = {k: v.to(dev) for k, v in inp.items()}
    out = D.forward_backward(d["latents"], d["prompt_embeds"], d["uncond_prompt_embeds"], d["noise"], d["index"], d["w"], backward=False)
    lf, l16, lh = float(ref["loss"]), float(ref16["loss"]), float(out["loss"].item())
    rep["loss"] = {"oracle_fp32": lf, "oracle_bf16_autocast": l16, "hip": lh, "hip_rel": abs(lh - lf) / abs(lf), "ref_bf16_autocast_rel": abs(l16 - lf) / abs(lf)}
    for k in ("noise_pred", "x_prev", "target_noise_pred", "model_pred", "target"):
        rep["step_" + k] = {"hip_vs_fp32": rel(out[k], ref[k].detach()), "ref_bf16_autocast_vs_fp32": rel(ref16[k].float(), ref[k].detach())}
    # sensitivity of the loss: d loss for a relative perturbation of eps at the bf16 level (random sign, 2^-9)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", "error_budget%s.json" % ("_tiny" if tiny else ""))
    json.dump(rep, open(path, "w"), indent=1)
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
