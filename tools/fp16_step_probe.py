"""One SD1.5-size distillation step (bs 2, 2 phases: BASELINE configs[0]) through the IEEE-half build of the library
(precision.set_precision("fp16"), lib/libpcm_hip_f16.so) against the committed fp32-oracle fixture, beside the bf16 build on the same inputs.
usage: python tools/fp16_step_probe.py [out.json]"""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "phased-consistency-model_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import step_golden_cases as S  # noqa: E402
from golden_fixture import golden, sk_cos, sk_rel, sketch  # noqa: E402
from oracle import unet_sd15 as O  # noqa: E402
from pcm_amd import precision  # noqa: E402
from pcm_amd.model import LoraState, UNetWeights  # noqa: E402
from pcm_amd.trainer import Distiller  # noqa: E402
from pcm_amd.unet_spec import UNetConfig  # noqa: E402


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def run(prec, b_std, sd):
    precision.set_precision(prec)
    W = UNetWeights(UNetConfig.sd15(), sd, "cuda")
    ref = golden(S.step_name(b_std), lambda: S.ref_sd15_step(b_std))
    inp = S.step_inputs()
    lora = LoraState(UNetConfig.sd15(), 64, 8.0, "cuda", seed=1, b_std=b_std)
    p_before = S.lora_flat(lora, "p")
    _, cfg = S.step_cfgs(2)
    D = Distiller(W, lora, cfg)
    dev = {k: v.cuda() for k, v in inp.items()}
    out = D.step(dev["latents"], dev["prompt_embeds"], dev["uncond_prompt_embeds"], dev["noise"], dev["index"], dev["w"])
    torch.cuda.synchronize()
    scale = 65536.0 if prec == "fp16" else 1.0
    rep = {k: rel(out[k], ref[k]) for k in S.KEYS7}
    loss, rloss = float(out["loss"].item()), float(ref["loss"])
    rep["loss_rel"] = abs(loss - rloss) / abs(rloss)
    gn = math.sqrt(float(out["grad_sumsq"].item())) / scale
    rep["grad_norm_rel"] = abs(gn - float(ref["grad_norm"])) / float(ref["grad_norm"])
    rep["grad_rel"] = sk_rel(sketch(S.lora_flat(lora, "g") / scale), ref["sk_grad"])
    p_after = S.lora_flat(lora, "p")
    rep["param_rel"] = sk_rel(sketch(p_after), ref["sk_param_after"])
    rep["update_cos"] = sk_cos(sketch(p_after - p_before), ref["sk_update"])
    rep["finite"] = bool(torch.isfinite(out["noise_pred"].float()).all()) and math.isfinite(gn)
    if D.loss_scale_dev is not None:
        rep["loss_scale_after"] = float(D.loss_scale_dev.item())
    rep.update(loss=loss, oracle_loss=rloss, precision=prec, b_std=b_std)
    del D, W, lora
    torch.cuda.empty_cache()
    return rep


if __name__ == "__main__":
    sd = O.init_state_dict(O.UNetConfig.sd15(), 0)
    res = []
    for b_std in (0.0, 0.02):
        for prec in ("bf16", "fp16"):
            r = run(prec, b_std, sd)
            res.append(r)
            print(json.dumps({k: (float("%.4g" % v) if isinstance(v, float) else v) for k, v in r.items()}), flush=True)
    precision.set_precision("bf16")
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(res, f, indent=1)
