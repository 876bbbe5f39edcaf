"""How repeatable is the LoRA gradient of the benchmarked step (SD1.5, bs 16, 4 phases) run to run?  Three eager forward+backward
passes on identical inputs with the fast reductions (fp32 / fp64 atomics) and three with the reproducible forms (ops.set_deterministic):
rel-L2 between runs overall and for the worst modules.  The reproducible forms must give exactly 0 -- anything else is a race or an
uninitialised read, not summation order.   usage: grad_repeatability.py [batch]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd")); sys.path.insert(0, ROOT)
import torch
from oracle import unet_sd15 as O
from pcm_amd import capi, ops
from pcm_amd.model import LoraState, UNetWeights
from pcm_amd.trainer import Distiller, StepConfig
from pcm_amd.unet_spec import UNetConfig
capi.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = UNetConfig.sd15()
W = UNetWeights(cfg, O.init_state_dict(O.UNetConfig.sd15(), 0), "cuda")
lora = LoraState(cfg, 64, 8.0, "cuda", seed=1, b_std=0.02)
D = Distiller(W, lora, StepConfig(multiphase=4, loss_type="huber", learning_rate=5e-6, adam_weight_decay=1e-3, w_min=4.0, w_max=5.0))
g = torch.Generator(device="cuda").manual_seed(453645634)
r = lambda *s: torch.randn(*s, generator=g, device="cuda")   # noqa: E731
inp = dict(latents=r(B, 4, 64, 64), prompt_embeds=r(B, 77, 768), uncond_prompt_embeds=r(B, 77, 768), noise=r(B, 4, 64, 64),
           index=torch.randint(0, 50, (B,), generator=g, device="cuda"), w=4 + torch.rand(B, generator=g, device="cuda"))
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))   # noqa: E731
for det in (False, True):
    ops.set_deterministic(det)
    runs = []
    for i in range(3):
        out = D.forward_backward(**inp)
        torch.cuda.synchronize()
        runs.append((float(out["loss"].item()), lora.grads.clone(), out["noise_pred"].clone()))
    ops.set_deterministic(False)
    print("deterministic" if det else "atomics      ", "loss", [x[0] for x in runs], "eps rel", rel(runs[1][2], runs[0][2]), rel(runs[2][2], runs[0][2]),
          "| grad rel run1/run0 %.3e run2/run0 %.3e" % (rel(runs[1][1], runs[0][1]), rel(runs[2][1], runs[0][1])), flush=True)
    worst = []
    for p, m in lora.modules.items():
        for nm, gbuf in (("A", m.gA), ("B", m.gB)):
            off, n = gbuf.storage_offset(), gbuf.numel()
            a, b = runs[1][1][off:off + n], runs[0][1][off:off + n]
            worst.append((rel(a, b), p + "." + nm, float(b.double().norm())))
    worst.sort(reverse=True)
    for w_ in worst[:6]:
        print("    %.3e  %-70s |g| %.3e" % w_)
