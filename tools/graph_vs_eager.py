"""Where does the captured-graph forward+backward differ from the eager one?  (tests/test_gpu_bench_config.py asserts grad rel < 1e-5.)
Eager twice, capture, replay twice, eager again: loss / eps / gradient rel-L2 between every pair, worst LoRA modules of graph vs eager;
then the same with the reproducible reductions (ops.set_deterministic), where every pair must be bitwise equal."""
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
if len(sys.argv) > 2:      # another build of the library (tools/probes/build_variant.py), e.g. the register-staged attention kernels
    capi.set_lib(capi.Lib(os.path.abspath(sys.argv[2])))
capi.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = UNetConfig.sd15()
W = UNetWeights(cfg, O.init_state_dict(O.UNetConfig.sd15(), 0), "cuda")
g = torch.Generator(device="cuda").manual_seed(453645634)
r = lambda *s: torch.randn(*s, generator=g, device="cuda")   # noqa: E731
inp = dict(latents=r(B, 4, 64, 64), prompt_embeds=r(B, 77, 768), uncond_prompt_embeds=r(B, 77, 768), noise=r(B, 4, 64, 64),
           index=torch.randint(0, 50, (B,), generator=g, device="cuda"), w=4 + torch.rand(B, generator=g, device="cuda"))
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))   # noqa: E731
for det in (False, True):
    ops.set_deterministic(det)
    lora = LoraState(cfg, 64, 8.0, "cuda", seed=1, b_std=0.02)
    D = Distiller(W, lora, StepConfig(multiphase=4, loss_type="huber", learning_rate=5e-6, adam_weight_decay=1e-3, w_min=4.0, w_max=5.0))
    runs = {}

    def eager(tag):
        out = D.forward_backward(**inp)
        torch.cuda.synchronize()
        runs[tag] = (float(out["loss"].item()), lora.grads.clone(), out["noise_pred"].clone())
    eager("eager0"); eager("eager1")
    D.capture(B)
    for k, v in inp.items():
        D._static[k].copy_(v)
    for tag in ("graph0", "graph1"):
        lora.grads.fill_(float("nan"))
        D._g_fb.replay()
        torch.cuda.synchronize()
        runs[tag] = (float(D._static_out["loss"].item()), lora.grads.clone(), D._static_out["noise_pred"].clone())
    eager("eager2")
    ops.set_deterministic(False)
    print("== reproducible reductions" if det else "== atomics")
    for a in ("eager1", "graph0", "graph1", "eager2"):
        print("  %s vs eager0: loss %.3e eps %.3e grad %.3e" % (a, abs(runs[a][0] - runs["eager0"][0]) / runs["eager0"][0], rel(runs[a][2], runs["eager0"][2]),
                                                              rel(runs[a][1], runs["eager0"][1])), flush=True)
    print("  graph1 vs graph0: grad %.3e" % rel(runs["graph1"][1], runs["graph0"][1]))
    worst = []
    for p, m in lora.modules.items():
        for nm, gbuf in (("A", m.gA), ("B", m.gB)):
            off, n = gbuf.storage_offset(), gbuf.numel()
            worst.append((rel(runs["graph0"][1][off:off + n], runs["eager0"][1][off:off + n]), p + "." + nm))
    worst.sort(reverse=True)
    for w_ in worst[:8]:
        print("    %.3e  %s" % w_)
    del D, lora
    torch.cuda.empty_cache()
