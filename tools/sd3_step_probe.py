"""Full-size SD3-medium (2.03 B-parameter MMDiT, 128x128x16 latents -> 4096 image tokens + 154 text tokens, LoRA r=32) PCM distillation
step on one MI355X with random-init weights and synthetic conditioning: checks that every layer shape of BASELINE.json configs[4] runs
and times the step (eager launches).   python tools/sd3_step_probe.py [batch] [adv|graph]
"adv": the adversarial trainer (22-entry LoRA list + 24 discriminator heads), one discriminator and one generator step per iteration."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import capi
from pcm_amd.mmdit import MMDiTWeights, sd3_lora_state
from pcm_amd.mmdit_spec import LORA_TARGETS_SD3_ADV, MMDiTConfig, random_state_dict
from pcm_amd.trainer_sd3 import SD3AdvDistiller, SD3Distiller, SD3StepConfig
capi.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
adv = len(sys.argv) > 2 and sys.argv[2] == "adv"
dev = torch.device("cuda", 0)
cfg = MMDiTConfig.sd3_medium()
t0 = time.time()
sd = random_state_dict(cfg, 0, dev)
W = MMDiTWeights(cfg, sd, dev)
del sd
lora = sd3_lora_state(cfg, 32, 8.0, dev, seed=1, b_std=0.01, **(dict(targets=LORA_TARGETS_SD3_ADV, init="kaiming") if adv else {}))
print("MMDiT packed in %.1f s, %.1f GB allocated, LoRA modules %d / %.1f M params (stored padded to rank 64)" % (
    time.time() - t0, torch.cuda.memory_allocated() / 1e9, len(lora.modules), lora.params.numel() / 1e6), flush=True)
scfg = SD3StepConfig(multiphase=2, num_euler_timesteps=100, learning_rate=5e-6, adam_weight_decay=1e-3)      # run.sh "2phases" recipe
if adv:
    from pcm_amd.discriminator import Discriminator
    D = SD3AdvDistiller(W, lora, scfg, Discriminator([cfg.inner_dim] * cfg.num_layers, num_h_per_head=1, device=dev, seed=2, ksize=1))
else:
    D = SD3Distiller(W, lora, scfg)
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, **k: torch.randn(*s, generator=g, device=dev, **k)
gstep = 0
for it in range(3):
    lat, nz = rn(B, 16, 128, 128), rn(B, 16, 128, 128)
    pe, pp, un, unp = rn(B, 154, 4096), rn(B, 2048), rn(B, 154, 4096), rn(B, 2048)
    idx = torch.randint(0, 100, (B,), generator=g, device=dev)
    torch.cuda.synchronize(); t1 = time.time()
    if adv:
        for _ in range(2):
            out = D.step_adv(gstep, lat, pe, pp, un, unp, nz, idx, rn(B, 16, 128, 128, dtype=torch.float64), rn(B, 16, 128, 128, dtype=torch.float64),
                             torch.rand(B, generator=g, device=dev))
            gstep += 1
        torch.cuda.synchronize()
        print("iter %d (D + G step): %.1f ms, loss_cm %.5f g_loss %.5f, peak %.1f GB" % (it, 1e3 * (time.time() - t1), float(out["loss_cm"]),
              float(out["g_loss"]), torch.cuda.max_memory_allocated() / 1e9), flush=True)
    else:
        out = D.step(lat, pe, pp, un, unp, nz, idx)
        torch.cuda.synchronize()
        print("step %d: %.1f ms, loss %.5f, grad sumsq %.3e, peak %.1f GB" % (it, 1e3 * (time.time() - t1), float(out["loss"]),
              float(out["grad_sumsq"]), torch.cuda.max_memory_allocated() / 1e9), flush=True)
        assert torch.isfinite(out["loss"]).all()
print("images/sec (eager launches, bs %d%s): %.2f" % (B, ", D+G pair" if adv else "", B / (time.time() - t1)))
if not adv and len(sys.argv) > 2 and sys.argv[2] == "graph":
    try:
        D.capture(B)
        for it in range(3):
            torch.cuda.synchronize(); t1 = time.time()
            out = D.step_graphed(lat, pe, pp, un, unp, nz, idx)
            torch.cuda.synchronize()
            print("graphed step %d: %.1f ms, loss %.5f" % (it, 1e3 * (time.time() - t1), float(out["loss"])), flush=True)
    except Exception as e:       # first hardware run of this path
        print("hipGraph capture of the SD3 step failed:", repr(e)[:300])
