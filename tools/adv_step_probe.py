"""BASELINE.json configs[2] on one GPU: SD1.5 PCM-LoRA + the latent discriminator (9 tapped features x 4 heads = 36 heads, 663.8 M head
parameters), 2 phases, per-GPU batch 8 -- times a discriminator step and a generator step (eager launches), counts launches.
python tools/adv_step_probe.py [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import capi
from pcm_amd.discriminator import ADAPTER_DIMS, Discriminator
from pcm_amd.model import LoraState, UNetWeights
from pcm_amd.trainer import AdvDistiller, StepConfig
from pcm_amd.unet_spec import UNetConfig, random_state_dict
L = capi.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
cfg = UNetConfig.sd15()
W = UNetWeights(cfg, random_state_dict(cfg, 0, dev), dev)
lora = LoraState(cfg, 64, 8.0, dev, seed=1, b_std=0.01)
disc = Discriminator(ADAPTER_DIMS, num_h_per_head=4, device=dev, seed=2)
print("heads %d, head params %.1f M, %.1f GB allocated" % (len(disc.heads), disc.numel / 1e6, torch.cuda.memory_allocated() / 1e9), flush=True)
D = AdvDistiller(W, lora, StepConfig(multiphase=2, loss_type="huber", learning_rate=5e-6, adam_weight_decay=1e-3, w_min=4.0, w_max=5.0), disc, adv_weight=0.1, adv_lr=1e-5)
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, **k: torch.randn(*s, generator=g, device=dev, **k)
# count C-ABI launches per step
calls = [0]
orig = capi.Lib.call
def counted(self, name, *a):
    calls[0] += 1
    return orig(self, name, *a)
capi.Lib.call = counted
res = {}
for gs in (0, 1, 2, 3, 4, 5):
    lat, pe, un, nz = rn(B, 4, 64, 64), rn(B, 77, 768), rn(B, 77, 768), rn(B, 4, 64, 64)
    idx = torch.randint(0, 50, (B,), generator=g, device=dev); w = 4.0 + torch.rand(B, generator=g, device=dev)
    nf, nr, au = rn(B, 4, 64, 64), rn(B, 4, 64, 64), torch.rand(B, generator=g, device=dev)
    calls[0] = 0
    torch.cuda.synchronize(); t0 = time.time()
    out = D.step_adv(gs, lat, pe, un, nz, idx, w, nf, nr, au)
    torch.cuda.synchronize(); dt = 1e3 * (time.time() - t0)
    kind = "D" if gs % 2 == 0 else "G"
    res.setdefault(kind, []).append(dt)
    print("global_step %d (%s step): %.1f ms, %d C-ABI calls, %s, peak %.1f GB" % (gs, kind, dt, calls[0],
          ("d_loss %.4f" % float(out["d_loss"])) if kind == "D" else ("loss_cm %.5f g_loss %.4f" % (float(out["loss_cm"]), float(out["g_loss"]))),
          torch.cuda.max_memory_allocated() / 1e9), flush=True)
d, gg = min(res["D"][1:]), min(res["G"][1:])
print("bs %d: D step %.1f ms, G step %.1f ms -> %.2f images/sec per GPU over a D+G pair (one student update per pair; eager launches)" % (B, d, gg, 2 * B / ((d + gg) * 1e-3)))
if len(sys.argv) > 2 and sys.argv[2] == "graph":     # the same D / G steps through hipGraph replay (AdvDistiller.capture_adv)
    capi.Lib.call = orig
    D.capture_adv(B)
    print("captured, %.1f GB allocated, peak %.1f GB" % (torch.cuda.memory_allocated() / 1e9, torch.cuda.max_memory_allocated() / 1e9), flush=True)
    res = {}
    for gs in (0, 1, 2, 3, 4, 5):
        lat, pe, un, nz = rn(B, 4, 64, 64), rn(B, 77, 768), rn(B, 77, 768), rn(B, 4, 64, 64)
        idx = torch.randint(0, 50, (B,), generator=g, device=dev); w = 4.0 + torch.rand(B, generator=g, device=dev)
        nf, nr, au = rn(B, 4, 64, 64), rn(B, 4, 64, 64), torch.rand(B, generator=g, device=dev)
        torch.cuda.synchronize(); t0 = time.time()
        out = D.step_adv_graphed(gs, lat, pe, un, nz, idx, w, nf, nr, au)
        torch.cuda.synchronize(); dt = 1e3 * (time.time() - t0)
        kind = "D" if gs % 2 == 0 else "G"
        res.setdefault(kind, []).append(dt)
        print("graphed global_step %d (%s step): %.1f ms, %s" % (gs, kind, dt, ("d_loss %.4f" % float(out["d_loss"])) if kind == "D" else
              ("loss_cm %.5f g_loss %.4f" % (float(out["loss_cm"]), float(out["g_loss"])))), flush=True)
    d, gg = min(res["D"][1:]), min(res["G"][1:])
    print("bs %d, hipGraph replay: D step %.1f ms, G step %.1f ms -> %.2f images/sec per GPU over a D+G pair" % (B, d, gg, 2 * B / ((d + gg) * 1e-3)))
