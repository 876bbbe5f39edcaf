"""Full-size SDXL (2.57 B-parameter UNet, 128x128x4 latents, LoRA r=64, 40 DDIM steps, 4 phases) PCM distillation step on one MI355X with
random-init weights and synthetic conditioning: checks that every layer shape of BASELINE.json configs[3] runs and times the step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from pcm_amd import capi
from pcm_amd.model import LoraState, UNetWeights
from pcm_amd.trainer import Distiller, StepConfig
from pcm_amd.unet_spec import UNetConfig, random_state_dict
capi.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda", 0)
cfg = UNetConfig.sdxl()
t0 = time.time()
sd = random_state_dict(cfg, 0, dev)
W = UNetWeights(cfg, sd, dev)
del sd
lora = LoraState(cfg, 64, 8.0, dev, seed=1, b_std=0.01)
print("SDXL UNet packed in %.1f s, %.1f GB allocated, LoRA modules %d / %.1f M params" % (time.time() - t0, torch.cuda.memory_allocated() / 1e9, len(lora.modules), lora.params.numel() / 1e6), flush=True)
D = Distiller(W, lora, StepConfig(multiphase=4, num_ddim_timesteps=40, w_min=6.0, w_max=7.0, learning_rate=2e-6, adam_weight_decay=0.0))
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device=dev)
tids = torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * B, device=dev)
ac = dict(text_embeds=rn(B, 1280), time_ids=tids)
uac = dict(text_embeds=torch.zeros(B, 1280, device=dev), time_ids=tids)
for it in range(3):
    lat, pe, un, nz = rn(B, 4, 128, 128), rn(B, 77, 2048), torch.zeros(B, 77, 2048, device=dev), rn(B, 4, 128, 128)
    idx = torch.randint(0, 40, (B,), generator=g, device=dev)
    w = 6.0 + torch.rand(B, generator=g, device=dev)
    torch.cuda.synchronize(); t1 = time.time()
    out = D.step(lat, pe, un, nz, idx, w, added_cond=ac, uncond_added_cond=uac)
    torch.cuda.synchronize()
    print("step %d: %.1f ms, loss %.5f, grad sumsq %.3e, peak %.1f GB" % (it, 1e3 * (time.time() - t1), float(out["loss"]), float(out["grad_sumsq"]), torch.cuda.max_memory_allocated() / 1e9), flush=True)
    assert torch.isfinite(out["loss"]).all()
print("images/sec (eager launches, bs %d): %.2f" % (B, B / (time.time() - t1)))
if len(sys.argv) > 2 and sys.argv[2] == "graph":      # the same step through hipGraph replay (Distiller.capture with the text_time inputs static)
    D.capture(B, H=128, W=128, ctx_len=77, ctx_dim=2048, added_cond=ac, uncond_added_cond=uac)
    for it in range(3):
        lat, pe, un, nz = rn(B, 4, 128, 128), rn(B, 77, 2048), torch.zeros(B, 77, 2048, device=dev), rn(B, 4, 128, 128)
        idx = torch.randint(0, 40, (B,), generator=g, device=dev)
        w = 6.0 + torch.rand(B, generator=g, device=dev)
        torch.cuda.synchronize(); t1 = time.time()
        out = D.step_graphed(lat, pe, un, nz, idx, w, added_cond=ac, uncond_added_cond=uac)
        torch.cuda.synchronize()
        print("graphed step %d: %.1f ms, loss %.5f" % (it, 1e3 * (time.time() - t1), float(out["loss"])), flush=True)
        assert torch.isfinite(out["loss"]).all()
    print("images/sec (hipGraph replay, bs %d): %.2f" % (B, B / (time.time() - t1)))
