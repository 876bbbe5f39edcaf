import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd")); sys.path.insert(0, ROOT)
import torch
from pcm_amd import capi
from pcm_amd.model import LoraState, UNetWeights
from pcm_amd.trainer import Distiller, StepConfig
from pcm_amd.unet_spec import UNetConfig, random_state_dict
capi.lib()
dev = "cuda"
kw = dict(block_out_channels=(64, 128, 128, 128), cross_attention_dim=64, heads=2, norm_num_groups=32)
cfg = UNetConfig(**kw) if "--tiny" in sys.argv else UNetConfig.sd15()
hw, cd = (16, 64) if "--tiny" in sys.argv else (64, 768)
W = UNetWeights(cfg, random_state_dict(cfg, 0, dev), dev)
scfg = StepConfig(multiphase=4, loss_type="huber", learning_rate=5e-6, adam_weight_decay=1e-3, w_min=4.0, w_max=5.0)
lora = LoraState(cfg, 64, 8.0, dev, seed=1, b_std=0.02)
D = Distiller(W, lora, scfg)
B = 4
g = torch.Generator(device=dev).manual_seed(1)
inp = dict(latents=torch.randn(B, 4, hw, hw, generator=g, device=dev), prompt_embeds=torch.randn(B, 77, cd, generator=g, device=dev),
           uncond_prompt_embeds=torch.randn(B, 77, cd, generator=g, device=dev), noise=torch.randn(B, 4, hw, hw, generator=g, device=dev),
           index=torch.randint(0, 50, (B,), generator=g, device=dev), w=4.0 + torch.rand(B, generator=g, device=dev))
D.capture(B, H=hw, W=hw, ctx_dim=cd)
p0 = [t.clone() for t in (lora.params, lora.exp_avg, lora.exp_avg_sq, D.step_dev)]
def restore():
    for dst, src in zip((lora.params, lora.exp_avg, lora.exp_avg_sq, D.step_dev), p0): dst.copy_(src)
    lora.repack()
D.step(**inp); torch.cuda.synchronize()
pe, ge, sq_e, st_e = lora.params.clone(), lora.grads.clone(), lora.gradsq.clone(), D.step_dev.clone()
restore()
D.step_graphed(**inp); torch.cuda.synchronize()
pg, gg, sq_g, st_g = lora.params.clone(), lora.grads.clone(), lora.gradsq.clone(), D.step_dev.clone()
ue, ug = (pe - p0[0]).double(), (pg - p0[0]).double()
print("step_dev eager/graph", st_e.item(), st_g.item(), "gradsq", sq_e.item(), sq_g.item())
print("grad rel", float((gg - ge).norm() / ge.norm()), "|ue|", float(ue.norm()), "|ug|", float(ug.norm()), "cos", float((ue * ug).sum() / (ue.norm() * ug.norm())))
print("ue stats", float(ue.abs().max()), float(ue.abs().mean()), " ug stats", float(ug.abs().max()), float(ug.abs().mean()))
print("sign agree", float((torch.sign(ue) == torch.sign(ug)).double().mean()))
decay = (-5e-6 * 1e-3 * p0[0]).double()
print("cos(ue, decay)", float((ue * decay).sum() / (ue.norm() * decay.norm())), "cos(ug, decay)", float((ug * decay).sum() / (ug.norm() * decay.norm())))
restore()
D.step_graphed(**inp); torch.cuda.synchronize()
ug2 = (lora.params - p0[0]).double()
print("second graphed replay vs first: rel", float((ug2 - ug).norm() / ug.norm()))
print("replay2: gradsq", lora.gradsq.item(), "grad rel vs eager", float((lora.grads - ge).norm() / ge.norm()), "step_dev", D.step_dev.item(), "loss", float(D._static_out["loss"].item()))
print("m norm", float(lora.exp_avg.norm()), "v norm", float(lora.exp_avg_sq.norm()), "p0 m norm", float(p0[1].norm()))
restore()
D._g_fb.replay(); torch.cuda.synchronize()
print("fb-only replay3: grad rel vs eager", float((lora.grads - ge).norm() / ge.norm()))
D._g_opt.replay(); torch.cuda.synchronize()
ug3 = (lora.params - p0[0]).double()
print("opt replay3: rel vs first", float((ug3 - ug).norm() / ug.norm()), "gradsq", lora.gradsq.item(), "step_dev", D.step_dev.item())
restore()
D._g_fb.replay(); torch.cuda.synchronize()
bad = []
for pth, m in lora.modules.items():
    ga, gb = m.gA.double(), m.gB.double()
    oa, ob = int((m.gA.data_ptr() - lora.grads.data_ptr()) // 4), int((m.gB.data_ptr() - lora.grads.data_ptr()) // 4)
    ra = float((ga - ge[oa:oa + ga.numel()].view_as(ga).double()).norm() / (ge[oa:oa + ga.numel()].double().norm() + 1e-30))
    rb = float((gb - ge[ob:ob + gb.numel()].view_as(gb).double()).norm() / (ge[ob:ob + gb.numel()].double().norm() + 1e-30))
    if not (ra < 1e-3 and rb < 1e-3): bad.append((pth, ra, rb))
print("modules with wrong grads in replay:", len(bad), "of", len(lora.modules))
for b_ in bad[:12]: print("  ", b_)
for k in ("noise_pred", "target", "model_pred"): print(k, "finite", bool(torch.isfinite(D._static_out[k]).all()))
