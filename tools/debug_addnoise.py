import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
import torch
from safetensors.torch import load_file
from pcm_amd import ops
g = load_file(os.path.join(ROOT, "tests/golden/pcm_math_golden.safetensors"))
acp = g["alphas_cumprod"].cuda(); x = g["x"].cuda(); n = g["noise"].cuda(); t = g["start_timesteps"].cuda()
out = ops.add_noise(x, n, acp, t).cpu()
ref = g["add_noise_fp32"]
bad = (out != ref)
print("mismatch", int(bad.sum()), "of", bad.numel(), "max diff", float((out-ref).abs().max()))
# isolate: x = 0 -> sb*noise ; noise = 0 -> sa*x
z = torch.zeros_like(x)
a = g["alphas_cumprod"][g["start_timesteps"]]
sa = (a ** 0.5).view(-1,1,1,1); sb = ((1 - a) ** 0.5).view(-1,1,1,1)
o1 = ops.add_noise(x, z, acp, t).cpu(); print("sa*x mismatch", int((o1 != sa * g["x"]).sum()))
o2 = ops.add_noise(z, n, acp, t).cpu(); print("sb*n mismatch", int((o2 != sb * g["noise"]).sum()))
ones = torch.ones_like(x)
o3 = ops.add_noise(ones, z, acp, t).cpu(); print("sa mismatch", int((o3[:,0,0,0] != sa.view(-1)).sum()), o3[:,0,0,0][:4], sa.view(-1)[:4])
print("torch gpu pow vs cpu:", int(((acp[t] ** 0.5).cpu() != a ** 0.5).sum()), "gpu sqrt vs cpu", int((torch.sqrt(acp[t]).cpu() != torch.sqrt(a)).sum()))
