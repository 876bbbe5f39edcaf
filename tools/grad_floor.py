"""One-off evidence at the real SD1.5 size (bs 2): the LoRA gradient of the HIP step against the rounding-point-matched oracle, with the
oracle's own fp64-vs-fp32 floor and the plain fp32 oracle beside it (tests/rounding_matched_cases.py::case_step_floor, with_grads=True).
Too slow for the GPU suite (three oracle backward passes, one of them in fp64).  Writes gpurun_out/lora_grad_floor_sd15.json.
TEST / EVIDENCE TOOL: imports oracle/ (allowed for tools and tests only)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "phased-consistency-model_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from pcm_amd import capi  # noqa: E402
capi.lib()
import rounding_matched_cases as R  # noqa: E402
kw = dict(block_out_channels=(320, 640, 1280, 1280), cross_attention_dim=768, heads=8, norm_num_groups=32)
t0 = time.time()
rep = {}
R.case_step_floor("cuda", kw, 2, 64, 768, index=[13, 37], report=rep, with_fp32=True, with_grads=True)
rep["seconds"] = time.time() - t0
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "lora_grad_floor_sd15.json"), "w"), indent=1)
print(json.dumps(rep["lora_grad"]), "%.0f s" % rep["seconds"])
