"""The north star's own pass on its own: the two-timestep student forward (online + target as ONE 2B-sample LoRA pass, activations saved for
the backward) at the BASELINE configs[1] size, run REPS times after one warm-up -- for `rocprofv3 --kernel-trace` (per-kernel table of the
forward alone: tools/prof_summary.py) and for event timing of library A/Bs.

    python tools/fwd2t_trace.py [--reps 5] [--batch 16] [--lib path/to/other/libpcm_hip.so] [--json out.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pcm_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--json", default=None)
    ap.add_argument("--frozen", action="store_true", help="also time the frozen (teacher) 2B pass")
    args = ap.parse_args()
    if args.lib:
        capi.set_lib(capi.Lib(os.path.abspath(args.lib)))
    from pcm_amd.model import LoraState, UNet, UNetWeights
    from pcm_amd.unet_spec import UNetConfig, random_state_dict
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ucfg = UNetConfig.sd15()
    with torch.no_grad():
        sd = random_state_dict(ucfg, seed=0, device=dev)
        W = UNetWeights(ucfg, sd, dev)
        del sd
        lora = LoraState(ucfg, 64, 8.0, dev, seed=1, b_std=0.02)
    student, teacher = UNet(W, lora), UNet(W, None)
    B = args.batch
    g = torch.Generator(device=dev).manual_seed(453645634)
    x2 = torch.randn(2 * B, 4, 64, 64, generator=g, device=dev)
    t2 = torch.randint(0, 1000, (2 * B,), generator=g, device=dev)
    c2 = torch.randn(2 * B, 77, 768, generator=g, device=dev)
    out = {}
    for name, fn in (("student_fwd_2t", lambda: student.forward(x2, t2, c2, save=True, save_half=True)),
                     ("teacher_2b", lambda: teacher.forward(x2, t2, c2, dup_halves=False))):
        if name == "teacher_2b" and not args.frozen:
            continue
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        out[name] = round(ms, 3)
        print("%s: %.3f ms (eager, %d reps)" % (name, ms, args.reps), flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f)


if __name__ == "__main__":
    main()
