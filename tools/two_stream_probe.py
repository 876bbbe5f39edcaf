"""Do the step's two INDEPENDENT network passes overlap usefully?  The frozen teacher's [cond; uncond] pass and the student's two-timestep pass of
one distillation step share only their inputs (train_pcm_lora_sd15.py:1192-1252): issued on two HIP streams, blocks of one pass could fill the
CUs the other leaves idle (deep-level launches fill 40-60 % of the chip; every phased-tile block owns a CU's LDS, so co-residency is per CU,
not per SIMD).  Times, at the BASELINE configs[1] size: student pass alone, teacher pass alone, both back to back on one stream, both on two
streams.  Eager launches, events on the default stream around a fork / join.

    python tools/two_stream_probe.py [--reps 5] [--batch 16]
"""
import argparse
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
    args = ap.parse_args()
    from pcm_amd.model import LoraState, UNet, UNetWeights
    from pcm_amd.unet_spec import UNetConfig, random_state_dict
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    capi.lib()
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
    x2[B:] = x2[:B]
    t2 = torch.randint(0, 1000, (B,), generator=g, device=dev).repeat(2)
    c2 = torch.randn(2 * B, 77, 768, generator=g, device=dev)
    side = torch.cuda.Stream()
    main_s = torch.cuda.current_stream()

    def f_student():
        return student.forward(x2, t2, c2, save=True, save_half=True)

    def f_teacher():
        return teacher.forward(x2, t2, c2, dup_halves=True)

    def both_serial():
        a = f_student(); b = f_teacher()
        return a, b

    def both_two_streams():
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
            b = f_teacher()
        a = f_student()
        main_s.wait_stream(side)
        return a, b

    # the pairing that is legal ACROSS steps: the frozen teacher's pass of step k+1 depends on nothing the student's backward of step k produces
    d_eps = torch.randn(B, 4, 64, 64, generator=g, device=dev) * 1e-3
    _, tape2 = f_student()
    tape = student.tape_first_half(tape2)

    def f_bwd():
        student.backward(d_eps, tape)
        return None

    def bwd_teacher_serial():
        f_bwd()
        return f_teacher()

    def bwd_teacher_two_streams():
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
            b = f_teacher()
        f_bwd()
        main_s.wait_stream(side)
        return b

    # the two halves of the student's two-timestep pass are independent too: ONE 2B pass (what the step runs) against two B passes on two streams
    student_b = UNet(W, lora)                # (a second runner: per-pass state is per runner)

    def two_halves_two_streams():
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
            b = student_b.forward(x2[B:], t2[B:], c2[B:])                          # target half: no tape
        a = student.forward(x2[:B], t2[:B], c2[:B], save=True)                     # online half: tape for the backward
        main_s.wait_stream(side)
        return a, b

    def two_halves_one_stream():
        a = student.forward(x2[:B], t2[:B], c2[:B], save=True)
        b = student_b.forward(x2[B:], t2[B:], c2[B:])
        return a, b

    for name, fn in (("student_2B_one_pass", f_student), ("student_two_B_passes_one_stream", two_halves_one_stream), ("student_two_B_passes_two_streams", two_halves_two_streams),
                     ("student_2B_one_pass_again", f_student), ("student_two_B_passes_two_streams_again", two_halves_two_streams)):
        keep = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            keep = fn()
        e1.record()
        torch.cuda.synchronize()
        del keep
        print("%-40s %.3f ms (eager, %d reps)" % (name, e0.elapsed_time(e1) / args.reps, args.reps), flush=True)

    for name, fn in (("student_fwd_2t", f_student), ("student_fwd_2t_again", f_student), ("teacher_2b_shared_prefix", f_teacher), ("both_one_stream", both_serial),
                     ("both_two_streams", both_two_streams), ("both_one_stream_again", both_serial), ("both_two_streams_again", both_two_streams),
                     ("student_bwd", f_bwd), ("bwd_then_teacher_one_stream", bwd_teacher_serial), ("bwd_and_teacher_two_streams", bwd_teacher_two_streams),
                     ("bwd_then_teacher_one_stream_again", bwd_teacher_serial), ("bwd_and_teacher_two_streams_again", bwd_teacher_two_streams)):
        keep = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            keep = fn()
        e1.record()
        torch.cuda.synchronize()
        del keep
        print("%-28s %.3f ms (eager, %d reps)" % (name, e0.elapsed_time(e1) / args.reps, args.reps), flush=True)


if __name__ == "__main__":
    main()
