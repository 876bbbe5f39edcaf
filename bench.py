#!/usr/bin/env python
"""bench.py — distillation images/sec of the SD1.5 PCM-LoRA step on MI355X (BASELINE.json metric).

A "step" = one full phased-consistency distillation step of train_pcm_lora_sd15.py:1117-1301 on a
synthetic batch already resident in HBM: student forward (grad) + batched teacher cond/uncond
forward + target forward + PCM solver math + loss + LoRA-only backward + grad all-reduce + clip +
AdamW + operand repack.  Workload = BASELINE.json configs[1]: SD1.5 UNet (random init, no weights
offline), 4 phases, 512 px (64x64x4 latents), per-GPU batch 16, bf16 MFMA compute / fp32 accumulate.

Launch:  python bench.py --gpus 1            (single process)
         python bench.py --gpus N            (spawns its own N ranks, one per GPU, rendezvous on 127.0.0.1)
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N
A WORLD_SIZE that disagrees with --gpus is an error, never a silent 1-rank run.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "phased-consistency-model_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

# algorithmic work (SURVEY §8d / BASELINE.md §2), TFLOP per sample
TF_STUDENT_FWD, TF_TEACHER_FWD, TF_BWD = 0.8976, 0.8033, 1.12
TF_STEP = 2 * TF_STUDENT_FWD + 2 * TF_TEACHER_FWD + TF_BWD     # 4.52
PEAK_BF16_TFLOPS = 2500.0                                        # dense MFMA bf16, MI355X_MICROARCH.md


def cpu_baseline(seed, max_seconds=400.0):
    """The oracle (CPU fp32 restatement of the reference step; kind "port": diffusers/peft are not installable here, see
    BASELINE.md section 3) timed on this host in BASELINE.json configs[0] exactly as SURVEY section 8(d) prescribes: SD1.5 UNet,
    bs 2, 2 phases, CFG solver on, fp32, torch AdamW, 1 warm-up step + up to 3 timed steps.  Bounded: timed steps stop early once
    ``max_seconds`` of CPU time have been spent (at least one timed step is always taken; the sample says how many)."""
    from oracle import pcm_step as OS
    from oracle import unet_sd15 as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count()
    # torch's default intra-op thread count honours the container's CPU quota; forcing one thread per visible core oversubscribes a
    # quota-limited box (measured on the GPU box: > 5 min per step instead of ~50 s).  ``cores`` reports the threads actually used.
    oc = O.UNetConfig.sd15()
    sd = O.init_state_dict(oc, 0)
    lora = O.init_lora(oc, 64, seed=1)
    cfg = OS.StepConfig(multiphase=2, loss_type="huber", lr=5e-6, adam_weight_decay=1e-3, w_min=4.0, w_max=5.0)
    bs, state, times, loss = 2, {}, [], float("nan")
    t_all = time.time()
    for i in range(4):
        inp = OS.draw_inputs(bs, cfg, seed=seed + i)
        t0 = time.time()
        out = OS.distill_step(oc, sd, lora, inp, cfg, state, i + 1)
        dt = time.time() - t0
        loss = float(out["loss"])
        if i > 0:
            times.append(dt)
        else:
            warm = dt
        if i >= 1 and time.time() - t_all > max_seconds:
            break
    s_step = sum(times) / len(times)
    return {"value": bs / s_step, "unit": "images/sec", "cores": torch.get_num_threads(), "visible_cores": cores, "kind": "port",
            "s_per_step": round(s_step, 2),
            "sample": "BASELINE.json configs[0]: SD1.5 UNet (random init), bs %d, 2 phases, CFG solver on, fp32, huber, torch AdamW; "
                      "1 warm-up step (%.1f s) + %d timed step(s) of the 3 prescribed (time cap %.0f s), oracle/pcm_step.py, last loss %.5f"
                      % (bs, warm, len(times), max_seconds, loss)}


def self_spawn(n):
    """Launcher for ``python bench.py --gpus N`` without torchrun: N children, one rank per GPU, same argv."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
        if rc:                      # one rank died: do not leave the others waiting in a collective
            for q in procs:
                if q.poll() is None:
                    q.terminate()
    return rc


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0 or os.environ.get("PCM_BENCH_DEBUG"):
        print("[bench %7.1fs] %s" % (time.time() - T0, msg), file=sys.stderr, flush=True)


T0 = time.time()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch (BASELINE config: 16)")
    ap.add_argument("--multiphase", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain ``python bench.py --gpus N`` (no torchrun): become the launcher -- one child process per GPU with the torchrun
        # environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), rendezvous on 127.0.0.1; rank 0's JSON line is the only stdout
        sys.exit(self_spawn(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with `python bench.py --gpus N` (self-spawning) or "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "PCM_FORCE_DEVICE" in os.environ:          # single-GPU rehearsal of the N>1 path (with PCM_DIST_BACKEND=gloo)
        local_rank = int(os.environ["PCM_FORCE_DEVICE"])
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        torch.distributed.init_process_group(os.environ.get("PCM_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    rccl_ranks, devices = 1, [torch.cuda.get_device_name(dev)]
    if world > 1:
        # what the collective library itself saw: an all-reduce of ones counts the ranks, an all-gather collects each rank's device
        ones = torch.ones(1, device=dev)
        torch.distributed.all_reduce(ones)
        rccl_ranks = int(ones.item())
        names = [None] * world
        torch.distributed.all_gather_object(names, "%s (cuda:%d, rank %d)" % (torch.cuda.get_device_name(dev), local_rank, rank))
        devices = names
        assert rccl_ranks == world == torch.distributed.get_world_size(), (rccl_ranks, world)

    from pcm_amd import capi, ops
    from pcm_amd.model import LoraState, UNetWeights
    from pcm_amd.trainer import Distiller, StepConfig
    from pcm_amd.unet_spec import UNetConfig, random_state_dict
    capi.lib()  # fail loudly if the HIP library is missing
    log("library loaded")

    ucfg = UNetConfig.sd15()
    with torch.no_grad():
        sd = random_state_dict(ucfg, seed=0, device=dev)
        W = UNetWeights(ucfg, sd, dev)
        del sd
        lora = LoraState(ucfg, 64, 8.0, dev, seed=1)   # peft init (B = 0), as the reference starts
    cfg = StepConfig(multiphase=args.multiphase, loss_type="huber", learning_rate=5e-6, adam_weight_decay=1e-3,
                     w_min=4.0, w_max=5.0)                 # train_pcm_lora_sd15.sh:5-29 hyper-parameters
    D = Distiller(W, lora, cfg, world_size=world)
    torch.cuda.synchronize()
    log("weights packed, LoRA state ready (%.1f GB allocated)" % (torch.cuda.memory_allocated() / 2**30))
    B = args.batch
    seed = 453645634 + rank                                # train_pcm_lora_sd15.sh:26 + per-rank offset (:797)
    g = torch.Generator(device=dev).manual_seed(seed)
    uncond = torch.randn(B, 77, 768, generator=g, device=dev)

    def draw():
        return dict(latents=torch.randn(B, 4, 64, 64, generator=g, device=dev),
                    prompt_embeds=torch.randn(B, 77, 768, generator=g, device=dev),
                    noise=torch.randn(B, 4, 64, 64, generator=g, device=dev),
                    index=torch.randint(0, cfg.num_ddim_timesteps, (B,), generator=g, device=dev),
                    w=(cfg.w_max - cfg.w_min) * torch.rand(B, generator=g, device=dev) + cfg.w_min)

    batches = [draw() for _ in range(args.warmup + args.steps)]   # resident in HBM before timing

    use_graph = not args.no_graph
    if use_graph:
        try:
            D.capture(B)
            torch.cuda.synchronize()
            log("step captured into hipGraphs")
        except RuntimeError as e:      # same launches issued eagerly: slower on the host side, identical device work
            log("hipGraph capture failed (%s); falling back to eager launches" % str(e).splitlines()[0])
            use_graph = False
            D._graph = None

    def run(b, eager=False):
        f = D.step if (eager or not use_graph) else D.step_graphed
        return f(b["latents"], b["prompt_embeds"], uncond, b["noise"], b["index"], b["w"])

    def sync():
        if world > 1:
            if torch.distributed.get_backend() == "nccl":
                torch.distributed.barrier(device_ids=[local_rank])
            else:
                torch.distributed.barrier()
        torch.cuda.synchronize()

    log("rank %d: batches ready" % rank)
    for i, b in enumerate(batches[:args.warmup]):
        run(b)
        torch.cuda.synchronize()
        log("rank %d: warmup step %d done" % (rank, i))
    sync()
    if world > 1 and use_graph:
        D.comm_events = []
    t0 = time.perf_counter()
    last = None
    for b in batches[args.warmup:]:
        last = run(b)
    t_enq = time.perf_counter() - t0        # host time to ENQUEUE the steps (launch-bound if ~= dt)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt * 1e3 / args.steps
    log("timed %d steps: %.1f ms/step (host enqueue %.1f ms/step)" % (args.steps, ms, t_enq * 1e3 / args.steps))
    value = world * B / (dt / args.steps)
    comm = None
    if world > 1:
        ev, D.comm_events = D.comm_events, None
        comm = {"backend": torch.distributed.get_backend(), "rccl_ranks": rccl_ranks, "devices": devices,
                "buckets": 2 if (D.bucketed and lora.late_offset is not None) else 1, "grad_bytes": int(lora.grads.numel() * 4),
                "exposed_allreduce_ms_per_step": round(sum(a.elapsed_time(b) for a, b in ev) / max(1, len(ev)), 3) if ev else None,
                "note": "exposed = stream time between the end of the backward graph and the optimizer graph on rank 0 (early bucket + wait for the "
                        "late bucket that was launched between the two backward graphs)"}
    loss = float(last["loss"].item())

    # north_star quantity: MFMA fraction of the TWO-TIMESTEP STUDENT FORWARD (online at t_{n+k} + target at t_n: rows a6 + a12 of
    # SURVEY section 8, 2 x B x 0.8976 TFLOP) -- the 2B-sample LoRA pass of the step, event-timed on the launch stream, eager
    fwd2t = None
    if not args.no_roofline:
        b = batches[-1]
        with torch.no_grad():
            x2 = torch.cat([b["latents"], b["noise"]]); t2 = torch.cat([D.tables.ddim_timesteps[b["index"]]] * 2)
            c2 = torch.cat([b["prompt_embeds"], b["prompt_embeds"]])
        reps = 3
        D.student.forward(x2, t2, c2, save=True, save_half=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            D.student.forward(x2, t2, c2, save=True, save_half=True)
        e1.record()
        torch.cuda.synchronize()
        f_ms = e0.elapsed_time(e1) / reps
        f_tf = 2 * B * TF_STUDENT_FWD
        fwd2t = {"what": "online + target student forward as one 2B-sample LoRA pass (activations saved for the backward), eager launches",
                 "ms": round(f_ms, 2), "algorithmic_tflop": round(f_tf, 2), "achieved": round(f_tf / (f_ms * 1e-3), 1),
                 "frac": round(f_tf / (f_ms * 1e-3) / PEAK_BF16_TFLOPS, 4), "target_frac": 0.5}
        log("two-timestep student forward: %.2f ms = %.0f TFLOP/s" % (f_ms, f_tf / (f_ms * 1e-3)))

    roofline = None
    if not args.no_roofline:
        # dominant kernel family = pcm_gemm_bf16 (conv3x3 implicit GEMM / Linear / LoRA): one extra,
        # instrumented step with HIP events around every launch on the launch stream.  Every rank runs
        # it (the step contains the gradient all-reduce); rank 0 reports.
        ops.GEMM_PROFILE = [] if rank == 0 else None
        run(batches[-1], eager=True)
        torch.cuda.synchronize()
    if rank == 0 and not args.no_roofline:
        prof, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
        times = [p[1].elapsed_time(p[2]) for p in prof]
        flops = sum(p[0] for p in prof)
        tms = sum(times)
        fam = flops / (tms * 1e-3) / 1e12
        # the dominant kernel of the step (rocprofv3: ~41 % of the kernel time) is the 256x320 phased tile pcm_gemm8p_kernel<3,false,false>
        # (plan code 5xxx of pcm_debug_last_gemm_plan); the family aggregate is reported next to it
        dom = [(p[0], t) for p, t in zip(prof, times) if p[4] // 1000 == 5]
        d_fl, d_ms = sum(x[0] for x in dom), sum(x[1] for x in dom)
        ach = d_fl / (d_ms * 1e-3) / 1e12 if dom else fam
        log("roofline leg done")
        if os.environ.get("PCM_GEMM_TABLE"):
            agg = {}
            for fl, e0, e1, key, _plan in prof:
                a = agg.setdefault(str(key) + " plan %d" % _plan, [0, 0.0, 0.0])
                a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += fl
            rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
            with open(os.environ["PCM_GEMM_TABLE"], "w") as f:
                for k, (n, t, fl) in rows:
                    f.write("%-56s calls %4d  total %8.3f ms  avg %7.1f us  %7.1f TF/s\n" % (k, n, t, 1e3 * t / n, fl / t / 1e9))
        # HBM-side bytes of the dominant kernel from the committed PMC passes (separate rocprofv3 --pmc runs, gfx950 FETCH_SIZE x2
        # correction calibrated on a known copy): profiles/r02_pmc_gemm8p_traffic.json (round 1: r01_e_...).  It is for ONE launch of the largest
        # 64x64-resolution conv (M=131072, 320->320 + LoRA; algorithmic 186 MB): the 9 taps re-read the activation tile through
        # the fabric (served by the 256 MB Infinity Cache, not by HBM); see DESIGN.md section 6.
        traffic, traffic_note = None, None
        try:
            p2 = os.path.join(ROOT, "profiles", "r02_pmc_gemm8p_traffic.json")
            if os.path.exists(p2):
                pj = json.load(open(p2))
                traffic = round(pj["kernel"]["traffic_MB_corrected"] * 1e6)
            else:
                pj = json.load(open(os.path.join(ROOT, "profiles", "r01_e_pmc_gemm8p_traffic.json")))
                traffic = round(pj["tap_outer_K_order (shipped)"]["traffic_MB_corrected"] * 1e6)
            traffic_note = "bytes per launch of the M=131072 320->320 conv3x3 (+LoRA) launch of this kernel; algorithmic %.0f MB" % (
                pj["algorithmic_MB"]["read"] + pj["algorithmic_MB"]["write"])
        except Exception:
            pass
        roofline = {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_note": traffic_note,
                    "kernel": "pcm_gemm8p_kernel<3,false,false> (256x320 phased tile; all its launches of one step)",
                    "launches": len(dom), "avg_launch_us": round(1e3 * d_ms / max(1, len(dom)), 1),
                    "algorithmic_tflop": round(d_fl / 1e12, 2), "kernel_ms_per_step": round(d_ms, 2),
                    "gemm_family": {"kernels": "pcm_gemm8p<3>/<2>, pcm_gemm_kernel tiles, pcm_gemm_n64 (every pcm_gemm_bf16 launch)",
                                    "launches": len(prof), "algorithmic_tflop_per_step": round(flops / 1e12, 2),
                                    "kernel_ms_per_step": round(tms, 2), "achieved": round(fam, 1), "frac": round(fam / PEAK_BF16_TFLOPS, 4)},
                    "step_tflops_algorithmic": round(TF_STEP * B / (ms * 1e-3), 1), "step_frac": round(TF_STEP * B / (ms * 1e-3) / PEAK_BF16_TFLOPS, 4),
                    "student_fwd_2t": fwd2t}
    if world > 1:
        torch.distributed.barrier()
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(453645634)
    if rank == 0:
        line = {"metric": "distillation images/sec (two UNet fwd + bwd) SD1.5 512px bs=16", "value": round(value, 3),
                "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "config": {"workload": "SD1.5 PCM-LoRA distillation step, %d phases, 64x64x4 latents, per-GPU batch %d, "
                                       "LoRA r=64 (67.25M trainable), huber, AdamW, random-init UNet (859.5M)" % (args.multiphase, B),
                           "global_batch": world * B, "parallelism": "dp%d" % world, "comm": comm, "launch": "hipGraph replay" if use_graph else "eager", "loss_last": round(loss, 6)},
                "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
