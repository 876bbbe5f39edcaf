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
PEAK_HBM_BYTES = 8.0e12                                          # HBM3E peak (6.29 TB/s measured copy), MI355X_MICROARCH.md


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
    while any(p.poll() is None for p in procs):      # poll ALL ranks: whichever dies first must not leave the others waiting in a collective
        for p in procs:
            if p.poll() not in (None, 0):
                rc = rc or p.returncode
        if rc:
            for q in procs:
                if q.poll() is None:
                    q.terminate()
            break
        time.sleep(0.2)
    for p in procs:
        rc = p.wait() or rc
    return rc


def bench_other_config(args, world, rank, local_rank, dev, sync):
    """BASELINE.json configs[2..4] on this launch's GPUs with the same timing contract as the default (configs[1]) run: synthetic inputs
    resident in HBM, W warm-up steps, K timed steps between barriers, one JSON line.  ``roofline`` here is the whole step against the
    bf16 MFMA peak with the algorithmic TFLOP of pcm_amd/flops.py (config walk, 2 FLOP/MAC) -- no per-kernel split, no cpu_baseline
    (BASELINE.json's CPU-runnable case is configs[0], reported by the default run)."""
    from pcm_amd import capi, flops
    capi.lib()
    cfgname = args.config
    g = torch.Generator(device=dev).manual_seed(453645634 + rank)
    rn = lambda *s_, **k: torch.randn(*s_, generator=g, device=dev, **k)   # noqa: E731
    note, pipe = None, False
    if cfgname == "c3":
        from pcm_amd.discriminator import ADAPTER_DIMS, Discriminator
        from pcm_amd.model import LoraState, UNetWeights
        from pcm_amd.trainer import AdvDistiller, StepConfig
        from pcm_amd.unet_spec import UNetConfig, random_state_dict
        B = args.batch or 8
        ucfg = UNetConfig.sd15()
        W = UNetWeights(ucfg, random_state_dict(ucfg, 0, dev), dev)
        lora = LoraState(ucfg, 64, 8.0, dev, seed=1)
        disc = Discriminator(ADAPTER_DIMS, num_h_per_head=4, device=dev, seed=2)
        D = AdvDistiller(W, lora, StepConfig(multiphase=2, loss_type="huber", learning_rate=5e-6, adam_weight_decay=1e-3, w_min=4.0, w_max=5.0),
                         disc, adv_weight=0.1, adv_lr=1e-5, world_size=world)
        m = flops.unet_macs(ucfg)
        t = flops.step_tflop(m)
        hf = 2e-12 * flops.heads_macs(ADAPTER_DIMS, (32, 16, 8, 8, 8, 16, 32, 64, 64))
        fw4 = 2 * t["student_fwd"] + 2 * t["teacher_fwd"]
        tf_d = fw4 + 2 * t["teacher_fwd"] + 2 * hf + 2 * (2 * hf)                 # [fake; real] feature pass + heads fwd + heads dgrad/wgrad
        tf_g = fw4 + t["teacher_fwd"] + hf + 2 * hf + (t["teacher_fwd"] + 2e-12 * m["attn_core"]) + t["backward"]
        tf_sample = 0.5 * (tf_d + tf_g)                                          # steps alternate D, G; every step consumes one batch
        # world > 1: segmented capture, cut at the head-gradient buckets / the LoRA exchange.  Over gloo (the one-device rehearsal) the
        # replayed D step waits ~137 s for the nine async bucket all-reduces (profiles/r03_d_segmented_replay_timing.txt; the same
        # segmented graphs with no-op host actions replay at full speed): graphs only with RCCL there unless PCM_ADV_GRAPH=1
        use_graph = not args.no_graph and (world == 1 or torch.distributed.get_backend() == "nccl" or os.environ.get("PCM_ADV_GRAPH") == "1")
        pipe = not args.no_prefetch and world == 1     # (world > 1: the D step is a segmented capture cut at its collectives -- one launch chain)
        if use_graph:
            try:
                D.capture_adv(B, pipeline=pipe)
            except RuntimeError as e:      # same launches issued eagerly (capture_adv leaves no capture open and no segment state behind)
                log("adversarial hipGraph capture failed (%s); falling back to eager launches" % str(e).splitlines()[0])
                use_graph = False

        def draw():
            return [rn(B, 4, 64, 64), rn(B, 77, 768), rn(B, 77, 768), rn(B, 4, 64, 64), torch.randint(0, 50, (B,), generator=g, device=dev),
                    4.0 + torch.rand(B, generator=g, device=dev), rn(B, 4, 64, 64), rn(B, 4, 64, 64), torch.rand(B, generator=g, device=dev)]
        state = {"gs": 0}

        def step(b, nxt=None):
            f = D.step_adv_graphed if use_graph else D.step_adv
            out = f(state["gs"], *b, prefetch=tuple(nxt[:6]) if (pipe and nxt is not None) else None)
            state["d" if state["gs"] % 2 == 0 else "g"] = out
            state["gs"] += 1
            return out
        workload = ("SD1.5 PCM-LoRA + latent discriminator (9 taps x 4 heads = 36 heads, 663.8M head params), 2 phases, 64x64x4 latents, per-GPU "
                    "batch %d, alternating D / G steps (each consumes one batch)" % B)
        metric = "distillation images/sec SD1.5 adversarial (36 heads) 512px"
        note = "algorithmic TFLOP/sample: D step %.2f, G step %.2f" % (tf_d, tf_g)
        if args.steps % 2:
            args.steps += 1      # whole D + G pairs
    elif cfgname == "c4":
        from pcm_amd.model import LoraState, UNetWeights
        from pcm_amd.trainer import Distiller, StepConfig
        from pcm_amd.unet_spec import UNetConfig, random_state_dict
        B = args.batch or 4
        ucfg = UNetConfig.sdxl()
        sd = random_state_dict(ucfg, 0, dev)
        W = UNetWeights(ucfg, sd, dev)
        del sd
        lora = LoraState(ucfg, 64, 8.0, dev, seed=1)
        D = Distiller(W, lora, StepConfig(multiphase=4, num_ddim_timesteps=40, w_min=6.0, w_max=7.0, learning_rate=2e-6, adam_weight_decay=0.0,
                                          loss_type="huber"), world_size=world)
        tf_sample = flops.step_tflop(flops.unet_macs(ucfg, 128, 128, 77, 64))["step"]
        tids = torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * B, device=dev)
        ac = dict(text_embeds=rn(B, 1280), time_ids=tids)
        uac = dict(text_embeds=torch.zeros(B, 1280, device=dev), time_ids=tids)
        un = torch.zeros(B, 77, 2048, device=dev)
        use_graph = not args.no_graph
        pipe = not args.no_prefetch          # the next batch's teacher pass beside this batch's student work (trainer.Distiller.step(prefetch=...))
        if use_graph:
            D.capture(B, H=128, W=128, ctx_len=77, ctx_dim=2048, added_cond=ac, uncond_added_cond=uac, pipeline=pipe)

        def draw():
            return [rn(B, 4, 128, 128), rn(B, 77, 2048), un, rn(B, 4, 128, 128), torch.randint(0, 40, (B,), generator=g, device=dev),
                    6.0 + torch.rand(B, generator=g, device=dev)]

        def step(b, nxt=None):
            return (D.step_graphed if use_graph else D.step)(*b, added_cond=ac, uncond_added_cond=uac,
                                                             prefetch=(tuple(nxt) + (ac, uac)) if (pipe and nxt is not None) else None)
        workload = "SDXL PCM-LoRA distillation step (2.57B UNet, text_time conditioning), 4 phases, 128x128x4 latents (1024 px), per-GPU batch %d, LoRA r=64" % B
        metric = "distillation images/sec SDXL 1024px"
    else:
        from pcm_amd.mmdit import MMDiTWeights, sd3_lora_state
        from pcm_amd.mmdit_spec import MMDiTConfig, random_state_dict
        from pcm_amd.trainer_sd3 import SD3Distiller, SD3StepConfig
        B = args.batch or 2
        mcfg = MMDiTConfig.sd3_medium()
        sd = random_state_dict(mcfg, 0, dev)
        W = MMDiTWeights(mcfg, sd, dev)
        del sd
        lora = sd3_lora_state(mcfg, 32, 8.0, dev, seed=1)
        D = SD3Distiller(W, lora, SD3StepConfig(multiphase=2, num_euler_timesteps=100, learning_rate=5e-6, adam_weight_decay=1e-3), world_size=world)
        tf_sample = flops.step_tflop(flops.mmdit_macs(mcfg))["step"]
        use_graph = not args.no_graph
        pipe = not args.no_prefetch
        if use_graph:
            D.capture(B, pipeline=pipe)

        def draw():
            return [rn(B, 16, 128, 128), rn(B, 154, 4096), rn(B, 2048), rn(B, 154, 4096), rn(B, 2048), rn(B, 16, 128, 128),
                    torch.randint(0, 100, (B,), generator=g, device=dev)]

        def step(b, nxt=None):
            return (D.step_graphed if use_graph else D.step)(*b, prefetch=tuple(nxt) if (pipe and nxt is not None) else None)
        workload = "SD3-medium (MMDiT 2.03B, 4096 image + 154 text tokens) PCM-LoRA distillation step, 2 phases, 128x128x16 latents, per-GPU batch %d, LoRA r=32" % B
        metric = "distillation images/sec SD3-medium 1024px"
    torch.cuda.synchronize()
    log("%s: model ready (%.1f GB allocated), %s" % (cfgname, torch.cuda.memory_allocated() / 2**30, "hipGraph replay" if use_graph else "eager"))
    batches = [draw() for _ in range(args.warmup + args.steps)]
    nb = len(batches)
    for i in range(args.warmup):
        step(batches[i], batches[(i + 1) % nb])
    sync()
    t0 = time.perf_counter()
    for i in range(args.warmup, nb):
        step(batches[i], batches[(i + 1) % nb])      # (c4: the next batch is announced, also in the last timed step -- one teacher pass per step)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    ms = dt * 1e3 / args.steps
    ach = tf_sample * B / (ms * 1e-3)
    if os.environ.get("PCM_GEMM_TABLE") and rank == 0 and world == 1:
        # diagnostic only (after the timed region): one more eager step (c3: a D + G pair) with HIP events around every pcm_gemm_bf16 launch
        from pcm_amd import ops
        ops.GEMM_PROFILE = []
        eager_steps = [draw() for _ in range(2 if cfgname == "c3" else 1)]
        ug, use_graph = use_graph, False
        for b in eager_steps:
            step(b)
        use_graph = ug
        torch.cuda.synchronize()
        prof, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
        agg = {}
        for fl, e0, e1, key, _plan, _nb in prof:
            a = agg.setdefault(str(key) + " plan %d" % _plan, [0, 0.0, 0.0])
            a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += fl
        with open(os.environ["PCM_GEMM_TABLE"], "w") as f:
            f.write("# %s: every pcm_gemm_bf16 launch of %d eager step(s), by shape (M, N, (K per segment), kind) and plan\n" % (cfgname, len(eager_steps)))
            for k, (n, t, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                f.write("%-56s calls %4d  total %8.3f ms  avg %7.1f us  %7.1f TF/s\n" % (k, n, t, 1e3 * t / n, fl / t / 1e9))
    losses = None
    if cfgname == "c3" and "d" in state and "g" in state:      # last D / G step of this rank (graph replay and eager launches must agree on them)
        losses = {"d_loss_last": round(float(state["d"]["d_loss"]), 6), "loss_cm_last": round(float(state["g"]["loss_cm"]), 6),
                  "g_loss_last": round(float(state["g"]["g_loss"]), 6)}
    if rank == 0:
        line = {"metric": metric, "value": round(world * B / (dt / args.steps), 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
                "data": "synthetic", "config": {"workload": workload, "baseline_config": cfgname, "global_batch": world * B, "parallelism": "dp%d" % world,
                                                "launch": "hipGraph replay" if use_graph else "eager", "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 1e9, 1), "losses": losses,
                                                "teacher_prefetch": "next batch's ODE-solver teacher pass on a second stream" if pipe else "off"},
                "roofline": {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                             "traffic": None, "what": "whole step, algorithmic TFLOP from the config walk of pcm_amd/flops.py",
                             "algorithmic_tflop_per_sample_step": round(tf_sample, 3), "note": note},
                "cpu_baseline": None}
        print(json.dumps(line), flush=True)


def pmc_row_of_committed_table(kernel_prefix, build_id):
    """the row of ``kernel_prefix`` in the newest committed per-kernel PMC table (profiles/r*_pmc_step_table.txt, written by
    tools/pmc_step_table.py from three separate rocprofv3 --pmc passes): columns are found BY NAME in the table's header line, and the table's
    `# source_id:` stamp is compared with the running library's build id -- a table measured on other kernel sources (or carrying no stamp:
    rounds <= 5) is returned with stale = True, never silently."""
    import glob
    import re
    tabs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_step_table.txt")),
                  key=lambda f: (int(re.match(r"r(\d+)", os.path.basename(f)).group(1)), os.path.basename(f)))
    if not tabs:
        return None
    sid, cols, row = None, None, None
    for ln in open(tabs[-1]):
        if ln.startswith("# source_id:"):
            sid = ln.split(":", 1)[1].strip()
        elif ln.startswith("kernel ") and cols is None:
            cols = ln.split()[1:]
        elif ln.startswith(kernel_prefix) and cols is not None and row is None:
            vals = ln.split()[-len(cols):]
            row = dict(zip(cols, vals))
    if row is None:
        return None
    need = ("calls", "dur_ms", "mfma_util", "rd_MB/l", "wr_MB/l", "GB/s")
    if any(k not in row for k in need):
        raise ValueError("PMC table %s lacks columns %s" % (tabs[-1], [k for k in need if k not in row]))
    return {"source": "profiles/" + os.path.basename(tabs[-1]), "source_id": sid, "stale": not (sid and build_id.startswith(sid + "-")),
            "launches": int(row["calls"]), "kernel_ms": float(row["dur_ms"]), "mfma_util": float(row["mfma_util"]),
            "read_MB_per_launch": float(row["rd_MB/l"]), "write_MB_per_launch": float(row["wr_MB/l"]), "fabric_GB_s": float(row["GB/s"]),
            "note": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128); FETCH_SIZE x 2 (gfx950) and WRITE_SIZE, separate passes"}


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0 or os.environ.get("PCM_BENCH_DEBUG"):
        print("[bench %7.1fs] %s" % (time.time() - T0, msg), file=sys.stderr, flush=True)


T0 = time.time()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the BASELINE config's -- c2 16, c3 8, c4 4, c5 2)")
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="BASELINE.json configs[1..4]; c2 (SD1.5 bs 16, the headline metric) is what the driver runs")
    ap.add_argument("--multiphase", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16"],
                    help="16-bit format of activations / weights / MFMA operands.  bf16 = BASELINE.json's configs (the default and the headline number); "
                         "fp16 = the IEEE-half build of the library (the reference's --mixed_precision=fp16 recipes), same MFMA rate, loss-scaled backward")
    ap.add_argument("--deterministic", action="store_true",
                    help="reproducible reductions (ops.set_deterministic: slabs / partials + ordered finalize instead of fp32 / fp64 atomics; "
                         "bitwise identical steps run to run) -- NOT the headline configuration, a cost measurement")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="(c2) the step as ONE launch chain.  Default: the frozen teacher's pass of batch k+1 is issued on a second HIP stream beside the "
                         "student's forward / backward on batch k (cross-step prefetch of the teacher targets, trainer.Distiller.step(prefetch=...)): "
                         "every timed step still runs exactly one teacher pass, one two-timestep student pass, one backward and one optimizer step")
    ap.add_argument("--teacher-fp16", action="store_true",
                    help="(c2, bf16) the ODE-solver teacher pass in IEEE half next to the bfloat16 student, as the reference's dtype-less "
                         "torch.autocast('cuda') does (train_pcm_lora_sd15.py:1218); a second packing of the frozen weights.  NOT the headline configuration")
    args = ap.parse_args()
    if args.precision == "fp16":
        from pcm_amd import precision
        precision.set_precision("fp16")
    if args.deterministic:
        from pcm_amd import ops as _ops
        _ops.set_deterministic(True)

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
    collective_ranks, devices = 1, [torch.cuda.get_device_name(dev)]
    if world > 1:
        # what the collective library itself saw: an all-reduce of ones counts the ranks, an all-gather collects each rank's device
        ones = torch.ones(1, device=dev)
        torch.distributed.all_reduce(ones)
        collective_ranks = int(ones.item())
        names = [None] * world
        torch.distributed.all_gather_object(names, "%s (cuda:%d, rank %d)" % (torch.cuda.get_device_name(dev), local_rank, rank))
        devices = names
        assert collective_ranks == world == torch.distributed.get_world_size(), (collective_ranks, world)

    if args.config != "c2":
        def sync0():
            if world > 1:
                torch.distributed.barrier(device_ids=[local_rank]) if torch.distributed.get_backend() == "nccl" else torch.distributed.barrier()
            torch.cuda.synchronize()
        bench_other_config(args, world, rank, local_rank, dev, sync0)
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    args.batch = args.batch or 16
    from pcm_amd import capi, ops
    from pcm_amd.model import LoraState, UNetWeights
    from pcm_amd.trainer import Distiller, StepConfig
    from pcm_amd.unet_spec import UNetConfig, random_state_dict
    capi.lib()  # fail loudly if the HIP library is missing (or was built from other sources than this tree's: capi.Lib)
    log("library loaded: build id %s, abi %d" % (capi.lib().build_id, capi.lib().dll.pcm_abi_version()))

    ucfg = UNetConfig.sd15()
    with torch.no_grad():
        sd = random_state_dict(ucfg, seed=0, device=dev)
        W = UNetWeights(ucfg, sd, dev)
        Wt = None
        if args.teacher_fp16 and args.precision == "bf16":
            from pcm_amd import precision
            with precision.format_scope("fp16"):
                Wt = UNetWeights(ucfg, sd, dev, need_bwd=False)
        del sd
        lora = LoraState(ucfg, 64, 8.0, dev, seed=1)   # peft init (B = 0), as the reference starts
    cfg = StepConfig(multiphase=args.multiphase, loss_type="huber", learning_rate=5e-6, adam_weight_decay=1e-3,
                     w_min=4.0, w_max=5.0)                 # train_pcm_lora_sd15.sh:5-29 hyper-parameters
    D = Distiller(W, lora, cfg, world_size=world, teacher_weights=Wt)
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
    pipeline = not args.no_prefetch
    if use_graph:
        try:
            D.capture(B, pipeline=pipeline)
            torch.cuda.synchronize()
            log("step captured into hipGraphs")
        except RuntimeError as e:      # same launches issued eagerly: slower on the host side, identical device work
            log("hipGraph capture failed (%s); falling back to eager launches" % str(e).splitlines()[0])
            use_graph = False
            D._graph = None

    def tup(b):
        return (b["latents"], b["prompt_embeds"], uncond, b["noise"], b["index"], b["w"])

    def run(b, eager=False, nxt=None):
        """one step on batch ``b``; ``nxt``: the batch of the next call (its teacher targets are computed beside this step's student work)"""
        f = D.step if (eager or not use_graph) else D.step_graphed
        return f(*tup(b), prefetch=tup(nxt) if (pipeline and nxt is not None) else None)

    def sync():
        if world > 1:
            if torch.distributed.get_backend() == "nccl":
                torch.distributed.barrier(device_ids=[local_rank])
            else:
                torch.distributed.barrier()
        torch.cuda.synchronize()

    log("rank %d: batches ready" % rank)
    nb = len(batches)
    for i, b in enumerate(batches[:args.warmup]):
        run(b, nxt=batches[(i + 1) % nb])
        torch.cuda.synchronize()
        log("rank %d: warmup step %d done" % (rank, i))
    sync()
    if world > 1 and use_graph:
        D.comm_events = []
    t0 = time.perf_counter()
    last = None
    for i in range(args.warmup, nb):
        # (the last timed step announces batch 0 again: every timed step carries one teacher pass, the pipeline is never drained inside the region)
        last = run(batches[i], nxt=batches[(i + 1) % nb])
    t_enq = time.perf_counter() - t0        # host time to ENQUEUE the steps (launch-bound if ~= dt)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt * 1e3 / args.steps
    # what the LAST TIMED step left behind, read before anything else runs: under hipGraph replay ``last`` is the graph's static output
    # dictionary and the comm-event list keeps growing, so the untimed probe step below would overwrite / extend both
    loss = float(last["loss"].item())
    timed_comm_events = None
    if world > 1:
        timed_comm_events, D.comm_events = D.comm_events, None
    # host cost of ONE step's launches with an empty queue (outside the timed region).  The "host enqueue" figure of the timed loop is
    # mostly back-pressure: with several steps queued hipGraphLaunch blocks until the GPU frees queue space, so it tracks the GPU time.
    D.bucket_log = []          # the collectives of ONE step, in issue order (first RCCL run diagnosable from the JSON line alone)
    t1 = time.perf_counter()
    run(batches[0])            # (the batch the last timed step announced: its teacher targets are in place, no eager prologue pass)
    host_idle_ms = (time.perf_counter() - t1) * 1e3
    bucket_log = list(D.bucket_log)
    sync()
    log("timed %d steps: %.1f ms/step (host enqueue %.1f ms/step with %d steps queued; %.2f ms for one step's launches on an idle queue)"
        % (args.steps, ms, t_enq * 1e3 / args.steps, args.steps, host_idle_ms))
    value = world * B / (dt / args.steps)
    comm = None
    if world > 1:
        ev = timed_comm_events
        comm = {"backend": torch.distributed.get_backend(), "collective_ranks": collective_ranks, "devices": devices,
                "buckets": 2 if (D.bucketed and lora.late_offset is not None) else 1, "grad_bytes": int(lora.grads.numel() * 4),
                "exposed_allreduce_ms_per_step": round(sum(a.elapsed_time(b) for a, b in ev) / max(1, len(ev)), 3) if ev else None,
                "collectives_of_one_step": [{"bucket": n_, "bytes": b_, "dtype": d_} for n_, b_, d_ in bucket_log],
                "note": "exposed = stream time between the end of the backward graph and the optimizer graph on rank 0 (early bucket + wait for the "
                        "late bucket that was launched between the two backward graphs)"}
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
        # (plan code 5xxx of pcm_gemm_plan_code); the family aggregate is reported next to it
        dom = [(p[0], t, p[5]) for p, t in zip(prof, times) if p[4] // 1000 == 5]
        d_fl, d_ms, d_nb = sum(x[0] for x in dom), sum(x[1] for x in dom), sum(x[2] for x in dom)
        ach = d_fl / (d_ms * 1e-3) / 1e12 if dom else fam
        # The kernel's launches fall into two classes with different roofs.  A launch is "hbm" when moving its ALGORITHMIC bytes at the
        # 8 TB/s peak takes longer than its flops at the 2.5 PFLOP/s peak (arithmetic intensity below 312 flop/byte: the 1x1 / linear
        # projections with K <= 1344), "mfma" otherwise (the 3x3 convolutions and long-K projections).  Each class against its own roof:
        cls = {"mfma": [0, 0.0, 0.0, 0.0], "hbm": [0, 0.0, 0.0, 0.0]}
        for fl, t_ms, nb in dom:
            c = cls["hbm" if nb / PEAK_HBM_BYTES > fl / (PEAK_BF16_TFLOPS * 1e12) else "mfma"]
            c[0] += 1; c[1] += t_ms; c[2] += fl; c[3] += nb
        classes = {}
        for name, (n_, t_ms, fl, nb) in cls.items():
            if not n_:
                continue
            tf_s, gb_s = fl / (t_ms * 1e-3) / 1e12, nb / (t_ms * 1e-3) / 1e9
            classes[name + "_bound_launches"] = {
                "launches": n_, "kernel_ms_per_step": round(t_ms, 2), "algorithmic_tflop": round(fl / 1e12, 2), "algorithmic_gb": round(nb / 1e9, 2),
                "achieved_tflops": round(tf_s, 1), "achieved_gb_s": round(gb_s, 1),
                "frac_of_own_roof": round(tf_s / PEAK_BF16_TFLOPS if name == "mfma" else gb_s * 1e9 / PEAK_HBM_BYTES, 4)}
        # the two-workgroups-per-CU kernel (plan code 1xxxx: the K <= 384 fused-GEGLU feed-forward projections), against both roofs
        w4l = [(p[0], t, p[5]) for p, t in zip(prof, times) if 10000 <= p[4] < 20000]
        # the weights-stationary kernel (plan code 30000 + K/32, round 6: the N = 320 / 960, K = 320 (+ 64) projections of the 64x64 level), against both roofs
        wsl = [(p[0], t, p[5]) for p, t in zip(prof, times) if p[4] >= 30000]
        ws = None
        if wsl:
            s_fl, s_ms, s_nb = sum(x[0] for x in wsl), sum(x[1] for x in wsl), sum(x[2] for x in wsl)
            ws = {"kernel": "pcm_gemm_ws_kernel (320-column weight slice in registers, 64-row activation tiles by LDS-DMA)", "launches": len(wsl),
                  "kernel_ms_per_step": round(s_ms, 2), "achieved_tflops": round(s_fl / (s_ms * 1e-3) / 1e12, 1),
                  "achieved_gb_s": round(s_nb / (s_ms * 1e-3) / 1e9, 1), "frac_hbm": round(s_nb / (s_ms * 1e-3) / PEAK_HBM_BYTES, 4)}
        w4 = None
        if w4l:
            w_fl, w_ms, w_nb = sum(x[0] for x in w4l), sum(x[1] for x in w4l), sum(x[2] for x in w4l)
            w4 = {"kernel": "pcm_gemm4w_kernel<5> (128x320 tile, two workgroups per CU)", "launches": len(w4l), "kernel_ms_per_step": round(w_ms, 2),
                  "achieved_tflops": round(w_fl / (w_ms * 1e-3) / 1e12, 1), "frac_mfma": round(w_fl / (w_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                  "achieved_gb_s": round(w_nb / (w_ms * 1e-3) / 1e9, 1), "frac_hbm": round(w_nb / (w_ms * 1e-3) / PEAK_HBM_BYTES, 4)}
        log("roofline leg done")
        if os.environ.get("PCM_GEMM_TABLE"):
            agg = {}
            for fl, e0, e1, key, _plan, _nb in prof:
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
        traffic, traffic_note, traffic_source, pmc = None, None, None, None
        try:
            # the per-kernel PMC table of one eager step on the final tree (tools/jobs/r05_e_pmc.sh -> tools/pmc_step_table.py: three separate
            # rocprofv3 --pmc passes): MFMA utilisation and fabric-side bytes of THIS kernel averaged over all its launches of a step -- the same
            # population `achieved` is taken over.  Committed measurement, named here; a PMC pass cannot run inside the timed process.
            pmc = pmc_row_of_committed_table("void pcm_gemm8p_kernel<3, false, false>", capi.lib().build_id)
        except Exception as e:
            log("PMC table not usable: %r" % (e,))
            pmc = None
        try:
            # NOT measured in this run (PMC passes need their own rocprofv3 runs): the newest committed measurement is copied in and named
            import glob
            import re
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_gemm8p_traffic.json")),
                           key=lambda f: (int(re.match(r"r(\d+)", os.path.basename(f)).group(1)), os.path.basename(f)))
            pj = json.load(open(cands[-1]))
            kern = pj["kernel"] if "kernel" in pj else pj["tap_outer_K_order (shipped)"]
            if pmc is not None:      # per average launch of the step (same population as `achieved`), from the PMC table above
                raise LookupError
            traffic = round(kern["traffic_MB_corrected"] * 1e6)
            traffic_source = "profiles/" + os.path.basename(cands[-1]) + " (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x2 per the gfx950 note; not re-measured by this run)"
            traffic_note = "bytes per launch of the M=131072 320->320 conv3x3 (+LoRA) launch of this kernel; algorithmic %.0f MB" % (
                pj["algorithmic_MB"]["read"] + pj["algorithmic_MB"]["write"])
        except Exception:
            pass
        if pmc is not None:
            traffic = round((pmc["read_MB_per_launch"] + pmc["write_MB_per_launch"]) * 1e6)
            traffic_source = pmc["source"] + (" (committed rocprofv3 --pmc passes over one eager step of THESE kernel sources, source id %s; not re-measured by this run)" % pmc["source_id"]
                                              if not pmc["stale"] else
                                              " -- STALE: measured on kernel sources %s, this run's library is %s; re-run tools/jobs/r06_pmc.sh" % (pmc["source_id"] or "without a stamp", capi.lib().build_id))
            traffic_note = "fabric-side bytes per AVERAGE launch of this kernel over the %d launches of a step (Infinity-Cache hits included); algorithmic %.0f MB per average launch" % (
                len(dom), d_nb / max(1, len(dom)) / 1e6)
        roofline = {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_note": traffic_note, "traffic_source": traffic_source,
                    "pmc": pmc,
                    "kernel": "pcm_gemm8p_kernel<3,false,false> (256x320 phased tile; all its launches of one step)",
                    "launches": len(dom), "avg_launch_us": round(1e3 * d_ms / max(1, len(dom)), 1),
                    "algorithmic_tflop": round(d_fl / 1e12, 2), "kernel_ms_per_step": round(d_ms, 2), "classes": classes,
                    "short_k_kernel": w4, "weights_stationary_kernel": ws,
                    "gemm_family": {"kernels": "pcm_gemm8p<3>/<2>, pcm_gemm4w<5>, pcm_gemm_kernel tiles, pcm_gemm_n64 (every pcm_gemm_bf16 launch)",
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
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
                "config": {"workload": "SD1.5 PCM-LoRA distillation step, %d phases, 64x64x4 latents, per-GPU batch %d, "
                                       "LoRA r=64 (67.25M trainable), huber, AdamW, random-init UNet (859.5M)" % (args.multiphase, B),
                           "global_batch": world * B, "parallelism": "dp%d" % world, "comm": comm, "launch": "hipGraph replay" if use_graph else "eager",
                           "teacher_prefetch": ("next batch's teacher pass on a second stream beside this batch's student forward / backward (same work per step, "
                                                "same numbers; --no-prefetch: one launch chain)") if pipeline else "off", "reductions": "reproducible" if args.deterministic else "atomics", "teacher_pass": "fp16" if (Wt is not None or args.precision == "fp16") else "bf16", "loss_last": round(loss, 6),
                           "host_ms_per_step_idle_queue": round(host_idle_ms, 2)},
                "roofline": roofline, "cpu_baseline": cpu, "build_id": capi.lib().build_id}
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
