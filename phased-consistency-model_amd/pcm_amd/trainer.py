"""The phased-consistency distillation step (reference: train_pcm_lora_sd15.py:1139-1301) as host
code over the HIP kernels: DDIM tables built once on device (the reference rebuilds the phase-edge
table with numpy three times per step), student / teacher(cond+uncond batched) / target forwards,
fused PCM math, LoRA-only backward, one flat all-reduce (RCCL over xGMI), fused clip+AdamW.

Behavioural notes kept from the reference: the "target network" is the ONLINE
LoRA weights under no-grad (update_ema is defined but never called; ``ema_rate`` here defaults to
None = reference behaviour); ``w`` only scales the teacher CFG step; index 0 is a boundary sample.
"""
import contextlib
import gc
import math
import os

import numpy as np
import torch

from . import capi, ops, precision
from .model import LoraState, UNet, UNetWeights


class StepConfig:
    """Hyper-parameters with the reference's argparse names and defaults (train_pcm_lora_sd15.py:381-735)."""

    def __init__(self, num_ddim_timesteps=50, multiphase=8, w_min=5.0, w_max=15.0, loss_type="l2", huber_c=0.001,
                 learning_rate=1e-4, adam_beta1=0.9, adam_beta2=0.999, adam_weight_decay=1e-2, adam_epsilon=1e-8,
                 max_grad_norm=1.0, lora_rank=64, lora_alpha=8.0, not_apply_cfg_solver=False, num_train_timesteps=1000,
                 beta_start=0.00085, beta_end=0.012, ema_rate=None):
        self.__dict__.update({k: v for k, v in locals().items() if k != "self"})


def scaled_linear_alphas_cumprod(n=1000, beta_start=0.00085, beta_end=0.012):
    """DDPMScheduler 'scaled_linear' table (scheduling_ddpm_modified.py:205-207,:220-221), fp32."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class DDIMTables:
    """DDIMSolver.__init__ (train_pcm_lora_sd15.py:289-311) + the phase edges (:1157-1163, :322-328),
    resident on the device.  ``acp_prev`` is float64 exactly like the reference's
    ddim_alpha_cumprods_prev (np.asarray over python floats)."""

    def __init__(self, cfg: StepConfig, device):
        acp = scaled_linear_alphas_cumprod(cfg.num_train_timesteps, cfg.beta_start, cfg.beta_end)
        a = acp.numpy()
        step_ratio = cfg.num_train_timesteps // cfg.num_ddim_timesteps
        t = (np.arange(1, cfg.num_ddim_timesteps + 1) * step_ratio).round().astype(np.int64) - 1
        t_prev = np.asarray([0] + t[:-1].tolist())
        acp_prev = np.asarray([a[0]] + a[t[:-1]].tolist())          # float64, as in the reference
        edges = np.floor(np.linspace(0, cfg.num_ddim_timesteps, num=cfg.multiphase, endpoint=False)).astype(np.int64)
        self.topk = step_ratio
        self.acp = acp.to(device)
        self.ddim_timesteps = torch.from_numpy(t).long().to(device)
        self.ddim_timesteps_prev = torch.from_numpy(t_prev).long().to(device)
        self.acp_prev = torch.from_numpy(acp_prev).to(device)
        assert self.acp_prev.dtype == torch.float64
        self.edges = torch.from_numpy(edges).long().to(device)


SEG_TIMING = os.environ.get("PCM_SEG_TIMING") == "1"
# the teacher's cond / uncond halves share sample and timestep: compute the common prefix once (0 restores the plain 2B pass, for A/B)
DEDUP_TEACHER_PREFIX = os.environ.get("PCM_DEDUP_TEACHER", "1") != "0"
SEG_FORCE = os.environ.get("PCM_SEG_FORCE") == "1"      # debugging aid: segmented capture of the adversarial step at world_size 1 too


class SegmentedGraph:
    """A launch sequence captured as SEVERAL hipGraphs cut at the points where a collective has to be issued from the host (RCCL calls are
    not captured): replay = graph, host action, graph, host action, ..., graph.  All segments share one memory pool."""

    def __init__(self, pool=None):
        self.items, self.cur, self.pool = [], None, pool

    def begin(self):
        self.cur = torch.cuda.CUDAGraph()
        if self.pool is None:
            self.cur.capture_begin(capture_error_mode="thread_local")
        else:
            self.cur.capture_begin(pool=self.pool, capture_error_mode="thread_local")

    def cut(self, host_fn):
        """called from inside the code being captured, where the eager path would issue a collective"""
        self.cur.capture_end()
        if self.pool is None:
            self.pool = self.cur.pool()
        self.items += [self.cur, host_fn]
        self.begin()

    def end(self):
        self.cur.capture_end()
        if self.pool is None:
            self.pool = self.cur.pool()
        self.items.append(self.cur)
        self.cur = None

    def abort(self):
        """the code being captured raised: close the open capture so that the stream is usable again (the segments are discarded)"""
        if self.cur is not None:
            try:
                self.cur.capture_end()
            except RuntimeError:
                pass
            self.cur = None
        self.items = []

    def replay(self):
        if SEG_TIMING:
            return self._replay_timed()
        for it in self.items:
            if isinstance(it, torch.cuda.CUDAGraph):
                it.replay()
            else:
                it()

    def _replay_timed(self):
        """PCM_SEG_TIMING=1: host time of every item (graph launch / host action) and the final drain, printed per replay (debugging aid)"""
        import sys
        import time
        t, parts = time.perf_counter(), []
        for it in self.items:
            if isinstance(it, torch.cuda.CUDAGraph):
                it.replay()
                kind = "g"
            else:
                it()
                kind = "h"
            t2 = time.perf_counter()
            parts.append("%s%.1f" % (kind, 1e3 * (t2 - t)))
            t = t2
        torch.cuda.synchronize()
        print("[seg replay, ms] " + " ".join(parts) + " | drain %.1f" % (1e3 * (time.perf_counter() - t)), file=sys.stderr, flush=True)


class Distiller:
    """Owns the frozen UNet weights, the LoRA state and the optimizer state of one rank."""
    _seg = None            # SegmentedGraph being captured (AdvDistiller.capture_adv at world_size > 1); None: collectives are issued directly
    comm_events = None     # a list: step_graphed appends (start, end) events around the gradient exchange
    loss_scale_dev = loss_good_dev = None      # device-side GradScaler state, set by __init__ under precision "fp16"

    def __init__(self, weights: UNetWeights, lora: LoraState, cfg: StepConfig, world_size=1, process_group=None, teacher_weights=None):
        """``teacher_weights``: a second packing of the SAME frozen state dict in the other 16-bit format (built under
        ``precision.format_scope``), used for the ODE-solver teacher pass only.  The reference runs that pass under
        ``torch.autocast("cuda")`` with no dtype (train_pcm_lora_sd15.py:1217-1218), i.e. in IEEE half even when the student trains under
        --mixed_precision=bf16; ``teacher_weights`` packed in "fp16" next to a bf16 student reproduces exactly that split.  None (default):
        every pass runs in the process's one format."""
        self.W, self.lora, self.cfg = weights, lora, cfg
        self.device = lora.device
        self.tables = DDIMTables(cfg, self.device)
        self.student = UNet(weights, lora)
        self.teacher = UNet(weights, None)
        self.teacher_ode, self._ode_scope = self.teacher, contextlib.nullcontext
        if teacher_weights is not None and teacher_weights.format != precision.precision():
            fmt = teacher_weights.format
            self.teacher_ode, self._ode_scope = UNet(teacher_weights, None), (lambda: precision.format_scope(fmt))
        self.world_size, self.pg = world_size, process_group
        self.step_count = 0
        self.fuse_online_target = True    # online + target forward as one 2B-sample schedule (False: two B-sample passes)
        # optimizer step count and learning rate also live on the device, so a captured hipGraph of the
        # step stays valid while both advance
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.lr_dev = torch.full((1,), float(cfg.learning_rate), dtype=torch.float32, device=self.device)
        self._graph = None
        self._late_work = None
        self._side, self._prefetched = None, None        # cross-step teacher prefetch (step(prefetch=...) / capture(pipeline=True))
        # fp16 build (precision.set_precision("fp16"), the reference's --mixed_precision=fp16): the backward runs on S * d(loss) with the
        # GradScaler state in device memory (S = 65536 at start, x2 after 2000 finite steps, x0.5 and no update after a non-finite
        # gradient norm: torch.cuda.amp.GradScaler defaults, which accelerate uses at train_pcm_lora_sd15.py:1034) -- capturable.
        self._init_loss_scaler()
        self.comm_events = None           # a list: step_graphed appends (start, end) events around the gradient exchange
        self._seg = None                  # SegmentedGraph being captured (AdvDistiller.capture_adv at world_size > 1)
        # two-bucket gradient exchange (world_size > 1): the mid/up-block bucket is reduced while the down blocks back-propagate
        self.bucketed = os.environ.get("PCM_DDP_BUCKETS", "1") != "0"
        self.bucket_log = []        # (name, bytes on the wire, dtype) of every collective issued since it was last cleared, in issue order
        self.ema = None
        if cfg.ema_rate is not None:
            self.ema = lora.params.clone()

    def _init_loss_scaler(self):
        """device-side GradScaler state under precision "fp16" (None otherwise); shared by the UNet and MMDiT trainers"""
        self.loss_scale_dev = self.loss_good_dev = None
        if precision.precision() == "fp16":
            self.loss_scale_dev = torch.full((1,), float(os.environ.get("PCM_LOSS_SCALE", "65536")), dtype=torch.float32, device=self.device)
            self.loss_good_dev = torch.zeros(1, dtype=torch.int32, device=self.device)

    def _disc_adamw(self, d, adv_lr, adv_lr_dev):
        """AdamW(betas = (0, 0.999)) + global-norm clip over the discriminator heads ``d`` (already reduced over ranks); under the half build
        on loss-scaled gradients with the trainer's one GradScaler state, updated once per global step as accelerate does"""
        cfg = self.cfg
        d.step_dev += 1
        ops.sumsq(d.grads, d.gradsq)
        if self.loss_scale_dev is not None:
            ops.adamw_clip_step_scaled(d.params, d.grads, d.exp_avg, d.exp_avg_sq, d.gradsq, cfg.max_grad_norm, adv_lr, 0.0, 0.999,
                                       cfg.adam_epsilon, cfg.adam_weight_decay, 1.0 / self.world_size, d.step_dev, adv_lr_dev, self.loss_scale_dev)
            ops.loss_scale_update(self.loss_scale_dev, self.loss_good_dev, d.step_dev, d.gradsq)
        else:
            ops.adamw_clip_step(d.params, d.grads, d.exp_avg, d.exp_avg_sq, d.gradsq, cfg.max_grad_norm, adv_lr, 0.0, 0.999,
                                cfg.adam_epsilon, cfg.adam_weight_decay, 1, 1.0 / self.world_size, step_dev=d.step_dev, lr_dev=adv_lr_dev)
        d.repack()

    # ---- a2: timestep sampling (train_pcm_lora_sd15.py:1143-1155) ----
    def timesteps_for(self, index):
        start = self.tables.ddim_timesteps[index]
        t = torch.clamp(start - self.tables.topk, min=0)
        return start, t

    # ---- the part of the step that depends on nothing trainable: noisy latents, the frozen teacher's [cond; uncond] pass, the CFG DDIM step
    TARGET_KEYS = ("noisy", "start_t", "t_n", "eps_c", "eps_u", "x_prev64", "x_prev32")
    N_BATCH_KEY = 6          # leading tensors of a batch tuple whose addresses identify it (the rest: optional added-cond dicts)

    def teacher_targets(self, latents, prompt_embeds, uncond_prompt_embeds, noise, index, w, added_cond=None, uncond_added_cond=None):
        """train_pcm_lora_sd15.py:1143-1178 + :1217-1258: timesteps, add_noise, teacher cond / uncond forward, CFG-augmented DDIM solver step.
        A pure function of the batch and the FROZEN weights (the LoRA state is not read), which is what lets ``step(..., prefetch=next
        batch)`` run it for the next batch on a side stream while the student works on the current one."""
        def cat2(a, b_):
            return None if a is None else {k: torch.cat([a[k], (b_ if b_ is not None else a)[k]]) for k in a}
        cfg, T = self.cfg, self.tables
        B = latents.shape[0]
        start_t, t_n = self.timesteps_for(index)
        noisy = ops.add_noise(latents, noise, T.acp, start_t)                                   # :1178
        # teacher cond (+ uncond) in ONE batched forward (no grad, no LoRA) --------------------- :1217-1252
        with self._ode_scope():     # (the other 16-bit build for this pass when the teacher was packed in it: Distiller.__init__)
            if cfg.not_apply_cfg_solver:
                eps_c = self.teacher_ode.forward(noisy, start_t, prompt_embeds, added_cond=added_cond)
                eps_u = eps_c
            else:
                # (the halves share sample and timestep: the prefix up to the first cross-attention is computed once, UNet.forward dup_halves)
                both = self.teacher_ode.forward(torch.cat([noisy, noisy]), torch.cat([start_t, start_t]),
                                                torch.cat([prompt_embeds, uncond_prompt_embeds]), added_cond=cat2(added_cond, uncond_added_cond),
                                                dup_halves=DEDUP_TEACHER_PREFIX)
                eps_c, eps_u = both[:B], both[B:]
        x_prev64, x_prev32 = ops.cfg_ddim_step(eps_c, eps_u, noisy, start_t, index, w, T.acp, T.acp_prev)   # :1254-1258
        return dict(noisy=noisy, start_t=start_t, t_n=t_n, eps_c=eps_c, eps_u=eps_u, x_prev64=x_prev64, x_prev32=x_prev32)

    def forward_backward(self, latents, prompt_embeds, uncond_prompt_embeds, noise, index, w, backward=True, added_cond=None,
                         uncond_added_cond=None, grad_scale=1.0, zero_grad=True, on_late=None, targets=None):
        """Everything of the step before the gradient exchange: returns a dict of device tensors.
        ``added_cond`` / ``uncond_added_cond``: SDXL ``added_cond_kwargs`` ({'text_embeds': [B,1280], 'time_ids': [B,6]},
        train_pcm_lora_sdxl_adv.py:1113-1131, :1409-1421) for UNets with text_time conditioning; None for SD1.5.
        ``targets``: the result of ``teacher_targets`` for THIS batch when it was computed ahead (cross-step prefetch); None: computed here,
        first -- the online forward does not depend on it, the target forward does."""
        def cat2(a, b_):
            return None if a is None else {k: torch.cat([a[k], (b_ if b_ is not None else a)[k]]) for k in a}
        cfg, T = self.cfg, self.tables
        B = latents.shape[0]
        if targets is None:
            targets = self.teacher_targets(latents, prompt_embeds, uncond_prompt_embeds, noise, index, w, added_cond, uncond_added_cond)
        noisy, start_t, t_n, eps_c, eps_u, x_prev64, x_prev32 = (targets[k] for k in self.TARGET_KEYS)
        if self.fuse_online_target:
            # online student forward at t_{n+k} (grad, :1192) and target forward at (x_prev, t_n) (same online weights incl.
            # LoRA, no grad, :1261-1268) as ONE 2B-sample schedule: samples are independent, so each half is exactly the
            # separate forward; the backward runs on the online half of the tape only
            eps_st, tape2 = self.student.forward(torch.cat([noisy, x_prev32]), torch.cat([start_t, t_n]),
                                                 torch.cat([prompt_embeds, prompt_embeds]), save=True, save_half=True,
                                                 added_cond=cat2(added_cond, None))
            eps_s, eps_t = eps_st[:B], eps_st[B:]
            tape = self.student.tape_first_half(tape2)
        else:
            eps_s, tape = self.student.forward(noisy, start_t, prompt_embeds, save=True, added_cond=added_cond)
            eps_t = self.student.forward(x_prev32, t_n, prompt_embeds, added_cond=added_cond)
        model_pred, coef, end_t = ops.phase_jump(eps_s, noisy, start_t, index, T.acp, T.acp_prev, T.ddim_timesteps_prev,
                                                 T.edges, target_mode=False)                     # :1200-1212
        target, _, _ = ops.phase_jump(eps_t, x_prev64, t_n, index, T.acp, T.acp_prev, T.ddim_timesteps_prev, T.edges,
                                      target_mode=True)                                          # :1269-1280
        loss, d_eps = ops.consistency_loss(model_pred, target, coef, cfg.loss_type == "huber", cfg.huber_c, grad_scale=grad_scale)   # :1283-1293
        if self.loss_scale_dev is not None:
            ops.scale_by_dev(d_eps, self.loss_scale_dev)          # GradScaler.scale(loss).backward(): the whole backward carries S
        out = dict(loss=loss, noisy_model_input=noisy, noise_pred=eps_s, model_pred=model_pred, cond_teacher_output=eps_c,
                   uncond_teacher_output=eps_u, x_prev=x_prev64, target_noise_pred=eps_t, target=target,
                   start_timesteps=start_t, timesteps=t_n, end_timesteps=end_t)
        if not backward:
            out["tape"], out["d_eps"] = tape, d_eps
            return out
        if zero_grad:
            self.lora.zero_grad()
        self.student.backward(d_eps, tape, on_late=on_late)                                      # :1296
        return out

    # ---- cross-step prefetch of the teacher targets (round 6).  The frozen teacher's pass of batch k+1 depends on nothing the student's
    # work on batch k produces, and on this chip two independent passes issued on two HIP streams finish 6 % sooner than back to back
    # (blocks of one fill the CUs the other's under-filled deep-level launches leave idle: 69.9 -> 65.7 ms for backward + teacher,
    # profiles/r06_g_*).  Results are the same numbers: only the launch order changes.
    def _prefetch(self, batch):
        """issue teacher_targets(batch) on the side stream; ``batch`` = (latents, prompt_embeds, uncond_prompt_embeds, noise, index, w[, added_cond,
        uncond_added_cond]) of the NEXT step() call"""
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream() if self.device.type == "cuda" else False
        key = tuple(t.data_ptr() for t in batch[:self.N_BATCH_KEY])
        if self._side:
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                tg = self.teacher_targets(*batch)
        else:
            tg = self.teacher_targets(*batch)              # (host emulator: same call order, no streams)
        self._prefetched = (key, tg, batch)                # (the batch tensors stay referenced: the key is their addresses)

    def _take_prefetched(self, batch):
        pf, self._prefetched = getattr(self, "_prefetched", None), None
        if pf is None:
            return None
        if self._side:
            torch.cuda.current_stream().wait_stream(self._side)
        return pf[1] if pf[0] == tuple(t.data_ptr() for t in batch) else None

    def step(self, latents, prompt_embeds, uncond_prompt_embeds, noise, index, w, lr=None, update=True, added_cond=None,
             uncond_added_cond=None, accum=None, prefetch=None):
        """One distillation step on this rank's batch (eager launches).  All inputs are device tensors:
        latents/noise [B,4,H,W] fp32, prompt embeds [B,77,768], index [B] int64, w [B] fp32.
        Returns a dict of device tensors (no host sync).
        ``accum=(i, k)``: micro-batch i of k under ``--gradient_accumulation_steps k`` (``accelerator.accumulate``, :1120): the loss
        gradient is scaled by 1/k, gradients add up over the k calls, and exchange + clip + AdamW run with the last one only.
        ``prefetch``: the NEXT call's batch (same tuple order as this call's first six arguments [+ added_cond, uncond_added_cond]): its teacher
        targets are computed on a side stream beside this call's student work, and the next call picks them up (matched by tensor address)."""
        i, k = accum if accum is not None else (0, 1)
        bucket = self.world_size > 1 and update and i == k - 1 and self.lora.late_offset is not None and self.bucketed
        targets = self._take_prefetched((latents, prompt_embeds, uncond_prompt_embeds, noise, index, w))
        if prefetch is not None:
            self._prefetch(prefetch)
        out = self.forward_backward(latents, prompt_embeds, uncond_prompt_embeds, noise, index, w, backward=update, added_cond=added_cond,
                                    uncond_added_cond=uncond_added_cond, grad_scale=1.0 / k, zero_grad=(i == 0),
                                    on_late=self._all_reduce_late if bucket else None, targets=targets)
        if not update or i < k - 1:
            return out
        if lr is not None:
            self.lr_dev.fill_(float(lr))
        self.optimizer_step()
        out["grad_sumsq"] = self.lora.gradsq
        return out

    # ---- hipGraph replay of the step: ~5400 launches become two graph launches --------------------
    def capture(self, B, H=64, W=64, ctx_len=77, ctx_dim=768, added_cond=None, uncond_added_cond=None, pipeline=False):
        """Capture forward+backward and the optimizer as two hipGraphs around the (eager) gradient
        all-reduce.  The eager warm-up pass runs on scratch state: LoRA / Adam state is restored.
        ``added_cond`` / ``uncond_added_cond`` (SDXL text_time conditioning): example dicts; their tensors become static graph inputs
        that step_graphed refreshes.
        ``pipeline``: the captured step holds TWO branches -- the teacher targets of the NEXT batch (static inputs ``_static_next``, results
        into ``_tg_next``) on a forked stream, and this batch's student forward / backward on the targets a previous replay left
        (copied to ``_tg_cur`` before the fork); step_graphed(..., prefetch=next batch) feeds it."""
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        self._static = dict(latents=torch.zeros(B, 4, H, W, **f32), prompt_embeds=torch.zeros(B, ctx_len, ctx_dim, **f32),
                            uncond_prompt_embeds=torch.zeros(B, ctx_len, ctx_dim, **f32), noise=torch.zeros(B, 4, H, W, **f32),
                            index=torch.zeros(B, dtype=torch.int64, device=dev), w=torch.ones(B, **f32))
        if added_cond is not None:
            self._static["added_cond"] = {k: v.clone() for k, v in added_cond.items()}
            self._static["uncond_added_cond"] = {k: v.clone() for k, v in (uncond_added_cond or added_cond).items()}
        lo = self.lora
        state = [lo.params, lo.exp_avg, lo.exp_avg_sq, self.step_dev, self.lr_dev]
        if self.loss_scale_dev is not None:
            state += [self.loss_scale_dev, self.loss_good_dev]
        saved = [t.clone() for t in state]
        if self.ema is not None:
            saved.append(self.ema.clone())
        count = self.step_count
        self._pipeline, self._pipe_key = bool(pipeline), None
        if pipeline:
            self._static_next = {k: ({kk: vv.clone() for kk, vv in v.items()} if isinstance(v, dict) else v.clone()) for k, v in self._static.items()}
            self._pipe_side = torch.cuda.Stream()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up: lazy init, allocator pools
            if pipeline:
                tg = self.teacher_targets(**self._static_next)
                self._tg_next = {k: v.clone() for k, v in tg.items()}
                self._tg_cur = {k: v.clone() for k, v in tg.items()}
                self.forward_backward(**self._static, targets=self._tg_cur)
            else:
                self.forward_backward(**self._static)
            self._optimizer_apply()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

        def body(on_late=None, join_before_cut=None):
            """what one replay does (non-pipelined: the plain step)"""
            if not pipeline:
                return self.forward_backward(**self._static, on_late=on_late)
            cur_s = torch.cuda.current_stream()
            for k in self.TARGET_KEYS:                     # the targets a previous replay (or the eager prologue) left for THIS batch
                self._tg_cur[k].copy_(self._tg_next[k])
            self._pipe_side.wait_stream(cur_s)             # fork: behind the copies, so the branch may overwrite _tg_next
            with torch.cuda.stream(self._pipe_side):
                tgn = self.teacher_targets(**self._static_next)
                for k in self.TARGET_KEYS:
                    self._tg_next[k].copy_(tgn[k])
            out_ = self.forward_backward(**self._static, on_late=on_late, targets=self._tg_cur)
            torch.cuda.current_stream().wait_stream(self._pipe_side)      # join (a capture must end with every forked stream joined)
            return out_
        self._g_fb, self._g_opt, self._g_fb2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), None
        split = (self.world_size > 1 and self.bucketed and lo.late_offset is not None) or os.environ.get("PCM_SPLIT_GRAPH") == "1"
        # thread_local: the RCCL watchdog thread of a multi-rank job may query its events while this thread captures
        if not split:
            with torch.cuda.graph(self._g_fb, capture_error_mode="thread_local"):
                self._static_out = body()
        else:
            # data parallel: the forward + backward is captured as TWO graphs cut where the backward leaves the mid block, so that the
            # all-reduce of the up/mid-block gradient bucket (not captured) is enqueued between them and overlaps the second graph
            self._g_fb2 = torch.cuda.CUDAGraph()
            torch.cuda.synchronize(); gc.collect(); torch.cuda.empty_cache()
            cap = torch.cuda.Stream()
            cap.wait_stream(torch.cuda.current_stream())

            def cut():
                if pipeline:       # the teacher branch of the next batch joins the first graph (it is long done: it started with the forward)
                    torch.cuda.current_stream().wait_stream(self._pipe_side)
                self._g_fb.capture_end()
                self._g_fb2.capture_begin(pool=self._g_fb.pool(), capture_error_mode="thread_local")
            with torch.cuda.stream(cap):
                self._g_fb.capture_begin(capture_error_mode="thread_local")
                self._static_out = body(on_late=cut)
                self._g_fb2.capture_end()
            torch.cuda.current_stream().wait_stream(cap)
        with torch.cuda.graph(self._g_opt, pool=self._g_fb.pool(), capture_error_mode="thread_local"):
            self._optimizer_apply()
        for dst, src in zip(state, saved):
            dst.copy_(src)
        if self.ema is not None:
            self.ema.copy_(saved[-1])
        self.step_count = count
        lo.repack()
        self._static_out["grad_sumsq"] = lo.gradsq
        self._graph = True

    def _fill_static(self, st, latents, prompt_embeds, uncond_prompt_embeds, noise, index, w, added_cond=None, uncond_added_cond=None):
        st["latents"].copy_(latents); st["prompt_embeds"].copy_(prompt_embeds)
        st["uncond_prompt_embeds"].copy_(uncond_prompt_embeds); st["noise"].copy_(noise)
        st["index"].copy_(index); st["w"].copy_(w)
        for name, val in (("added_cond", added_cond), ("uncond_added_cond", uncond_added_cond)):
            if val is not None:
                for k, v in val.items():
                    st[name][k].copy_(v)

    def _pipe_feed(self, batch, added_cond, uncond_added_cond, prefetch):
        """pipelined capture, before a replay: make sure THIS batch's teacher targets are where the graph expects them, and hand the next
        batch to the graph's teacher branch"""
        key = tuple(t.data_ptr() for t in batch)
        if self._pipe_key != key:
            # prologue (first call, or a batch that was not announced): this batch's teacher targets, eagerly
            tg = self.teacher_targets(*batch, added_cond, uncond_added_cond) if self.N_BATCH_KEY == 6 else self.teacher_targets(*batch)
            for k in self.TARGET_KEYS:
                self._tg_next[k].copy_(tg[k])
        self._pipe_key = None
        if prefetch is not None:
            self._fill_static(self._static_next, *prefetch)
            self._pipe_key = tuple(t.data_ptr() for t in prefetch[:self.N_BATCH_KEY])

    def step_graphed(self, latents, prompt_embeds, uncond_prompt_embeds, noise, index, w, lr=None, added_cond=None, uncond_added_cond=None,
                     prefetch=None):
        """Same as step() through the captured graphs.  Returned tensors are the graph's static outputs
        (overwritten by the next call).  ``prefetch`` (capture(pipeline=True)): the NEXT call's batch, as in step()."""
        st = self._static
        self._fill_static(st, latents, prompt_embeds, uncond_prompt_embeds, noise, index, w, added_cond, uncond_added_cond)
        if getattr(self, "_pipeline", False):
            self._pipe_feed((latents, prompt_embeds, uncond_prompt_embeds, noise, index, w), added_cond, uncond_added_cond, prefetch)
        if lr is not None:
            self.lr_dev.fill_(float(lr))
        self._g_fb.replay()
        if self._g_fb2 is not None:
            if self.world_size > 1:
                self._all_reduce_late()
            self._g_fb2.replay()
        ev = self.comm_events
        if ev is not None and self.world_size > 1:   # bench.py: exposed time of the exchange = what sits between backward and optimizer
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.all_reduce_grads()
            e1.record()
            ev.append((e0, e1))
        else:
            self.all_reduce_grads()
        self.step_count += 1
        self._g_opt.replay()
        return self._static_out

    def _all_reduce_late(self):
        """called by the backward when the up / mid-block gradients are final: their bucket goes out (async, on RCCL's own stream, ordered
        behind the launches issued so far) while the down blocks still back-propagate"""
        self._late_work = torch.distributed.all_reduce(self.lora.grads[self.lora.late_offset:], op=torch.distributed.ReduceOp.SUM,
                                                       group=self.pg, async_op=True)
        self._log_bucket("lora[late: mid + up blocks]", (self.lora.grads.numel() - self.lora.late_offset) * 4, "fp32")

    def _log_bucket(self, name, nbytes, dtype):
        log_ = getattr(self, "bucket_log", None)          # (subclasses / test doubles that do not run Distiller.__init__)
        if log_ is None:
            log_ = self.bucket_log = []
        if len(log_) < 64:        # (a diagnostic of ONE step's issue order: bench.py clears it before the step it reports)
            log_.append((name, int(nbytes), dtype))

    def all_reduce_grads(self):
        """DDP exchange: all-reduce (sum) of the flat 67 M-element fp32 LoRA gradient buffer over RCCL/xGMI, as ONE collective or --
        when the backward already started the late bucket -- as the remaining early (down-block) bucket plus a wait; the 1/world_size mean is
        folded into the AdamW kernel's grad_scale."""
        if self.world_size <= 1:
            return
        if getattr(self, "_late_work", None) is not None:
            torch.distributed.all_reduce(self.lora.grads[:self.lora.late_offset], op=torch.distributed.ReduceOp.SUM, group=self.pg)
            self._log_bucket("lora[early: down blocks]", self.lora.late_offset * 4, "fp32")
            self._late_work.wait()
            self._late_work = None
        else:
            torch.distributed.all_reduce(self.lora.grads, op=torch.distributed.ReduceOp.SUM, group=self.pg)
            self._log_bucket("lora[all]", self.lora.grads.numel() * 4, "fp32")

    def _collective(self, fn):
        """issue a collective -- or, while a SegmentedGraph is being captured, cut the graph there and make ``fn`` the host action between
        the two segments (the captured pass itself never communicates)"""
        if self._seg is not None:
            self._seg.cut(fn)
        else:
            fn()

    def optimizer_step(self):
        self._collective(self.all_reduce_grads)
        if self._seg is None:
            self.step_count += 1
        self._optimizer_apply()

    def _optimizer_apply(self):
        """clip + AdamW + EMA + operand repack on the (already reduced) flat gradient buffer; the step
        count and lr are read from device memory (capturable)."""
        cfg, lo = self.cfg, self.lora
        gscale = 1.0 / self.world_size
        self.step_dev += 1
        ops.sumsq(lo.grads, lo.gradsq)                                                           # :1298 clip_grad_norm_
        if self.loss_scale_dev is not None:      # unscale + clip + step (skipped when the norm is not finite), then GradScaler.update()
            ops.adamw_clip_step_scaled(lo.params, lo.grads, lo.exp_avg, lo.exp_avg_sq, lo.gradsq, cfg.max_grad_norm, cfg.learning_rate,
                                       cfg.adam_beta1, cfg.adam_beta2, cfg.adam_epsilon, cfg.adam_weight_decay, gscale,
                                       self.step_dev, self.lr_dev, self.loss_scale_dev)
            ops.loss_scale_update(self.loss_scale_dev, self.loss_good_dev, self.step_dev, lo.gradsq)
        else:
            ops.adamw_clip_step(lo.params, lo.grads, lo.exp_avg, lo.exp_avg_sq, lo.gradsq, cfg.max_grad_norm,
                                cfg.learning_rate, cfg.adam_beta1, cfg.adam_beta2, cfg.adam_epsilon,
                                cfg.adam_weight_decay, 1, gscale, step_dev=self.step_dev, lr_dev=self.lr_dev)   # :1299
        if self.ema is not None:     # (half build: not after a step GradScaler semantics skipped)
            ops.ema_update(self.ema, lo.params, cfg.ema_rate, gradsq=lo.gradsq if self.loss_scale_dev is not None else None)
        lo.repack()

    def applied_steps(self):
        """Host read of the number of optimizer steps actually APPLIED (device counter: with loss-scaled half gradients a step whose global
        norm overflowed is skipped and pcm_loss_scale_update takes the count back).  accelerate steps the lr scheduler only when the
        optimizer step was not skipped: the CLIs position their schedule with this under --mixed_precision=fp16 (one host sync per step, as
        GradScaler's own found_inf read); bf16 / fp32 runs never skip and use the host counter."""
        return int(self.step_dev.item())

    def grad_norm(self):
        """Host read of the last global grad norm (forces a sync; for logging only)."""
        s = float(self.loss_scale_dev.item()) if self.loss_scale_dev is not None else 1.0
        return math.sqrt(float(self.lora.gradsq.item())) / self.world_size / s


class AdvDistiller(Distiller):
    """PCM-LoRA + latent adversarial consistency (reference: train_pcm_lora_sd15_adv.py:1288-1431).
    Even ``global_step``: discriminator update only; odd: student update with loss_cm + adv_weight * g_loss."""

    def __init__(self, weights, lora, cfg, discriminator, adv_weight=0.1, adv_lr=1e-5, world_size=1, process_group=None, teacher_weights=None,
                 head_grad_exchange=None):
        """``head_grad_exchange``: "fp32" (default; the reference's DDP reduces fp32 gradients) or "bf16" (the 2.66 GB of head
        gradients of a discriminator step cross xGMI as 1.33 GB -- each bucket is rounded to bfloat16, summed by the collective and
        widened back; the D update then differs from the fp32 exchange by the 16-bit rounding of the per-rank gradients, bounded in
        tests/test_ddp_gloo.py).  Default from PCM_HEAD_GRAD_EXCHANGE.  bfloat16 build only (half gradients are loss-scaled)."""
        super().__init__(weights, lora, cfg, world_size, process_group, teacher_weights=teacher_weights)
        self.disc, self.adv_weight, self.adv_lr = discriminator, adv_weight, adv_lr
        self.head_grad_exchange = head_grad_exchange or os.environ.get("PCM_HEAD_GRAD_EXCHANGE", "fp32")
        if self.head_grad_exchange not in ("fp32", "bf16"):
            raise ValueError("head_grad_exchange must be 'fp32' or 'bf16'")
        self.adv_lr_dev = torch.full((1,), float(adv_lr), dtype=torch.float32, device=self.device)
        if os.environ.get("PCM_ADV_FUSE_PASSES") == "0":       # A/B switch: online and target forward as two B-sample passes (round <= 5)
            self.fuse_online_target = False

    def step_adv(self, global_step, latents, prompt_embeds, uncond_prompt_embeds, noise, index, w, noise_fake, noise_real, adv_u,
                 lr=None, added_cond=None, uncond_added_cond=None, targets=None, prefetch=None):
        """adv_u [B] in [0,1): adv_timesteps = end_timesteps + floor(adv_u * (T // multiphase))   (:1288-1298).
        ``targets`` / ``prefetch``: as in Distiller.forward_backward / step -- the ODE-solver teacher's results for this batch computed ahead,
        and the next call's batch whose teacher pass is issued on a side stream beside this call's work."""
        cfg, T, disc = self.cfg, self.tables, self.disc
        B = latents.shape[0]
        ac, uac, taps = added_cond, uncond_added_cond, getattr(disc, "taps", True)

        def cat2(a, b_):
            return None if a is None else {k: torch.cat([a[k], (b_ if b_ is not None else a)[k]]) for k in a}
        is_d = (global_step % 2 == 0)                                                           # :1375 / :1399
        if targets is None:
            targets = self._take_prefetched((latents, prompt_embeds, uncond_prompt_embeds, noise, index, w))
        if prefetch is not None:
            self._prefetch(prefetch)
        if targets is None:     # the ODE-solver teacher pass (sd15_adv.py:1312 ``torch.autocast("cuda")``) + the CFG DDIM step (:1307-1352)
            targets = self.teacher_targets(latents, prompt_embeds, uncond_prompt_embeds, noise, index, w, ac, uac)
        noisy, start_t, t_n, x_prev64, x_prev32 = (targets[k] for k in ("noisy", "start_t", "t_n", "x_prev64", "x_prev32"))
        eps_t = None
        if self.fuse_online_target:
            # online forward at t_{n+k} and target forward at (x_prev, t_n) as ONE 2B-sample schedule, as in Distiller.forward_backward (the
            # teacher's results are at hand before either); discriminator steps back-propagate nothing through the student: no tape
            eps_st = self.student.forward(torch.cat([noisy, x_prev32]), torch.cat([start_t, t_n]), torch.cat([prompt_embeds, prompt_embeds]),
                                          save=not is_d, save_half=not is_d, added_cond=cat2(ac, None))
            tape = None
            if not is_d:
                eps_st, tape2 = eps_st
                tape = self.student.tape_first_half(tape2)
            eps_s, eps_t = eps_st[:B], eps_st[B:]
        elif is_d:
            eps_s, tape = self.student.forward(noisy, start_t, prompt_embeds, added_cond=ac), None
        else:
            eps_s, tape = self.student.forward(noisy, start_t, prompt_embeds, save=True, added_cond=ac)
        model_pred, coef, end_t = ops.phase_jump(eps_s, noisy, start_t, index, T.acp, T.acp_prev, T.ddim_timesteps_prev, T.edges, target_mode=False)
        span = cfg.num_train_timesteps // cfg.multiphase
        adv_t = end_t + torch.clamp((adv_u * span).long(), max=span - 1)
        fake_adv, sr = ops.noise_travel(model_pred, noise_fake, T.acp, end_t, adv_t)            # :1303-1305
        if eps_t is None:
            eps_t = self.student.forward(x_prev32, t_n, prompt_embeds, added_cond=ac)
        target, _, _ = ops.phase_jump(eps_t, x_prev64, t_n, index, T.acp, T.acp_prev, T.ddim_timesteps_prev, T.edges, target_mode=True)
        out = dict(model_pred=model_pred, target=target, end_timesteps=end_t, adv_timesteps=adv_t, fake_adv=fake_adv, is_d=is_d)
        if is_d:
            real_adv, _ = ops.noise_travel(target, noise_real, T.acp, end_t, adv_t)             # :1379-1381
            feats = self.teacher.forward(torch.cat([fake_adv, real_adv]), torch.cat([adv_t, adv_t]),
                                         torch.cat([prompt_embeds, prompt_embeds]), features=taps, added_cond=cat2(ac, None))
            logits, dtape = disc.forward(feats, save=True)
            disc.grads.zero_()
            self._disc_works = []
            bucket = (lambda a, b_: self._collective(lambda: self._disc_bucket(a, b_))) if (self.world_size > 1 or SEG_FORCE) else None
            out["d_loss"] = disc.d_loss_backward(logits, dtape, B, on_bucket=bucket, loss_scale_dev=self.loss_scale_dev)           # :1383-1391
            out["real_adv"] = real_adv
            self._disc_optimizer_step()
            return out
        feats, utape = self.teacher.forward(fake_adv, adv_t, prompt_embeds, features=taps, save=True, added_cond=ac)
        logits, dtape = disc.forward(feats, save=True)
        g_loss, d_feats = disc.g_loss_backward(logits, dtape, grad_scale=self.adv_weight, loss_scale_dev=self.loss_scale_dev)       # :1414-1421
        d_fake = self.teacher.backward(None, utape, d_feats=d_feats, need_input_grad=True)
        loss_cm, d_eps = ops.consistency_loss(model_pred, target, coef, cfg.loss_type == "huber", cfg.huber_c)
        if self.loss_scale_dev is not None:      # half build: both terms of loss_cm + adv_weight * g_loss carry the loss scale (d_fake already does)
            ops.scale_by_dev(d_eps, self.loss_scale_dev)
        ops.scale_add_rows(d_eps, d_fake, sr, coef)          # d fake_adv/d model_pred = sqrt(r); d model_pred/d eps = coef
        out.update(loss_cm=loss_cm, g_loss=g_loss, d_fake_adv=d_fake, d_eps=d_eps)
        self.lora.zero_grad()
        self.student.backward(d_eps, tape)
        if lr is not None:
            self.lr_dev.fill_(float(lr))
        self.optimizer_step()
        out["grad_sumsq"] = self.lora.gradsq
        return out

    def capture_adv(self, B, H=64, W=64, ctx_len=77, ctx_dim=768, added_cond=None, uncond_added_cond=None, pipeline=False):
        """Capture the discriminator step and the generator step as two hipGraphs (single GPU: the steps contain no host decision and
        no collective; ~3200 / ~4500 launches each, whose enqueue time otherwise bounds the adversarial step).  Warm-up runs on scratch
        state: LoRA, heads and both optimizers are restored afterwards.
        ``pipeline`` (single GPU): both graphs carry the ODE-solver teacher pass of the NEXT batch as a forked branch, as in
        Distiller.capture(pipeline=True); step_adv_graphed(..., prefetch=next batch) feeds it.  At the adversarial configs' batch sizes (2 per
        GPU) the student / feature passes fill a fraction of the chip, which is what the branch runs in."""
        dev, d, lo = self.device, self.disc, self.lora
        f32 = dict(dtype=torch.float32, device=dev)
        st = dict(latents=torch.zeros(B, 4, H, W, **f32), prompt_embeds=torch.zeros(B, ctx_len, ctx_dim, **f32),
                  uncond_prompt_embeds=torch.zeros(B, ctx_len, ctx_dim, **f32), noise=torch.zeros(B, 4, H, W, **f32),
                  index=torch.zeros(B, dtype=torch.int64, device=dev), w=torch.ones(B, **f32),
                  noise_fake=torch.zeros(B, 4, H, W, **f32), noise_real=torch.zeros(B, 4, H, W, **f32), adv_u=torch.zeros(B, **f32))
        if added_cond is not None:
            st["added_cond"] = {k: v.clone() for k, v in added_cond.items()}
            st["uncond_added_cond"] = {k: v.clone() for k, v in (uncond_added_cond or added_cond).items()}
        self._adv_static = st
        pipeline = bool(pipeline) and self.world_size == 1 and not SEG_FORCE
        self._pipeline, self._pipe_key = pipeline, None
        if pipeline:
            self._static_next = {k: ({kk: vv.clone() for kk, vv in v.items()} if isinstance(v, dict) else v.clone()) for k, v in st.items()
                                 if k not in ("noise_fake", "noise_real", "adv_u")}
            self._pipe_side = torch.cuda.Stream()
        keep = (lo.params, lo.exp_avg, lo.exp_avg_sq, self.step_dev, self.lr_dev, d.params, d.exp_avg, d.exp_avg_sq, d.step_dev)
        if self.loss_scale_dev is not None:
            keep += (self.loss_scale_dev, self.loss_good_dev)
        saved = [t.clone() for t in keep]
        count = self.step_count
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            if pipeline:
                tg = self.teacher_targets(**self._static_next)
                self._tg_next = {k: v.clone() for k, v in tg.items()}
                self._tg_cur = {k: v.clone() for k, v in tg.items()}
            self.step_adv(0, **st)
            self.step_adv(1, **st)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

        def body(gs):
            if not pipeline:
                return self.step_adv(gs, **st)
            for k in self.TARGET_KEYS:                     # the targets a previous replay (or the eager prologue) left for THIS batch
                self._tg_cur[k].copy_(self._tg_next[k])
            self._pipe_side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._pipe_side):
                tgn = self.teacher_targets(**self._static_next)
                for k in self.TARGET_KEYS:
                    self._tg_next[k].copy_(tgn[k])
            out_ = self.step_adv(gs, **st, targets=self._tg_cur)
            torch.cuda.current_stream().wait_stream(self._pipe_side)
            return out_
        if self.world_size == 1 and not SEG_FORCE:
            self._g_d, self._g_g = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g_d):
                self._out_d = body(0)
            with torch.cuda.graph(self._g_g, pool=self._g_d.pool()):
                self._out_g = body(1)
        else:
            # data parallel: the D step is cut at its nine head-gradient buckets (each all-reduce goes out between two segments and
            # overlaps the heads that are still back-propagating) and in front of the head optimizer; the G step in front of the LoRA
            # gradient exchange.  10 + 1 and 1 + 1 graph launches per step instead of ~3200 / ~4500 eager launches.
            torch.cuda.synchronize(); gc.collect(); torch.cuda.empty_cache()
            cap = torch.cuda.Stream()
            cap.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cap):
                try:
                    self._g_d = self._seg = SegmentedGraph()
                    self._seg.begin()
                    self._out_d = self.step_adv(0, **st)
                    self._seg.end()
                    self._g_g = self._seg = SegmentedGraph(pool=self._g_d.pool)
                    self._seg.begin()
                    self._out_g = self.step_adv(1, **st)
                    self._seg.end()
                except BaseException:
                    # a failure inside a segment (OOM, a collective issued outside _collective): no capture may stay open and _seg must not
                    # keep routing later collectives into cut() of a dead graph; the caller falls back to eager steps
                    if self._seg is not None:
                        self._seg.abort()
                    self._g_d = self._g_g = None
                    raise
                finally:
                    self._seg = None
                    self._disc_works = []
            torch.cuda.current_stream().wait_stream(cap)
        for dst, src in zip(keep, saved):
            dst.copy_(src)
        self.step_count = count
        lo.repack()
        d.repack()

    def step_adv_graphed(self, global_step, latents, prompt_embeds, uncond_prompt_embeds, noise, index, w, noise_fake, noise_real, adv_u,
                         lr=None, added_cond=None, uncond_added_cond=None, prefetch=None):
        """step_adv through the captured graphs; the returned tensors are the graph's static outputs.  ``prefetch``
        (capture_adv(pipeline=True)): the NEXT call's (latents, prompt_embeds, uncond_prompt_embeds, noise, index, w[, added_cond, uncond_added_cond])."""
        st = self._adv_static
        for k, v in (("latents", latents), ("prompt_embeds", prompt_embeds), ("uncond_prompt_embeds", uncond_prompt_embeds), ("noise", noise),
                     ("index", index), ("w", w), ("noise_fake", noise_fake), ("noise_real", noise_real), ("adv_u", adv_u)):
            st[k].copy_(v)
        for name, val in (("added_cond", added_cond), ("uncond_added_cond", uncond_added_cond)):
            if val is not None:
                for k, v in val.items():
                    st[name][k].copy_(v)
        if getattr(self, "_pipeline", False):
            self._pipe_feed((latents, prompt_embeds, uncond_prompt_embeds, noise, index, w), added_cond, uncond_added_cond, prefetch)
        if global_step % 2 == 0:
            self._g_d.replay()
            return self._out_d
        if lr is not None:
            self.lr_dev.fill_(float(lr))
        self._g_g.replay()
        self.step_count += 1
        return self._out_g

    def _disc_bucket(self, off0, off1):
        """the heads of one tapped feature are done: their 0.07-0.47 GB of fp32 gradients go out (async) while the remaining heads still
        back-propagate -- 9 collectives per discriminator step instead of one 2.66 GB all-reduce at its end (fp32 like the
        reference's DDP, and only on discriminator steps: generator steps never call this)"""
        if self.world_size > 1:
            g = self.disc.grads[off0:off1]
            if self.head_grad_exchange == "bf16" and self.loss_scale_dev is None:
                h = ops.cast_bf16(g)                   # C-ABI cast kernel into a staging buffer that lives until the bucket is widened back
                wk = torch.distributed.all_reduce(h, op=torch.distributed.ReduceOp.SUM, group=self.pg, async_op=True)
                self._disc_works.append((wk, h, g))
                self._log_bucket("heads[%d:%d]" % (off0, off1), h.numel() * 2, "bf16")
            else:
                self._disc_works.append((torch.distributed.all_reduce(g, op=torch.distributed.ReduceOp.SUM, group=self.pg, async_op=True), None, g))
                self._log_bucket("heads[%d:%d]" % (off0, off1), g.numel() * 4, "fp32")

    def _disc_finish_exchange(self):
        """wait for the per-tap buckets launched from inside the head backward (or, if none were, reduce the whole buffer)"""
        d = self.disc
        if self.world_size <= 1:
            return
        if getattr(self, "_disc_works", None):
            for wk, h, g in self._disc_works:
                wk.wait()
                if h is not None:        # 16-bit exchange: the summed bucket back into the fp32 gradient buffer the optimizer reads
                    ops.cast_f32(h, out=g)
            self._disc_works = []
        else:
            torch.distributed.all_reduce(d.grads, op=torch.distributed.ReduceOp.SUM, group=self.pg)

    def _disc_optimizer_step(self):
        """optimizer_discriminator (:1026-1032): AdamW(lr=adv_lr, betas=(0, 0.999)), global-norm clip over the heads."""
        cfg, d = self.cfg, self.disc
        if self.world_size > 1 or SEG_FORCE:
            self._collective(self._disc_finish_exchange)
        self._disc_adamw(d, self.adv_lr, self.adv_lr_dev)
