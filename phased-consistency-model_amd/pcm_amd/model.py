"""SD1.5 UNet forward and LoRA-only backward as an explicit schedule of C-ABI kernel launches.

No autograd and no tracing compiler: the UNet is a static graph, so forward and backward are
written out once as launch sequences on torch's current HIP stream (capturable in a hipGraph).
Activations are channels-last bf16 ``[B, H*W, C]``; base weights are frozen, packed once into bf16
MFMA operand layouts in BOTH orientations (forward and dgrad; 2 x 1.7 GB resident in HBM), the
LoRA factors live in one flat fp32 buffer (master / grad / Adam m,v) with bf16 operand copies that
are refreshed after every optimizer step.

Mirrors diffusers' ``UNet2DConditionModel.forward(sample, timestep, encoder_hidden_states).sample``
(reference wiring: discriminator_sd15.py:84-345) and peft's ``lora.Linear`` / ``lora.Conv2d``
(train_pcm_lora_sd15.py:866-885).
"""
import math
import os
from collections import OrderedDict

import torch

from . import capi, ops
from .ops import Seg
from .unet_spec import UNetConfig, lora_target_modules, param_spec

from .precision import act_dtype as _act_dtype, precision  # noqa: E402



# ----------------------------------------------------------------------------------------------
# frozen base weights, packed
# ----------------------------------------------------------------------------------------------
GEGLU_GROUP = 2     # interleave granularity of the fused-GEGLU operands (include/pcm_hip.h, PCM_ACT_GEGLU)


def geglu_perm(inner, device="cpu"):
    """row order of the interleaved [values | gates] projection: G values, their G gates, the next G values, ..."""
    j = torch.arange(inner // GEGLU_GROUP, device=device)
    e = torch.arange(GEGLU_GROUP, device=device)
    return torch.stack([(GEGLU_GROUP * j)[:, None] + e, inner + (GEGLU_GROUP * j)[:, None] + e], 1).reshape(-1)


class PackedLayer:
    """One Linear / conv1x1 / conv3x3 of the base model in MFMA operand layouts."""
    __slots__ = ("kind", "N", "K", "C", "w_fwd", "w_bwd", "bias", "w_geglu", "bias_geglu")

    def __init__(self, w, bias, device, need_bwd, scale=1.0):
        """``scale``: folded into the packed operands (both orientations) and the bias -- ONE rounding of scale * w from the fp32 values
        (the attention query projections carry the softmax scale: UNetWeights)"""
        w = w.to(device=device, dtype=torch.float32).contiguous()
        self.N = w.shape[0]
        self.bias = None if bias is None else (bias.to(device=device, dtype=torch.float32) * scale).contiguous()
        if w.dim() == 4 and w.shape[-1] == 3:
            assert scale == 1.0
            self.kind, self.C = "conv3", w.shape[1]
            self.K = 9 * self.C
            self.w_fwd, self.w_bwd = ops.pack_conv3x3(w, True, need_bwd)
        else:
            self.kind, self.C = "lin", 0
            self.K = w.numel() // self.N
            self.w_fwd, self.w_bwd = ops.pack_linear(w.view(self.N, self.K), True, need_bwd, scale=scale)
        self.w_geglu = self.bias_geglu = None

    def pack_geglu(self):
        """Extra forward operand for the fused GEGLU epilogue (PCM_ACT_GEGLU): rows interleaved [2 values, their 2 gates, ...] so that the
        four consecutive channels one MFMA lane accumulates are one output pair's values and gates.  A row permutation of the packed bf16 rows."""
        inner = self.N // 2
        perm = geglu_perm(inner, self.w_fwd.device)
        self.w_geglu = self.w_fwd.view(self.N, self.K)[perm].contiguous()
        self.bias_geglu = None if self.bias is None else self.bias[perm].contiguous()


class UNetWeights:
    """Frozen SD1.5 UNet weights (diffusers key names in, packed operands out).  One instance is
    shared by the teacher, the online student and the target passes (the reference loads the same
    checkpoint twice: train_pcm_lora_sd15.py:840-851)."""

    def __init__(self, cfg: UNetConfig, state_dict, device, need_bwd=True):
        self.cfg, self.device = cfg, torch.device(device)
        self.format = precision()            # the 16-bit format the operands below are packed in (pcm_amd/precision.py)
        spec = param_spec(cfg)
        missing = [k for k, _ in spec if k not in state_dict]
        if missing:
            raise KeyError(f"UNetWeights: {len(missing)} missing keys, e.g. {missing[:3]}")
        for k, shp in spec:
            if tuple(state_dict[k].shape) != tuple(shp):
                raise ValueError(f"UNetWeights: {k} has shape {tuple(state_dict[k].shape)}, expected {shp}")
        self.layers, self.norms, self.q_scale = {}, {}, {}
        f32 = dict(device=self.device, dtype=torch.float32)
        for k, shp in spec:
            if not k.endswith(".weight"):
                continue
            path = k[:-7]
            leaf = path.rsplit(".", 1)[-1]
            if leaf.startswith("norm") or leaf == "conv_norm_out":
                self.norms[path] = (state_dict[k].to(**f32).contiguous(), state_dict[path + ".bias"].to(**f32).contiguous())
            elif path in ("conv_in", "conv_out"):
                continue
            else:
                # attention query projections are packed times head_dim^-1/2 * log2(e): the attention kernels take a PRE-SCALED query
                # (csrc/attention_ps.hip) and q is stored once, already in the log2 domain; LoraState scales the matching s*B copies
                qs = cfg.q_scale_of_path(path, shp[0])
                self.layers[path] = PackedLayer(state_dict[k], state_dict.get(path + ".bias"), self.device, need_bwd, scale=qs or 1.0)
                if qs is not None:
                    self.q_scale[path] = qs
                if path.endswith("ff.net.0.proj"):
                    self.layers[path].pack_geglu()
        # self-attention q/k/v of the frozen (LoRA-free, no-grad) pass as ONE projection: rows [Wq; Wk; Wv], the
        # activation is read once and attention reads q/k/v in place with row stride 3C
        self.qkv, self.qkv_bwd = {}, {}
        for path in [p_ for p_ in self.layers if p_.endswith("attn1.to_q")]:
            base = path[:-4]
            lq, lk, lv = (self.layers[base + n] for n in ("to_q", "to_k", "to_v"))
            if lq.bias is None and lk.bias is None and lv.bias is None and lq.K == lk.K == lv.K:
                self.qkv[base] = torch.cat([lq.w_fwd.view(lq.N, lq.K), lk.w_fwd.view(lk.N, lk.K), lv.w_fwd.view(lv.N, lv.K)]).contiguous()
                if need_bwd and lq.N == lk.N == lv.N:   # dgrad operand of the fused projection: [K][3N] = [Wq^T | Wk^T | Wv^T]
                    self.qkv_bwd[base] = torch.cat([lq.w_bwd.view(lq.K, lq.N), lk.w_bwd.view(lk.K, lk.N), lv.w_bwd.view(lv.K, lv.N)], dim=1).contiguous()
        # cross-attention K / V of the frozen pass: the text embeddings are the SAME input for every transformer block, so all their to_k / to_v
        # projections are ONE GEMM per pass (rows [Wk_0; Wv_0; Wk_1; ...], 32 launches of 14-17 us -> one at SD1.5 size); attention reads
        # each block's K / V in place with the wide row stride
        self.kv_cat, self.kv_off = None, {}
        kv_paths = [p_[:-4] for p_ in self.layers if p_.endswith("attn2.to_k")]
        if kv_paths and all(self.layers[b + n].bias is None for b in kv_paths for n in ("to_k", "to_v")) and \
                len({self.layers[b + "to_k"].K for b in kv_paths}) == 1:
            off, rows = 0, []
            for b in kv_paths:
                lk, lv = self.layers[b + "to_k"], self.layers[b + "to_v"]
                self.kv_off[b] = (off, lk.N)
                rows += [lk.w_fwd.view(lk.N, lk.K), lv.w_fwd.view(lv.N, lv.K)]
                off += lk.N + lv.N
            self.kv_cat = torch.cat(rows).contiguous()
        # every resnet's time_emb_proj reads the same [B, K] input: ONE GEMM per pass (rows [W_0; W_1; ...], 22 launches -> one at SD1.5 size;
        # UNet._temb_all scatters the [B, sum N] result into per-resnet contiguous row vectors with one segmented-pack launch)
        self.temb_cat, self.temb_bias, self.temb_off = None, None, {}
        tp = [p_ for p_ in self.layers if p_.endswith("time_emb_proj")]
        if len(tp) >= 2 and all(self.layers[t].kind == "lin" and self.layers[t].bias is not None for t in tp) and len({self.layers[t].K for t in tp}) == 1:
            off = 0
            for t in tp:
                self.temb_off[t] = (off, self.layers[t].N)
                off += self.layers[t].N
            self.temb_cat = torch.cat([self.layers[t].w_fwd.view(self.layers[t].N, self.layers[t].K) for t in tp]).contiguous()
            self.temb_bias = torch.cat([self.layers[t].bias.reshape(-1) for t in tp]).contiguous()
        self.conv_in = (state_dict["conv_in.weight"].to(**f32).contiguous(), state_dict["conv_in.bias"].to(**f32).contiguous())
        self.conv_out = (state_dict["conv_out.weight"].to(**f32).contiguous(), state_dict["conv_out.bias"].to(**f32).contiguous())


# ----------------------------------------------------------------------------------------------
# LoRA state
# ----------------------------------------------------------------------------------------------
class LoraModule:
    __slots__ = ("path", "kind", "N", "K", "C", "r", "A", "B", "gA", "gB", "A_fwd", "A_bwd", "Bs_fwd", "Bs_bwd", "Bs_geglu")


class HalfSaved:
    """A tensor saved for the backward that already holds ONLY the first (grad-requiring) half of the batch rows
    (``UNet.forward(save=True, save_half=True)``): ``tape_first_half`` takes it as is instead of slicing it."""
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t


# debug hook: PCM_FUSE_GEGLU=0 keeps the feed-forward's GEGLU as a separate pass in the grad-requiring forward (A/B measurement)
FUSE_GEGLU_GRAD = os.environ.get("PCM_FUSE_GEGLU", "1") != "0"
# debug hook: PCM_TEXT_KV=0 runs the frozen pass's cross-attention K / V projections layer by layer (A/B measurement)
FUSE_TEXT_KV = os.environ.get("PCM_TEXT_KV", "1") != "0"
FUSE_TEXT_KV_LORA = os.environ.get("PCM_TEXT_KV_LORA", "1") != "0"   # LoRA pass: pass-wide rank-64 down-projection of the text + one K|V GEMM per block
FUSE_TEMB = os.environ.get("PCM_TEMB_BATCH", "1") != "0"      # every resnet's time_emb_proj of a pass as one GEMM (UNet._temb_all); 0: one GEMM per resnet
# debug hook: PCM_LORA_QKV=0 runs the self-attention q/k/v LoRA projections as three separate layers (A/B measurement)
FUSE_LORA_QKV = os.environ.get("PCM_LORA_QKV", "1") != "0"


# round 6 (include/pcm_hip.h abi 5).  PCM_GN_FUSE=1: GroupNorm statistics from the producing contraction's epilogue instead of their own pass
# over the tensor; PCM_CAT_FUSE=0: torch.cat([h, skip], dim=1) of the up blocks as a copy kernel instead of the producers writing
# straight into the concatenated buffer (A/B measurement hooks)
FUSE_GN_STATS = os.environ.get("PCM_GN_FUSE", "0") == "1"      # opt-in: built and measured SLOWER on MI355X (fp64 atomic traffic + the column pass)
FUSE_CONCAT = os.environ.get("PCM_CAT_FUSE", "1") != "0"
FUSE_GN_BWD_ADD = os.environ.get("PCM_GN_BWD_ADD", "1") != "0"      # the skip-path gradient of a resnet / transformer joins in the GroupNorm backward's apply pass


def _cs_of(t):
    """the per-channel statistics a producing contraction attached to its output (ops.ChStats, or a pair for a concatenation); None: none"""
    return getattr(t, "_pcm_cs", None)


def _view(t, *shape):
    """t.view(*shape) that keeps the attached statistics (a view is a new tensor object)"""
    v = t.view(*shape)
    cs = _cs_of(t)
    if cs is not None:
        v._pcm_cs = cs
    return v


class Skip:
    """one entry of the down path's skip stack.  ``cb``: the [B, HW, Ch + Cs] buffer the up path will read as torch.cat([h, skip], dim=1) whose
    right-hand channels were already written by this skip's producer (pcm_gemm_epi.out2); None: concatenate with a copy"""
    __slots__ = ("t", "H", "W", "cb")

    def __init__(self, t, H, W, cb=None):
        self.t, self.H, self.W, self.cb = t, H, W, cb


class LoraTextKV:
    """concatenated operands of the cross-attention to_k / to_v LoRA factors (LoraState.text_kv): A_cat [2 * nb * r][K] (block i: rows of
    to_k then to_v), Bs_kv[block prefix] = block-diagonal [2C][2r], index[block prefix] = i"""
    __slots__ = ("paths", "K", "A_cat", "Bs_kv", "index")


class LoraTemb:
    """concatenated operands of every resnet's time_emb_proj LoRA (LoraState.temb): paths in row order, off[path] = (first row, N, index)"""
    __slots__ = ("paths", "K", "N", "off", "A_cat", "Bs_cat")


class LoraQKV:
    """concatenated operands of one self-attention's to_q/to_k/to_v LoRA factors (see LoraState.__init__)."""
    __slots__ = ("K", "N", "r3", "q", "k", "v", "A_cat_fwd", "A_cat_bwd", "Bs_cat_fwd", "Bs_cat_bwd")


class LoraState:
    """peft-0.9 LoRA factors for the reference's 14 target patterns in ONE flat fp32 buffer
    (+ grads + Adam moments), plus the bf16 operand copies the kernels read.
    scaling = lora_alpha / r (peft default lora_alpha = 8; LoraConfig at :866 does not set it)."""

    def __init__(self, cfg, rank=64, lora_alpha=8.0, device="cuda", seed=1, b_std=0.0, targets=None, init="kaiming"):
        """``targets``: [(module path, weight shape)] (default: the UNet's 14 patterns for ``cfg``).  ``rank`` <= 64: the HIP LoRA
        kernels are rank-64; a smaller rank (SD3 recipe: 32) is stored ZERO-PADDED to 64 -- the padded rows of A / columns of B get
        exactly zero gradients (t_pad = x A_pad^T = 0, u_pad = dy B_pad = 0) and AdamW leaves exact zeros at zero, so the padded
        factors ARE the rank-``rank`` factors; checkpoints carry the real rank.  ``init``: "kaiming" (peft default, SD1.5/SDXL
        recipes) or "gaussian" (init_lora_weights="gaussian", train_pcm_lora_sd3.py:977: A ~ N(0, 1/r))."""
        if not (0 < rank <= 64):
            raise ValueError("pcm_amd: the HIP LoRA kernels are built for rank <= 64 (reference recipes: 64, SD3 32)")
        self.real_rank = rank
        rank = 64
        self.cfg, self.rank, self.alpha, self.scaling = cfg, rank, lora_alpha, lora_alpha / self.real_rank
        self.device = torch.device(device)
        targets = targets if targets is not None else lora_target_modules(cfg)
        # UNet attention query projections: the packed s*B copies (and the dB gradient factor, layer_bwd) carry the query scale the base
        # weights carry (UNetWeights.q_scale); the fp32 master factors -- what checkpoints hold -- are untouched
        qsf = getattr(cfg, "q_scale_of_path", None)
        self.q_scale = {path: qsf(path, shp[0]) for path, shp in targets if qsf is not None and qsf(path, shp[0]) is not None}
        total = 0
        layout = []
        for path, shp in targets:
            ain = math.prod(shp[1:])
            layout.append((path, shp, total, total + rank * ain))
            total += rank * ain + shp[0] * rank
        self.numel = total
        # flat-buffer offset of the first module of the mid / up blocks: the backward finishes [late_offset:] (up blocks, then mid) before it
        # enters the down blocks, so that part of the gradient buffer can be all-reduced while the down blocks still back-propagate
        late = [oa for path, _, oa, _ in layout if path.startswith(("mid_block", "up_blocks"))]
        self.late_offset = min(late) if late and min(late) > 0 else None
        self.params = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros_like(self.params)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.gradsq = torch.zeros(1, dtype=torch.float64, device=self.device)
        self.modules = OrderedDict()
        g = torch.Generator().manual_seed(seed)
        for path, shp, oa, ob in layout:
            m = LoraModule()
            m.path, m.N, m.r = path, shp[0], rank
            ain = math.prod(shp[1:])
            if len(shp) == 4 and shp[-1] == 3:
                m.kind, m.C, m.K = "conv3", shp[1], 9 * shp[1]
                a_shape, b_shape = (rank, 3, 3, shp[1]), (shp[0], rank, 1, 1)   # INTERNAL layout [r][kh][kw][ci]
            elif len(shp) == 4:     # 1x1 conv, or a k x k / stride k patch conv run as a GEMM over flattened (c, kh, kw) patches
                m.kind, m.C, m.K = "lin", 0, ain
                a_shape, b_shape = (rank,) + tuple(shp[1:]), (shp[0], rank, 1, 1)
            else:
                m.kind, m.C, m.K = "lin", 0, shp[1]
                a_shape, b_shape = (rank, shp[1]), (shp[0], rank)
            m.A = self.params[oa:oa + rank * ain].view(a_shape)
            m.B = self.params[ob:ob + shp[0] * rank].view(b_shape)
            m.gA = self.grads[oa:oa + rank * ain].view(a_shape)
            m.gB = self.grads[ob:ob + shp[0] * rank].view(b_shape)
            # peft init: kaiming_uniform_(A, a=sqrt(5)) == U(+-1/sqrt(fan_in)) or N(0, 1/r) ("gaussian"); B = 0
            bound = 1.0 / math.sqrt(ain)
            rr = self.real_rank

            def draw(shape):
                if init == "gaussian":
                    return torch.randn(shape, generator=g) / rr
                return (torch.rand(shape, generator=g) * 2 - 1) * bound
            if m.kind == "conv3":   # draw in peft shape [r, C, 3, 3] (same RNG stream as peft order), store permuted
                m.A[:rr].copy_(draw((rr, shp[1], 3, 3)).permute(0, 2, 3, 1).to(self.device))
            else:
                m.A[:rr].copy_(draw((rr,) + tuple(a_shape[1:])).to(self.device))
            if b_std > 0:
                m.B[:, :rr].copy_((torch.randn((b_shape[0], rr) + tuple(b_shape[2:]), generator=g) * b_std).to(self.device))
            self.modules[path] = m
        # bf16 MFMA operand copies of every factor in ONE flat buffer, refreshed by ONE segmented-pack launch
        import ctypes as C
        import numpy as np
        ototal, descs = 0, []
        r = rank

        def alloc(n):
            nonlocal ototal
            off = ototal
            ototal += (n + 7) // 8 * 8      # keep every operand 16-byte aligned
            return off
        layout2, geglu_ops = [], []
        for path, shp, oa, ob in layout:
            m = self.modules[path]
            o_af, o_ab, o_bf, o_bb = alloc(r * m.K), alloc(m.K * r), alloc(m.N * r), alloc(r * m.N)
            layout2.append((m, o_af, o_ab, o_bf, o_bb))
            if m.kind == "conv3":
                # A internal [r][tap][c] -> fwd operand [r][9C] (copy); dgrad operand [c][(8-tap)][r]: 9 strided transposes
                descs.append((oa, o_af, -1, r, m.K, m.K, m.K, 0, 1.0))
                for tap in range(9):
                    descs.append((oa + tap * m.C, -1, o_ab + (8 - tap) * r, r, m.C, m.K, 0, 9 * r, 1.0))
            else:
                descs.append((oa, o_af, o_ab, r, m.K, m.K, m.K, r, 1.0))               # A [r][K] -> copy + A^T [K][r]
            descs.append((ob, o_bf, o_bb, m.N, r, r, r, m.N, self.scaling * self.q_scale.get(path, 1.0)))   # s*B [N][r] -> copy + transpose [r][N]
            if path.endswith("ff.net.0.proj") and m.kind == "lin" and m.N % 16 == 0:
                # fused-GEGLU forward operand: rows of s*B interleaved [G values, G gates, ...] like Layer.w_geglu -- two strided copies
                # of (G rows x r) blocks: values block j -> rows 2Gj.., gates block j -> rows 2Gj+G..
                o_bg, inner, G = alloc(m.N * r), m.N // 2, GEGLU_GROUP
                geglu_ops.append((m, o_bg))
                descs.append((ob, o_bg, -1, inner // G, G * r, G * r, 2 * G * r, 0, self.scaling))
                descs.append((ob + inner * r, o_bg + G * r, -1, inner // G, G * r, G * r, 2 * G * r, 0, self.scaling))
        # self-attention q/k/v triples additionally get CONCATENATED operands, so the three rank-64 down-projections
        # are one [M,C]x[C,192] GEMM and the three up-projections ride the fused QKV GEMM as one block-diagonal K=192
        # segment (the off-diagonal blocks stay at the zeros this buffer is created with):
        #   A_cat_fwd [3r][C], A_cat_bwd [C][3r], Bs_cat_fwd [3N][3r] (block-diag), Bs_cat_bwd [3r][3N] (block-diag)
        offs = {path: (oa, ob) for path, shp, oa, ob in layout}
        qkv_layout = []
        for path in list(offs):
            if not (path.endswith("attn1.to_q") or path.endswith("attn.to_q")):     # UNet self-attention / MMDiT joint attention (image stream)
                continue
            p = path[:-len("to_q")]
            trio = [self.modules.get(p + n) for n in ("to_q", "to_k", "to_v")]
            if any(t is None or t.kind != "lin" or t.K != trio[0].K or t.N != trio[0].N for t in trio):
                continue
            Kc, Nc = trio[0].K, trio[0].N
            o_caf, o_cab, o_cbf, o_cbb = alloc(3 * r * Kc), alloc(Kc * 3 * r), alloc(3 * Nc * 3 * r), alloc(3 * r * 3 * Nc)
            qkv_layout.append((p, Kc, Nc, o_caf, o_cab, o_cbf, o_cbb))
            for j, t in enumerate(trio):
                oa_j, ob_j = offs[t.path]
                descs.append((oa_j, o_caf + j * r * Kc, o_cab + j * r, r, Kc, Kc, Kc, 3 * r, 1.0))
                descs.append((ob_j, o_cbf + j * Nc * 3 * r + j * r, o_cbb + j * r * 3 * Nc + j * Nc, Nc, r, r, 3 * r, 3 * Nc,
                              self.scaling * self.q_scale.get(t.path, 1.0)))
        # every resnet's time_emb_proj reads the SAME input (silu of the time embedding, [B, 1280]): their rank-64 down-projections are one
        # [B, K] x [K, n*r] GEMM and their up-projections ride ONE batched base GEMM as a block-diagonal K = n*r segment (UNet._temb_all;
        # off-diagonal blocks stay at the zeros this buffer is created with):  A_cat [n*r][K],  Bs_cat [sum N][n*r]
        temb_layout = None
        tp = [path for path, _, _, _ in layout if path.endswith("time_emb_proj") and self.modules[path].kind == "lin"]
        if len(tp) >= 2 and len({self.modules[t].K for t in tp}) == 1:
            nt, Kt, SNt = len(tp), self.modules[tp[0]].K, sum(self.modules[t].N for t in tp)
            o_tc, o_tb = alloc(nt * r * Kt), alloc(SNt * nt * r)
            roff, toff = 0, {}
            for j, t in enumerate(tp):
                oa_j, ob_j = offs[t]
                descs.append((oa_j, o_tc + j * r * Kt, -1, r, Kt, Kt, Kt, 0, 1.0))
                descs.append((ob_j, o_tb + roff * nt * r + j * r, -1, self.modules[t].N, r, r, nt * r, 0, self.scaling))
                toff[t] = (roff, self.modules[t].N, j)
                roff += self.modules[t].N
            temb_layout = (tp, Kt, SNt, o_tc, o_tb, toff)
        # cross-attention to_k / to_v of every transformer block read the SAME text: their rank-64 down-projections are ONE pass-wide
        # [M_text, K] x [K, 2 * nb * r] GEMM (A_kv_cat), and per block the K and V projections are one GEMM with a block-diagonal K = 2r second
        # segment (Bs_kv[block]: [2C][2r]) over the rows [W_k; W_v] of UNetWeights.kv_cat (UNet._text_kv_t / _attn_fwd)
        kv_layout = None
        kvp = [path[:-4] for path, _, _, _ in layout if path.endswith("attn2.to_k") and (path[:-4] + "to_v") in self.modules]
        if len(kvp) >= 1 and len({self.modules[b + "to_k"].K for b in kvp} | {self.modules[b + "to_v"].K for b in kvp}) == 1 and \
                all(self.modules[b + "to_k"].N == self.modules[b + "to_v"].N and self.modules[b + n].kind == "lin" for b in kvp for n in ("to_k", "to_v")):
            nb, Kx = len(kvp), self.modules[kvp[0] + "to_k"].K
            o_ka = alloc(2 * nb * r * Kx)
            blocks = []
            for i, b in enumerate(kvp):
                Cb = self.modules[b + "to_k"].N
                o_kb = alloc(2 * Cb * 2 * r)
                for jj, nme in enumerate(("to_k", "to_v")):
                    oa_j, ob_j = offs[b + nme]
                    descs.append((oa_j, o_ka + (2 * i + jj) * r * Kx, -1, r, Kx, Kx, Kx, 0, 1.0))
                    descs.append((ob_j, o_kb + jj * Cb * 2 * r + jj * r, -1, Cb, r, r, 2 * r, 0, self.scaling))
                blocks.append((b, Cb, o_kb))
            kv_layout = (kvp, Kx, o_ka, blocks)
        self.operands = torch.zeros(ototal, dtype=_act_dtype(), device=self.device)
        self.text_kv = None
        if kv_layout is not None:
            kvp, Kx, o_ka, blocks = kv_layout
            f = LoraTextKV()
            f.paths, f.K = kvp, Kx
            f.A_cat = self.operands[o_ka:o_ka + 2 * len(kvp) * r * Kx].view(2 * len(kvp) * r, Kx)
            f.Bs_kv = {b: self.operands[o:o + 4 * Cb * r].view(2 * Cb, 2 * r) for b, Cb, o in blocks}
            f.index = {b: i for i, b in enumerate(kvp)}
            self.text_kv = f
        self.temb = None
        if temb_layout is not None:
            tp, Kt, SNt, o_tc, o_tb, toff = temb_layout
            f = LoraTemb()
            f.paths, f.K, f.N, f.off = tp, Kt, SNt, toff
            f.A_cat = self.operands[o_tc:o_tc + len(tp) * r * Kt].view(len(tp) * r, Kt)
            f.Bs_cat = self.operands[o_tb:o_tb + SNt * len(tp) * r].view(SNt, len(tp) * r)
            self.temb = f
        self.qkv = {}
        for p, Kc, Nc, o_caf, o_cab, o_cbf, o_cbb in qkv_layout:
            f = LoraQKV()
            f.K, f.N, f.r3 = Kc, Nc, 3 * r
            f.q, f.k, f.v = (self.modules[p + n] for n in ("to_q", "to_k", "to_v"))
            f.A_cat_fwd = self.operands[o_caf:o_caf + 3 * r * Kc].view(3 * r, Kc)
            f.A_cat_bwd = self.operands[o_cab:o_cab + Kc * 3 * r].view(Kc, 3 * r)
            f.Bs_cat_fwd = self.operands[o_cbf:o_cbf + 9 * Nc * r].view(3 * Nc, 3 * r)
            f.Bs_cat_bwd = self.operands[o_cbb:o_cbb + 9 * Nc * r].view(3 * r, 3 * Nc)
            self.qkv[p] = f
        for m, o_af, o_ab, o_bf, o_bb in layout2:
            if m.kind == "conv3":
                m.A_fwd, m.A_bwd = self.operands[o_af:o_af + r * m.K].view(r, m.K), self.operands[o_ab:o_ab + m.K * r].view(m.C, 9 * r)
            else:
                m.A_fwd, m.A_bwd = self.operands[o_af:o_af + r * m.K].view(r, m.K), self.operands[o_ab:o_ab + m.K * r].view(m.K, r)
            m.Bs_fwd, m.Bs_bwd = self.operands[o_bf:o_bf + m.N * r].view(m.N, r), self.operands[o_bb:o_bb + r * m.N].view(r, m.N)
        for m, o_bg in geglu_ops:
            m.Bs_geglu = self.operands[o_bg:o_bg + m.N * r].view(m.N, r)
        arr = (capi.PackDesc * len(descs))()
        starts = np.zeros(len(descs) + 1, dtype=np.int32)
        for i, d in enumerate(descs):
            (arr[i].src_off, arr[i].dst_copy_off, arr[i].dst_t_off, arr[i].R, arr[i].Cc, arr[i].lds, arr[i].ldc, arr[i].ldt, arr[i].scale) = d
            starts[i + 1] = starts[i] + ((d[3] + 31) // 32) * ((d[4] + 31) // 32)
        self._ndesc, self._pack_blocks = len(descs), int(starts[-1])
        self._descs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        self._blk_start = torch.from_numpy(starts).to(self.device)
        self.repack()

    def repack(self):
        """fp32 master -> bf16 operand copies (after init / load / every optimizer step): one launch."""
        capi.lib().call("pcm_pack_segmented", ops.ptr(self.params), ops.ptr(self.operands), ops.ptr(self._descs),
                        ops.ptr(self._blk_start), self._ndesc, self._pack_blocks, capi.Lib.stream())

    def zero_grad(self):
        self.grads.zero_()

    # conv-A factors are stored [r, kh, kw, ci] internally (contiguous weight-gradient atomics; AdamW is
    # elementwise so the permutation is invisible to the optimizer); peft layout is [r, ci, kh, kw]
    @staticmethod
    def to_peft(m, t):
        return t.permute(0, 3, 1, 2).contiguous() if m.kind == "conv3" else t

    def A_peft(self, m):
        return self.to_peft(m, m.A)

    def gA_peft(self, m):
        return self.to_peft(m, m.gA)

    # ---- checkpoint formats (train_pcm_lora_sd15.py:52-72, :918-944, :1374-1382) ----
    def peft_state_dict(self):
        out = OrderedDict()
        rr = self.real_rank
        for p, m in self.modules.items():
            out[f"base_model.model.{p}.lora_A.weight"] = self.A_peft(m)[:rr].detach().clone()
            out[f"base_model.model.{p}.lora_B.weight"] = m.B[:, :rr].detach().clone()
        return out

    def load_peft_state_dict(self, sd):
        rr = self.real_rank
        for p, m in self.modules.items():
            a = sd[f"base_model.model.{p}.lora_A.weight"].to(self.device)
            m.A[:rr].copy_(a.permute(0, 2, 3, 1) if m.kind == "conv3" else a.view_as(m.A[:rr]))
            m.B[:, :rr].copy_(sd[f"base_model.model.{p}.lora_B.weight"].to(self.device).view_as(m.B[:, :rr]))
        self.repack()


# ----------------------------------------------------------------------------------------------
# one (optionally LoRA-wrapped) contraction layer: forward, dgrad, LoRA wgrad
# ----------------------------------------------------------------------------------------------
class Geo:
    """conv geometry: source tensor dims, stride, source mode, output dims."""
    __slots__ = ("Hs", "Ws", "stride", "src_mode", "Ho", "Wo")

    def __init__(self, Hs, Ws, stride=1, src_mode=capi.SRC_DIRECT):
        self.Hs, self.Ws, self.stride, self.src_mode = Hs, Ws, stride, src_mode
        Hv, Wv = (Hs * 2, Ws * 2) if src_mode != capi.SRC_DIRECT else (Hs, Ws)
        self.Ho, self.Wo = (Hv - 1) // stride + 1, (Wv - 1) // stride + 1

    def conv(self):
        return dict(Hs=self.Hs, Ws=self.Ws, stride=self.stride, src_mode=self.src_mode)


def layer_fwd(W: UNetWeights, lora, path, x, M, geo=None, save=None, rowvec=None, rows_per_batch=0, residual=None,
              act=capi.ACT_NONE, out_dtype=None, out=None, out2=None, stats=None):
    """y = base(x) + s*B(A(x)) [+ bias + rowvec + residual].  x: [M, K] (lin) or NHWC source (conv3).
    ``save`` (dict) receives what the backward needs.  ``out``: write into this [M, N] view (row stride out.stride(0): the left-hand channels of
    a concatenated buffer); ``out2``: also write a second copy there (a skip's slot in its concatenated buffer); ``stats`` = (ops.ChStatArena,
    rows per sample): ask the epilogue for the per-channel statistics of the output -- attached to the returned tensor when the plan emits them."""
    L = W.layers[path]
    out_dtype = _act_dtype() if out_dtype is None else out_dtype
    lm = lora.modules.get(path) if lora is not None else None
    conv = geo.conv() if L.kind == "conv3" else None
    Ho, Wo = (geo.Ho, geo.Wo) if conv else (0, 0)
    segs = [Seg(x, L.w_fwd, conv=conv)]
    t = None
    if lm is not None:
        t = torch.empty(M, lm.r, dtype=_act_dtype(), device=x.device)
        ops.gemm([Seg(x, lm.A_fwd, conv=conv)], M, lm.r, t, Ho=Ho, Wo=Wo)
        segs.append(Seg(t, lm.Bs_fwd))
    y = out if out is not None else torch.empty(M, L.N, dtype=out_dtype, device=x.device)
    cs = None
    if stats is not None and stats[0] is not None and FUSE_GN_STATS and y.dtype == _act_dtype():
        cs = stats[0].take(M // stats[1], L.N, stats[1])
    if out2 is not None and (L.N == 64 or M <= 16):
        raise ValueError("layer_fwd: out2 is not served by the rank-64 / batch-row kernels")
    ops.gemm(segs, M, L.N, y, bias=L.bias, rowvec=rowvec, rows_per_batch=rows_per_batch, residual=residual, act=act, Ho=Ho, Wo=Wo,
             ldo=y.stride(0), ldr=(residual.stride(-2) if residual is not None else None), out2=out2, ldo2=(out2.stride(0) if out2 is not None else None),
             chstats=cs)
    if cs is not None and cs.rows:
        y._pcm_cs = cs
    if save is not None:
        save["x"], save["t"], save["M"], save["geo"] = x, t, M, geo
    return y


class WgradSide:
    """LoRA weight-gradient launches on a SIDE stream.  The backward's critical path is the input-gradient chain (dgrad GEMMs, attention,
    norms); the weight gradients hang off it as leaves (they only feed the flat gradient buffer) and are latency / atomics bound, so they
    run beside the chain: fork after the operands exist, join once at the end of the backward.  Works under hipGraph capture (the side
    stream is forked from the capturing stream, so its launches become parallel branches of the same graph).  Operand tensors are kept
    alive until the join: the caching allocator must not hand their memory to a later main-stream allocation while the side stream reads."""

    def __init__(self):
        self.stream, self.keep = torch.cuda.Stream(), []

    def run(self, fn, *tensors):
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            fn()
        self.keep.extend(tensors)

    def join(self):
        torch.cuda.current_stream().wait_stream(self.stream)
        self.keep.clear()


_SIDE = None      # set by UNet.backward for the duration of one backward pass (None: weight gradients inline, e.g. host emulation)
# opt-in (PCM_WGRAD_SIDE=1): measured neutral on one MI355X at bs 16 (126.1 ms/step inline vs 127.0 with the side stream: the dgrad GEMM
# blocks own a CU's whole LDS, so the weight-gradient blocks only find room between launches, and a multi-branch hipGraph enqueues slower)
WGRAD_SIDE_STREAM = os.environ.get("PCM_WGRAD_SIDE", "0") == "1"


# PCM_WGRAD_DEFER=n (round 6): the weight-gradient jobs of up to n LoRA modules of a backward are collected and go out as ONE
# pcm_lora_wgrad_multi_bf16 call (<= 64 jobs per launch) instead of one call per module -- they are leaves of the backward (they only feed the flat
# gradient buffer) and the backward's tensors are never modified in place, so issuing them later changes nothing but the launch count (196 -> ~30
# multi launches per SD1.5 step).  Measured on one MI355X, interleaved (profiles/r06_r_*): C2 105.0 -> 103.8 ms per step at n = 32 (16: 103.9, 64: 104.0,
# one flush per backward: 104.2), C4 206.3 -> 204.4, C3 105.7 -> 105.3.  0 / 1: one call per module.  Under set_deterministic the jobs of a call run one by
# one with their ordered finalize, so the gradients are bitwise the same either way (tests/test_emu_unet.py, tests/test_gpu_step.py).
WGRAD_DEFER = int(os.environ.get("PCM_WGRAD_DEFER", "32"))
_DEFER = None     # [(fn, operand tensors)] of the backward in progress


def _flush_deferred():
    if _DEFER:
        jobs = list(_DEFER)
        del _DEFER[:]
        with ops.wgrad_batch():
            for fn, _ in jobs:
                fn()


def _wgrad(fn, *tensors):
    if _SIDE is not None:
        _SIDE.run(fn, *tensors)
    elif _DEFER is not None:
        _DEFER.append((fn, tensors))      # (the operands stay referenced until the flush)
        if len(_DEFER) >= WGRAD_DEFER:
            _flush_deferred()
    else:
        fn()


def layer_bwd(W: UNetWeights, lora, path, dy, saved, need_dx=True, residual=None):
    """dy [M, N] -> dx ([M_in, K] or NHWC of the source); accumulates LoRA grads.  ``residual`` is
    added to dx in the GEMM epilogue."""
    L = W.layers[path]
    lm = lora.modules.get(path) if lora is not None else None
    x, t, M, geo = saved["x"], saved["t"], saved["M"], saved["geo"]
    u = None
    ldy = dy.stride(0) if (L.kind != "conv3" and dy.dim() == 2) else None     # (a column slice of a wider matrix: the fused K|V gradient)
    if lm is not None:
        u = torch.empty(M, lm.r, dtype=_act_dtype(), device=dy.device)
        ops.gemm([Seg(dy, lm.Bs_bwd, lda=ldy)], M, lm.r, u)              # u = dy (sB)   [M, r]

        def wg():
            with ops.wgrad_batch():       # dB and dA share one launch where the kernels allow it
                ops.lora_wgrad(dy, t, lm.gB, lora.scaling * lora.q_scale.get(path, 1.0), M, G=L.N, g_stride=lm.r, r_stride=1, ldb=ldy)   # dB = s dy^T t
                if L.kind == "conv3":
                    ops.lora_wgrad(x, u, lm.gA, 1.0, M, conv=dict(Hs=geo.Hs, Ws=geo.Ws, Ho=geo.Ho, Wo=geo.Wo, stride=geo.stride,
                                                                  src_mode=geo.src_mode), g_stride=1, r_stride=L.K)
                else:
                    ops.lora_wgrad(x, u, lm.gA, 1.0, M, G=L.K, g_stride=1, r_stride=L.K)          # dA = u^T x
        _wgrad(wg, dy, t, x, u)
    if not need_dx:
        return None
    if L.kind == "conv3":
        # input gradient = 3x3 conv of dy with tap-flipped, in/out-transposed weights, evaluated on the
        # (virtual) input grid; a stride-2 forward reads dy zero-inserted
        if geo.src_mode == capi.SRC_UPSAMPLE2:
            Hin, Win = 2 * geo.Hs, 2 * geo.Ws  # gradient wrt the upsampled image (pooled by the caller)
        else:
            Hin, Win = geo.Hs, geo.Ws
        B = M // (geo.Ho * geo.Wo)
        Min = B * Hin * Win
        dconv = dict(Hs=geo.Ho, Ws=geo.Wo, stride=1, src_mode=capi.SRC_ZEROINS2 if geo.stride == 2 else capi.SRC_DIRECT)
        segs = [Seg(dy, L.w_bwd, conv=dconv)]
        if lm is not None:
            segs.append(Seg(u, lm.A_bwd, conv=dconv))
        dx = torch.empty(Min, L.C, dtype=_act_dtype(), device=dy.device)
        ops.gemm(segs, Min, L.C, dx, residual=residual, Ho=Hin, Wo=Win)
        return dx
    segs = [Seg(dy, L.w_bwd, lda=ldy)]
    if lm is not None:
        segs.append(Seg(u, lm.A_bwd))
    dx = torch.empty(M, L.K, dtype=_act_dtype(), device=dy.device)
    ops.gemm(segs, M, L.K, dx, residual=residual)
    return dx


# ----------------------------------------------------------------------------------------------
# UNet blocks
# ----------------------------------------------------------------------------------------------
class UNet:
    """Runner bound to frozen weights and (optionally) LoRA factors.

    forward(sample[B,4,H,W] fp32 NCHW, timesteps[B] int64, encoder_hidden_states[B,77,768]) ->
    eps [B,4,H,W] fp32 (``UNet2DConditionModel.forward(...).sample``).  With ``save=True`` the call
    returns a tape for ``backward(d_eps)`` which accumulates LoRA grads into ``lora.grads``.
    """

    def __init__(self, weights: UNetWeights, lora: LoraState = None):
        self.W, self.lora, self.cfg = weights, lora, weights.cfg
        self._arena = None      # per-pass arena of pre-zeroed GroupNorm statistics (ops.StatArena)
        self._kv_all = None     # frozen pass: every cross-attention's K / V projection of the text, one GEMM (UNetWeights.kv_cat)
        self._text_t = None     # LoRA pass: rank-64 down-projections of the text for every cross-attention's to_k / to_v, one GEMM (_text_kv_t)
        self._side = None       # WgradSide of this runner's backward passes
        self._save_half = False
        self._temb = None       # this pass's time_emb_proj outputs of every resnet, one batched GEMM (_temb_all); None: per-resnet GEMMs
        self._temb_plans = {}   # (batch, rank) -> device descriptor tables of the two scatter launches
        self._cs_arena = None   # per-pass arena of pre-zeroed per-channel statistics the contraction epilogues fill (ops.ChStatArena)
        # channels whose per-channel statistics a pass may request: every resnet conv1 / conv2, transformer proj_out, down / up-sampler conv
        self._cs_channels = sum(L.N for path, L in weights.layers.items()
                                if path.endswith(("conv1", "conv2", "proj_out", "downsamplers.0.conv", "upsamplers.0.conv")))
        # the LoRA state's block-diagonal operand is laid out in ITS module order: batch only when that is the weights' order (same rows)
        self._temb_orders_agree = (lora is not None and getattr(lora, "temb", None) is not None and list(weights.temb_off) == list(lora.temb.paths)
                                   and all(weights.temb_off[t] == lora.temb.off[t][:2] for t in lora.temb.paths) and lora.temb.K == weights.temb_cat.shape[1])

    # ---- norm helpers ----
    def _gn(self, path, x, act, eps, save):
        g, b = self.W.norms[path]
        cs = _cs_of(x) if FUSE_GN_STATS else None
        cs, cs2 = cs if isinstance(cs, tuple) else (cs, None)
        if cs is not None and (cs.rows != x.shape[1] or cs.B != x.shape[0] or cs.C + (cs2.C if cs2 is not None else 0) != x.shape[2]):
            cs = cs2 = None          # (not this tensor's geometry: e.g. a duplicated half batch)
        y, stats = ops.groupnorm_fwd(x, g, b, self.cfg.norm_num_groups, eps, act, arena=self._arena, chstats=cs, chstats2=cs2)
        if save is not None:
            save["gn_x"], save["gn_stats"] = x, stats
        return y

    def _gn_bwd(self, path, dy, act, eps, saved, dres=None):
        g, b = self.W.norms[path]
        if dres is not None and not FUSE_GN_BWD_ADD:
            return ops.add(ops.groupnorm_bwd(saved["gn_x"], dy, saved["gn_stats"], g, b, self.cfg.norm_num_groups, eps, act, arena=self._arena), dres.view_as(saved["gn_x"]))
        return ops.groupnorm_bwd(saved["gn_x"], dy, saved["gn_stats"], g, b, self.cfg.norm_num_groups, eps, act, arena=self._arena,
                                 dres=None if dres is None else dres.view_as(saved["gn_x"]))

    @staticmethod
    def _scatter_table(descs, dev):
        """device descriptor table of one pcm_pack_segmented launch that rounds an fp32 matrix to the 16-bit format and scatters column
        ranges into contiguous blocks: descs = [(src_off, dst_off, -1, R, Cc, lds, ldc, 0, 1.0)]"""
        import numpy as np
        arr = (capi.PackDesc * len(descs))()
        starts = np.zeros(len(descs) + 1, dtype=np.int32)
        for i, d in enumerate(descs):
            (arr[i].src_off, arr[i].dst_copy_off, arr[i].dst_t_off, arr[i].R, arr[i].Cc, arr[i].lds, arr[i].ldc, arr[i].ldt, arr[i].scale) = d
            starts[i + 1] = starts[i] + ((d[3] + 31) // 32) * ((d[4] + 31) // 32)
        return (torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev), torch.from_numpy(starts).to(dev), len(descs), int(starts[-1]))

    # ---- LoRA pass: rank-64 down-projections of the text for every cross-attention's to_k / to_v ----
    def _text_kv_t(self, text2d):
        """{block prefix: (t_kv [Mt, 2r], t_k [Mt, r], t_v [Mt, r])}, all contiguous: ONE GEMM [Mt, K] x [K, 2 * nb * r] in fp32 + one scatter launch
        (the same single rounding of the fp32 accumulator as the per-module down-projection); t_kv feeds the block's fused K|V GEMM, t_k / t_v
        are what the backward's weight-gradient jobs read"""
        lora = self.lora
        f, r, Mt, dev = lora.text_kv, lora.rank, text2d.shape[0], text2d.device
        nb = len(f.paths)
        plan = self._temb_plans.get(("kv", Mt))
        if plan is None:
            W2 = 2 * nb * r
            descs = [(2 * i * r, i * Mt * 2 * r, -1, Mt, 2 * r, W2, 2 * r, 0, 1.0) for i in range(nb)]
            descs += [(j * r, nb * Mt * 2 * r + j * Mt * r, -1, Mt, r, W2, r, 0, 1.0) for j in range(2 * nb)]
            plan = self._temb_plans[("kv", Mt)] = self._scatter_table(descs, dev)
        t32 = torch.empty(Mt, 2 * nb * r, dtype=torch.float32, device=dev)
        ops.gemm([Seg(text2d, f.A_cat)], Mt, 2 * nb * r, t32)
        tb = torch.empty(2 * nb * Mt * 2 * r, dtype=_act_dtype(), device=dev)
        dsc, st, nd, nblk = plan
        capi.lib().call("pcm_pack_segmented", ops.ptr(t32), ops.ptr(tb), ops.ptr(dsc), ops.ptr(st), nd, nblk, capi.Lib.stream())
        res, base = {}, nb * Mt * 2 * r
        for i, b in enumerate(f.paths):
            res[b] = (tb[i * Mt * 2 * r:(i + 1) * Mt * 2 * r].view(Mt, 2 * r),
                      tb[base + 2 * i * Mt * r:base + (2 * i + 1) * Mt * r].view(Mt, r), tb[base + (2 * i + 1) * Mt * r:base + (2 * i + 2) * Mt * r].view(Mt, r))
        return res

    # ---- every resnet's time_emb_proj of one pass ----
    def _temb_all(self, emb_act, B):
        """{path: (temb [B, N] bf16 contiguous, t [B, r] bf16 contiguous or None)} for every resnet of the pass from ONE base GEMM (+ ONE rank-64
        down-projection GEMM with LoRA): all of them read the same ``emb_act`` [B, K].  The GEMMs write fp32 ([B, sum N] / [B, n*r]); one
        segmented-pack launch each rounds to the 16-bit format -- the same single rounding of the fp32 accumulator the per-resnet GEMM
        epilogue does -- and scatters the columns into per-resnet CONTIGUOUS blocks (conv1's epilogue reads its row vector with row stride
        N; the backward reads t with row stride r).  22 + 22 + 22 launches of 8-16 us -> 2 + 4 per step at SD1.5 size."""
        W, lora = self.W, self.lora
        dev = emb_act.device
        paths, SN = list(W.temb_off), W.temb_cat.shape[0]
        nt = len(paths)
        r = lora.rank if lora is not None else 0
        key = (B, r)
        plan = self._temb_plans.get(key)
        if plan is None:
            table = lambda descs: self._scatter_table(descs, dev)   # noqa: E731
            # out32 [B][SN] -> blocks [B][N_j] at element offset B * off_j (every N_j is a multiple of 8: 16-byte aligned blocks)
            d_out = [(off, B * off, -1, B, N, SN, N, 0, 1.0) for off, N in (W.temb_off[t] for t in paths)]
            d_t = None
            if r:
                # t32 [B][nt*r] -> the whole matrix (the batched GEMM's second segment) at 0, then blocks [B][r] behind it
                d_t = [(0, 0, -1, B, nt * r, nt * r, nt * r, 0, 1.0)] + [(j * r, B * nt * r + j * B * r, -1, B, r, nt * r, r, 0, 1.0) for j in range(nt)]
            plan = self._temb_plans[key] = (table(d_out), table(d_t) if d_t else None)
        t_all, t_blocks, segs = None, None, [Seg(emb_act, W.temb_cat)]
        if r:
            t32 = torch.empty(B, nt * r, dtype=torch.float32, device=dev)
            ops.gemm([Seg(emb_act, lora.temb.A_cat)], B, nt * r, t32)
            tb = torch.empty(2 * B * nt * r, dtype=_act_dtype(), device=dev)
            dsc, st, nd, nb = plan[1]
            capi.lib().call("pcm_pack_segmented", ops.ptr(t32), ops.ptr(tb), ops.ptr(dsc), ops.ptr(st), nd, nb, capi.Lib.stream())
            t_all, t_blocks = tb[:B * nt * r].view(B, nt * r), tb[B * nt * r:]
            segs.append(Seg(t_all, lora.temb.Bs_cat, k_algo=r))       # (block-diagonal: r of the nt * r columns are non-zero per output row)
        out32 = torch.empty(B, SN, dtype=torch.float32, device=dev)
        ops.gemm(segs, B, SN, out32, bias=W.temb_bias)
        ob = torch.empty(B * SN, dtype=_act_dtype(), device=dev)
        dsc, st, nd, nb = plan[0]
        capi.lib().call("pcm_pack_segmented", ops.ptr(out32), ops.ptr(ob), ops.ptr(dsc), ops.ptr(st), nd, nb, capi.Lib.stream())
        res = {}
        for j, t in enumerate(paths):
            off, N = W.temb_off[t]
            res[t] = (ob[B * off:B * (off + N)].view(B, N), t_blocks[j * B * r:(j + 1) * B * r].view(B, r) if r else None)
        return res

    # ---- resnet ----
    def resnet_fwd(self, p, x, emb_act, B, H, Wd, tape, out=None, out2=None, want_stats=True):
        """``out`` / ``out2`` / ``want_stats``: where the block's output goes (layer_fwd) and whether a GroupNorm reads it next"""
        W, lora = self.W, self.lora
        M = B * H * Wd
        Cout = W.layers[p + "conv1"].N
        sv = {} if tape is not None else None
        s1 = {} if sv is not None else None
        n1 = self._gn(p + "norm1", x, capi.ACT_SILU, self.cfg.norm_eps, s1)
        st_ = {} if sv is not None else None
        if self._temb is not None:
            # (one batched GEMM per pass: _temb_all; a pass that runs this resnet on the first rows only -- dup_halves -- takes the prefix)
            temb, t_r = self._temb[p + "time_emb_proj"]
            temb, t_r = temb[:B], (t_r[:B] if t_r is not None else None)
            if st_ is not None:
                st_["x"], st_["t"], st_["M"], st_["geo"] = emb_act, t_r, B, None
        else:
            temb = layer_fwd(W, lora, p + "time_emb_proj", emb_act, B, save=st_)            # [B, Cout]
        geo = Geo(H, Wd)
        c1 = {} if sv is not None else None
        st = (self._cs_arena, H * Wd) if (H * Wd) % 64 == 0 else None
        h = layer_fwd(W, lora, p + "conv1", n1, M, geo, save=c1, rowvec=temb, rows_per_batch=H * Wd, stats=st)
        s2 = {} if sv is not None else None
        n2 = self._gn(p + "norm2", _view(h, B, H * Wd, Cout), capi.ACT_SILU, self.cfg.norm_eps, s2)
        sc = None
        if (p + "conv_shortcut") in W.layers:
            sc = {} if sv is not None else None
            res = layer_fwd(W, lora, p + "conv_shortcut", x.view(M, -1), M, save=sc)
        else:
            res = x.view(M, -1)
        c2 = {} if sv is not None else None
        out = layer_fwd(W, lora, p + "conv2", n2, M, geo, save=c2, residual=res, out=out, out2=out2, stats=st if want_stats else None)
        if tape is not None:
            tape.append(("resnet", p, dict(s1=s1, st=st_, c1=c1, s2=s2, sc=sc, c2=c2, B=B, H=H, W=Wd, Cout=Cout)))
        return _view(out, B, H * Wd, Cout)

    def resnet_bwd(self, p, d_out, sv, need_dx=True):
        W, lora = self.W, self.lora
        B, H, Wd, Cout = sv["B"], sv["H"], sv["W"], sv["Cout"]
        M = B * H * Wd
        d_out = d_out.view(M, Cout)
        d_n2 = layer_bwd(W, lora, p + "conv2", d_out, sv["c2"])
        d_h = self._gn_bwd(p + "norm2", d_n2.view(B, H * Wd, Cout), capi.ACT_SILU, self.cfg.norm_eps, sv["s2"])
        d_temb = ops.cast_bf16(ops.colsum(d_h))                                               # [B, Cout]
        layer_bwd(W, lora, p + "time_emb_proj", d_temb, sv["st"], need_dx=False)
        d_n1 = layer_bwd(W, lora, p + "conv1", d_h.view(M, Cout), sv["c1"], need_dx=need_dx)
        if not need_dx:
            if sv["sc"] is not None:
                layer_bwd(W, lora, p + "conv_shortcut", d_out, sv["sc"], need_dx=False)
            return None
        Cin = d_n1.shape[-1]
        if sv["sc"] is not None:
            d_x = self._gn_bwd(p + "norm1", d_n1.view(B, H * Wd, Cin), capi.ACT_SILU, self.cfg.norm_eps, sv["s1"])
            d_x = layer_bwd(W, lora, p + "conv_shortcut", d_out, sv["sc"], residual=d_x.view(M, Cin))
        else:       # identity skip: its gradient joins in the GroupNorm backward's apply pass (no add kernel)
            d_x = self._gn_bwd(p + "norm1", d_n1.view(B, H * Wd, Cin), capi.ACT_SILU, self.cfg.norm_eps, sv["s1"], dres=d_out.contiguous())
        return d_x.view(B, H * Wd, Cin)

    # ---- transformer (Transformer2DModel with one BasicTransformerBlock) ----
    def _attn_fwd(self, p, xn, ctx, B, L, Lk, C, resid, sv, Hh):
        W, lora = self.W, self.lora
        d = C // Hh
        M = B * L
        sq, sk, svv, so = ({} if sv is not None else None for _ in range(4))
        Mk = B * Lk
        if lora is None and sv is None and ctx is xn and p in W.qkv:
            qkv = torch.empty(M, 3 * C, dtype=_act_dtype(), device=xn.device)
            ops.gemm([Seg(xn, W.qkv[p])], M, 3 * C, qkv)
            qkv = qkv.view(B, L, 3 * C)
            o, lse = ops.attn_fwd(qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:], Hh, d, prescaled=True)
            return layer_fwd(W, lora, p + "to_out.0", o.view(M, C), M, residual=resid)
        fq = lora.qkv.get(p) if (lora is not None and FUSE_LORA_QKV and ctx is xn and p in W.qkv and (sv is None or p in W.qkv_bwd)) else None
        if fq is not None:
            # LoRA self-attention: the three rank-64 down-projections as one N=192 GEMM, then ONE QKV GEMM whose second
            # K-segment is the block-diagonal s*B operand; attention reads q/k/v in place (row stride 3C)
            t3 = torch.empty(M, fq.r3, dtype=_act_dtype(), device=xn.device)
            ops.gemm([Seg(xn, fq.A_cat_fwd)], M, fq.r3, t3)
            qkv = torch.empty(M, 3 * C, dtype=_act_dtype(), device=xn.device)
            ops.gemm([Seg(xn, W.qkv[p]), Seg(t3, fq.Bs_cat_fwd, k_algo=lora.rank)], M, 3 * C, qkv)
            q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
            o, lse = ops.attn_fwd(q.unflatten(0, (B, L)), k.unflatten(0, (B, L)), v.unflatten(0, (B, L)), Hh, d, prescaled=True)
            out = layer_fwd(W, lora, p + "to_out.0", o.view(M, C), M, save=so, residual=resid)
            if sv is not None:
                sv.update(fused=True, x=xn, t3=t3, so=so, q=q, k=k, v=v, o=o, lse=lse, L=L, Lk=Lk)
            return out
        q = layer_fwd(W, lora, p + "to_q", xn, M, save=sq)
        if lora is None and sv is None and self._kv_all is not None and p in W.kv_off:
            # frozen pass: this block's K / V are column slices of the pass-wide text projection (UNetWeights.kv_cat)
            off, Ck = W.kv_off[p]
            kv = self._kv_all.view(B, Lk, -1)
            o, lse = ops.attn_fwd(q.view(B, L, C), kv[:, :, off:off + Ck], kv[:, :, off + Ck:off + 2 * Ck], Hh, d, prescaled=True)
            return layer_fwd(W, lora, p + "to_out.0", o.view(M, C), M, residual=resid)
        if lora is not None and ctx is not xn and self._text_t is not None and p in self._text_t and W.kv_off[p][1] == C and ctx.numel() == Mk * W.kv_cat.shape[1] \
                and self._text_t[p][0].shape[0] == Mk:
            # LoRA cross-attention: K and V of this block as ONE GEMM over the rows [W_k; W_v] of the pass-wide text operand with a block-diagonal
            # K = 2r second segment; the rank-64 down-projections of the text were computed for every block at once (_text_kv_t).  Attention
            # reads k / v in place (row stride 2C); the backward's weight-gradient jobs read the contiguous t_k / t_v
            off, _ = W.kv_off[p]
            t_kv, t_k, t_v = self._text_t[p]
            ctx2 = ctx.view(Mk, -1)
            kv = torch.empty(Mk, 2 * C, dtype=_act_dtype(), device=xn.device)
            ops.gemm([Seg(ctx2, W.kv_cat[off:off + 2 * C]), Seg(t_kv, lora.text_kv.Bs_kv[p], k_algo=lora.rank)], Mk, 2 * C, kv)
            k, v = kv[:, :C], kv[:, C:]
            if sv is not None:
                sk.update(x=ctx2, t=t_k, M=Mk, geo=None)
                svv.update(x=ctx2, t=t_v, M=Mk, geo=None)
        else:
            k = layer_fwd(W, lora, p + "to_k", ctx, Mk, save=sk)
            v = layer_fwd(W, lora, p + "to_v", ctx, Mk, save=svv)
        o, lse = ops.attn_fwd(q.view(B, L, C), k.unflatten(0, (B, Lk)), v.unflatten(0, (B, Lk)), Hh, d, prescaled=True)
        out = layer_fwd(W, lora, p + "to_out.0", o.view(M, C), M, save=so, residual=resid)
        if sv is not None:
            sv.update(sq=sq, sk=sk, sv=svv, so=so, q=q, k=k, v=v, o=o, lse=lse, L=L, Lk=Lk)
        return out

    def _attn_bwd(self, p, d_out, sv, B, C, need_dctx, Hh):
        """returns (d_xn [M,C], d_ctx or None)"""
        W, lora = self.W, self.lora
        d, L, Lk = C // Hh, sv["L"], sv["Lk"]
        d_o = layer_bwd(W, lora, p + "to_out.0", d_out, sv["so"])
        if sv.get("fused"):
            fq, M, r = lora.qkv[p], B * L, lora.rank
            q, k, v, x, t3 = sv["q"], sv["k"], sv["v"], sv["x"], sv["t3"]
            d3 = torch.empty(M, 3 * C, dtype=_act_dtype(), device=d_o.device)        # [dq | dk | dv], written in place by attention
            dq, dk, dv = d3[:, :C], d3[:, C:2 * C], d3[:, 2 * C:]
            ops.attn_bwd(q.unflatten(0, (B, L)), k.unflatten(0, (B, L)), v.unflatten(0, (B, L)), sv["o"], d_o.view(B, L, C),
                         sv["lse"], Hh, d, out=(dq, dk, dv), prescaled=True)
            u3 = torch.empty(M, fq.r3, dtype=_act_dtype(), device=d_o.device)
            ops.gemm([Seg(d3, fq.Bs_cat_bwd, k_algo=C)], M, fq.r3, u3)                  # u_j = d_j (s B_j): block-diagonal operand
            def wg():
                with ops.wgrad_batch():       # six weight gradients, one launch
                    for j, (lm, dj) in enumerate(((fq.q, dq), (fq.k, dk), (fq.v, dv))):
                        ops.lora_wgrad(dj, t3[:, j * r:(j + 1) * r], lm.gB, lora.scaling * lora.q_scale.get(lm.path, 1.0), M, G=C, g_stride=r,
                                       r_stride=1, ldb=3 * C, lds=fq.r3)
                        ops.lora_wgrad(x, u3[:, j * r:(j + 1) * r], lm.gA, 1.0, M, G=fq.K, g_stride=1, r_stride=fq.K, lds=fq.r3)
            _wgrad(wg, d3, t3, x, u3)
            d_xn = torch.empty(M, fq.K, dtype=_act_dtype(), device=d_o.device)
            ops.gemm([Seg(d3, W.qkv_bwd[p]), Seg(u3, fq.A_cat_bwd)], M, fq.K, d_xn)
            return d_xn
        if sv["k"].stride(0) != C:       # fused K|V forward (row stride 2C): the gradients are written into one [Mk, 2C] matrix with the same strides
            dq = torch.empty(B * L, C, dtype=_act_dtype(), device=d_o.device)
            dkv = torch.empty(B * Lk, 2 * C, dtype=_act_dtype(), device=d_o.device)
            dk, dv = dkv[:, :C], dkv[:, C:]
            ops.attn_bwd(sv["q"].view(B, L, C), sv["k"].unflatten(0, (B, Lk)), sv["v"].unflatten(0, (B, Lk)), sv["o"], d_o.view(B, L, C),
                         sv["lse"], Hh, d, out=(dq, dk, dv), prescaled=True)
        else:
            dq, dk, dv = ops.attn_bwd(sv["q"].view(B, L, C), sv["k"].view(B, Lk, C), sv["v"].view(B, Lk, C), sv["o"],
                                      d_o.view(B, L, C), sv["lse"], Hh, d, prescaled=True)
            dk, dv = dk.view(B * Lk, C), dv.view(B * Lk, C)
        d_xn = layer_bwd(W, lora, p + "to_q", dq.view(B * L, C), sv["sq"])
        if need_dctx:  # self-attention: K/V inputs are xn too
            d_xn = layer_bwd(W, lora, p + "to_k", dk, sv["sk"], residual=d_xn)
            d_xn = layer_bwd(W, lora, p + "to_v", dv, sv["sv"], residual=d_xn)
        else:          # cross-attention: text embeddings need no gradient
            layer_bwd(W, lora, p + "to_k", dk, sv["sk"], need_dx=False)
            layer_bwd(W, lora, p + "to_v", dv, sv["sv"], need_dx=False)
        return d_xn

    def transformer_fwd(self, p, x, text, B, H, Wd, tape, depth=1, heads=None, dup_after_attn1=False, out=None, out2=None, want_stats=True):
        """Transformer2DModel: GroupNorm -> proj_in -> ``depth`` BasicTransformerBlocks -> proj_out + input residual.
        ``dup_after_attn1``: ``x`` holds B samples that stand for a batch [x; x] of 2B whose halves differ only in ``text`` (2B rows):
        everything up to and including the first self-attention is computed once and duplicated there; returns 2B samples."""
        W, lora = self.W, self.lora
        heads = heads if heads is not None else self.cfg.heads_at(0)
        C, L, M = x.shape[-1], H * Wd, B * H * Wd
        assert not dup_after_attn1 or (tape is None and text.shape[0] == 2 * B)
        Lt = text.shape[1]
        rec = tape is not None
        sgn, spi, spo = ({} if rec else None for _ in range(3))
        n = self._gn(p + "norm", x, capi.ACT_NONE, 1e-6, sgn)
        h = layer_fwd(W, lora, p + "proj_in", n.view(M, C), M, save=spi)
        blocks = []
        for k in range(depth):
            b = p + f"transformer_blocks.{k}."
            sa1, sa2, sf0, sf2 = ({} if rec else None for _ in range(4))
            g1, b1 = W.norms[b + "norm1"]
            n1, mu1, rs1 = ops.layernorm_fwd(h, g1, b1)
            h1 = self._attn_fwd(b + "attn1.", n1, n1, B, L, L, C, h, sa1, heads)
            if dup_after_attn1 and k == 0:      # the first cross-attention is where the two halves start to differ
                h1 = torch.cat([h1.view(M, C), h1.view(M, C)])
                x = torch.cat([x.view(M, C), x.view(M, C)]).view(2 * B, L, C)
                B, M = 2 * B, 2 * M
            g2, b2 = W.norms[b + "norm2"]
            n2, mu2, rs2 = ops.layernorm_fwd(h1, g2, b2)
            h2 = self._attn_fwd(b + "attn2.", n2, text.view(B * Lt, -1), B, L, Lt, C, h1, sa2, heads)
            g3, b3 = W.norms[b + "norm3"]
            n3, mu3, rs3 = ops.layernorm_fwd(h2, g3, b3)
            Lff = W.layers[b + "ff.net.0.proj"]
            lmff = lora.modules.get(b + "ff.net.0.proj") if lora is not None else None
            pre = None
            if Lff.w_geglu is not None and M >= 128 and (lmff is None or getattr(lmff, "Bs_geglu", None) is not None) and \
                    (FUSE_GEGLU_GRAD or (lora is None and not rec)):
                # GEGLU applied in the projection's epilogue: the 2*inner-wide pre-activation does not make the round trip through HBM
                # for a separate activation pass.  Frozen no-grad pass: it never reaches HBM at all.  Recording pass: the epilogue
                # also stores it (bf16, interleaved column order) for the rows whose backward will run -- with save_half only the
                # online half of the fused online + target batch -- and geglu_bwd_interleaved reads it.
                hg = None
                segs, t_ff = [Seg(n3, Lff.w_geglu)], None
                if lmff is not None:
                    t_ff = torch.empty(M, lmff.r, dtype=_act_dtype(), device=n3.device)
                    ops.gemm([Seg(n3, lmff.A_fwd)], M, lmff.r, t_ff)
                    segs.append(Seg(t_ff, lmff.Bs_geglu))
                if rec:
                    pre = torch.empty(M // 2 if self._save_half else M, Lff.N, dtype=_act_dtype(), device=n3.device)
                    sf0["x"], sf0["t"], sf0["M"], sf0["geo"] = n3, t_ff, M, None
                gg = torch.empty(M, Lff.N // 2, dtype=_act_dtype(), device=n3.device)
                ops.gemm(segs, M, Lff.N, gg, bias=Lff.bias_geglu, act=capi.ACT_GEGLU, ldo=Lff.N // 2, pre_out=pre)
                if pre is not None and self._save_half:
                    pre = HalfSaved(pre)
            else:
                hg = layer_fwd(W, lora, b + "ff.net.0.proj", n3, M, save=sf0)
                gg = ops.geglu_fwd(hg)
            h3 = layer_fwd(W, lora, b + "ff.net.2", gg, M, save=sf2, residual=h2)
            if rec:
                blocks.append(dict(sa1=sa1, sa2=sa2, sf0=sf0, sf2=sf2, h=h, mu1=mu1, rs1=rs1, h1=h1, mu2=mu2, rs2=rs2, h2=h2, mu3=mu3,
                                   rs3=rs3, hg=hg, pre=pre))
            h = h3
        st = (self._cs_arena, L) if (want_stats and L % 64 == 0) else None
        out = layer_fwd(W, lora, p + "proj_out", h, M, save=spo, residual=x.view(M, C), out=out, out2=out2, stats=st)
        if rec:
            tape.append(("transformer", p, dict(sgn=sgn, spi=spi, spo=spo, B=B, H=H, W=Wd, C=C, heads=heads,
                                                **{f"blk{k}": blocks[k] for k in range(depth)})))
        return _view(out, B, L, C)

    def transformer_bwd(self, p, d_out, sv):
        W, lora = self.W, self.lora
        B, H, Wd, C, heads = sv["B"], sv["H"], sv["W"], sv["C"], sv["heads"]
        M = B * H * Wd
        d_out = d_out.view(M, C)
        d_h = layer_bwd(W, lora, p + "proj_out", d_out, sv["spo"])                       # residual r: + d_out at the end
        depth = sum(1 for k in sv if k.startswith("blk"))
        for k in reversed(range(depth)):
            b, bs = p + f"transformer_blocks.{k}.", sv[f"blk{k}"]
            # ff: h3 = h2 + ff2(geglu(ff0(LN3(h2))))
            d_gg = layer_bwd(W, lora, b + "ff.net.2", d_h, bs["sf2"])
            d_hg = ops.geglu_bwd_interleaved(bs["pre"], d_gg) if bs.get("pre") is not None else ops.geglu_bwd(bs["hg"], d_gg)
            d_n3 = layer_bwd(W, lora, b + "ff.net.0.proj", d_hg, bs["sf0"])
            d_h2 = ops.layernorm_bwd(bs["h2"], d_n3, W.norms[b + "norm3"][0], bs["mu3"], bs["rs3"], dres=d_h)
            # attn2: h2 = h1 + attn2(LN2(h1), text)
            d_n2 = self._attn_bwd(b + "attn2.", d_h2, bs["sa2"], B, C, False, heads)
            d_h1 = ops.layernorm_bwd(bs["h1"], d_n2, W.norms[b + "norm2"][0], bs["mu2"], bs["rs2"], dres=d_h2)
            # attn1: h1 = h + attn1(LN1(h))
            d_n1 = self._attn_bwd(b + "attn1.", d_h1, bs["sa1"], B, C, True, heads)
            d_h = ops.layernorm_bwd(bs["h"], d_n1, W.norms[b + "norm1"][0], bs["mu1"], bs["rs1"], dres=d_h1)
        d_n = layer_bwd(W, lora, p + "proj_in", d_h, sv["spi"])
        return self._gn_bwd(p + "norm", d_n.view(B, H * Wd, C), capi.ACT_NONE, 1e-6, sv["sgn"], dres=d_out.contiguous()).view(B, H * Wd, C)

    # ---- whole network ----
    def forward(self, sample, timesteps, encoder_hidden_states, save=False, features=False, added_cond=None, save_half=False, dup_halves=False):
        """``features=True`` is the reference's ``modified_forward`` (discriminator_sd15.py:16-345): returns the 9
        hidden states after every down block, the mid block and every up block (no conv_norm_out / conv_out).
        ``save_half``: the caller will back-propagate through ``tape_first_half`` only (fused online + target batch), so tensors
        that exist only for the backward may be kept for the first half of the batch alone.
        ``dup_halves``: the caller guarantees that the two halves of the batch have IDENTICAL sample, timestep and added_cond rows and
        differ only in encoder_hidden_states (the teacher's cond / uncond pass, train_pcm_lora_sd15.py:1217-1252): conv_in, the first
        resnet and the first transformer block's self-attention are computed on one half and duplicated (result-identical)."""
        cfg, W, lora = self.cfg, self.W, self.lora
        dup_halves = bool(dup_halves and not save and not features and sample.shape[0] % 2 == 0 and cfg.down_attn[0]
                          and not cfg.addition_time_embed_dim)
        self._save_half = bool(save and save_half)
        B, _, H, Wd = sample.shape
        boc, n = cfg.block_out_channels, len(cfg.block_out_channels)
        tape = [] if save else None
        self._arena = ops.StatArena.for_pass(sample.device, W, B, cfg.norm_num_groups)
        # fp32 (what the CLIs pass) or already in THIS pass's 16-bit format; a 16-bit tensor of the other format (bfloat16 embeddings into a half
        # teacher pass under format_scope) goes through fp32 -- the cast kernel reads 4-byte elements
        text = encoder_hidden_states if encoder_hidden_states.dtype == _act_dtype() else ops.cast_bf16(encoder_hidden_states.float().contiguous())
        self._kv_all = None
        if lora is None and not save and W.kv_cat is not None and FUSE_TEXT_KV:
            Mt = text.shape[0] * text.shape[1]
            self._kv_all = torch.empty(Mt, W.kv_cat.shape[0], dtype=_act_dtype(), device=text.device)
            ops.gemm([Seg(text.view(Mt, -1), W.kv_cat)], Mt, W.kv_cat.shape[0], self._kv_all)
        t_emb = ops.timestep_embedding(timesteps, boc[0])
        e1 = layer_fwd(W, None, "time_embedding.linear_1", t_emb, B, act=capi.ACT_SILU)
        if cfg.addition_time_embed_dim:
            # SDXL "text_time" conditioning (added_cond_kwargs, train_pcm_lora_sdxl_adv.py:1113-1131): emb = time_emb +
            # add_embedding([pooled text embeds | sinusoid(6 time ids)]); everything upstream of silu(emb) is frozen
            assert added_cond is not None, "this UNet needs added_cond={'text_embeds': [B,P], 'time_ids': [B,6]}"
            emb_t = layer_fwd(W, None, "time_embedding.linear_2", e1, B)
            ids = added_cond["time_ids"].to(torch.int64).reshape(-1)
            tid = ops.timestep_embedding(ids, cfg.addition_time_embed_dim).view(B, -1)
            te = added_cond["text_embeds"]
            te = te if te.dtype == _act_dtype() else ops.cast_bf16(te.float().contiguous())
            add_in = torch.cat([te, tid], dim=1).contiguous()
            a1 = layer_fwd(W, None, "add_embedding.linear_1", add_in, B, act=capi.ACT_SILU)
            emb = layer_fwd(W, None, "add_embedding.linear_2", a1, B, residual=emb_t)
            emb_act = ops.silu(emb)
        else:
            emb_act = layer_fwd(W, None, "time_embedding.linear_2", e1, B, act=capi.ACT_SILU)  # silu(emb): only use of emb
        self._text_t = None
        if lora is not None and FUSE_TEXT_KV_LORA and W.kv_cat is not None and getattr(lora, "text_kv", None) is not None and \
                lora.text_kv.K == W.kv_cat.shape[1] and all(b in W.kv_off for b in lora.text_kv.paths):
            self._text_t = self._text_kv_t(text.view(text.shape[0] * text.shape[1], -1))
        self._temb = None
        if FUSE_TEMB and W.temb_cat is not None and (lora is None or (lora.temb is not None and self._temb_orders_agree)):
            self._temb = self._temb_all(emb_act, B)
        # ---- concat-free skips (round 6): the producer of a down-path skip tensor also writes it into the right-hand channels of the buffer the up
        # path reads as torch.cat([h, skip], dim=1) (pcm_gemm_epi.out2), and the up path's producers write h into the left-hand channels
        # (row stride = the buffer's): no concat pass.  cat_h[k] = channels of h when skip k is popped (from the config).
        lpb = cfg.layers_per_block
        n_skips = self._n_down_skips()
        cat_h, ch, k_ = {}, boc[-1], n_skips - 1
        rev = list(reversed(boc))
        for i in range(n):
            for j in range(lpb + 1):
                cat_h[k_] = ch
                ch = rev[i]
                k_ -= 1
        cat_on = FUSE_CONCAT and not features and capi.lib().dll.pcm_abi_version() >= 5
        self._cs_arena = ops.ChStatArena(sample.device, B * self._cs_channels * 16) if (FUSE_GN_STATS and not ops.DETERMINISTIC) else None

        def skip_slot(Cs, H_, W_):
            """(concat buffer, out2 view) for the skip about to be produced at index len(skips), or (None, None)"""
            k = len(skips)
            if not cat_on or Cs == 64 or B * H_ * W_ <= 16:
                return None, None
            cb = torch.empty(B, H_ * W_, cat_h[k] + Cs, dtype=_act_dtype(), device=sample.device)
            return cb, cb.view(B * H_ * W_, -1)[:, cat_h[k]:]

        if dup_halves:
            Bh = B // 2
            h = ops.conv_in_fwd(sample[:Bh].contiguous(), W.conv_in[0], W.conv_in[1], boc[0])
            skips = [Skip(torch.cat([h, h]), H, Wd)]
        else:
            skips = []
            cb, o2 = skip_slot(boc[0], H, Wd)
            if o2 is not None and (cat_h[0] + boc[0]) % 8:
                cb, o2 = None, None
            h = ops.conv_in_fwd(sample.contiguous(), W.conv_in[0], W.conv_in[1], boc[0], out2=o2)
            skips = [Skip(h, H, Wd, cb)]
        feats = []
        for i in range(n):
            for j in range(lpb):
                if dup_halves and i == 0 and j == 0:
                    h = self.resnet_fwd("down_blocks.0.resnets.0.", h, emb_act[:Bh].contiguous(), Bh, H, Wd, tape)
                    h = self.transformer_fwd("down_blocks.0.attentions.0.", h, text, Bh, H, Wd, tape, cfg.transformer_depth[0], cfg.heads_at(0),
                                             dup_after_attn1=True)
                    skips.append(Skip(h, H, Wd))
                    continue
                cb, o2 = skip_slot(boc[i], H, Wd)
                if cfg.down_attn[i]:
                    h = self.resnet_fwd(f"down_blocks.{i}.resnets.{j}.", h, emb_act, B, H, Wd, tape)
                    h = self.transformer_fwd(f"down_blocks.{i}.attentions.{j}.", h, text, B, H, Wd, tape, cfg.transformer_depth[i], cfg.heads_at(i), out2=o2)
                else:
                    h = self.resnet_fwd(f"down_blocks.{i}.resnets.{j}.", h, emb_act, B, H, Wd, tape, out2=o2)
                skips.append(Skip(h, H, Wd, cb))
            if i < n - 1:
                geo = Geo(H, Wd, stride=2)
                sv = {} if save else None
                cb, o2 = skip_slot(boc[i], geo.Ho, geo.Wo)
                Mo = B * geo.Ho * geo.Wo
                st = (self._cs_arena, geo.Ho * geo.Wo) if (geo.Ho * geo.Wo) % 64 == 0 else None
                h = layer_fwd(W, lora, f"down_blocks.{i}.downsamplers.0.conv", h, Mo, geo, save=sv, out2=o2, stats=st)
                H, Wd = geo.Ho, geo.Wo
                h = _view(h, B, H * Wd, -1)
                if save:
                    tape.append(("down", f"down_blocks.{i}.downsamplers.0.conv", dict(sv=sv, B=B, H=H, W=Wd)))
                skips.append(Skip(h, H, Wd, cb))
            if features:
                feats.append((h, H, Wd))
                if save:
                    tape.append(("feat", None, dict(k=len(feats) - 1)))

        def cat_target(Cout):
            """where the tensor that will be concatenated with the NEXT skip goes: the left-hand channels of that skip's buffer (or None)"""
            if not skips or skips[-1].cb is None:
                return None
            cb = skips[-1].cb
            if cb.shape[-1] - skips[-1].t.shape[-1] != Cout:
                return None
            return cb.view(-1, cb.shape[-1])[:, :Cout]

        h = self.resnet_fwd("mid_block.resnets.0.", h, emb_act, B, H, Wd, tape)
        h = self.transformer_fwd("mid_block.attentions.0.", h, text, B, H, Wd, tape, cfg.mid_depth, cfg.heads_at(n - 1))
        h = self.resnet_fwd("mid_block.resnets.1.", h, emb_act, B, H, Wd, tape, out=None if features else cat_target(boc[-1]))
        if features:
            feats.append((h, H, Wd))
            if save:
                tape.append(("feat", None, dict(k=len(feats) - 1)))
            if features == "down_mid":     # SDXL discriminator taps (discriminator_sdxl.py:311): stop after the mid block
                if save:
                    tape.append(("feats_end", None, dict(B=B, H=H, W=Wd)))
                    return feats, tape
                return feats
        for i in range(n):
            Cout = rev[i]
            for j in range(lpb + 1):
                sk = skips.pop()
                s = sk.t
                Ch = h.shape[-1]
                if sk.cb is not None and h.data_ptr() == sk.cb.data_ptr() and h.stride(-2) == sk.cb.shape[-1]:
                    x = sk.cb                       # both halves are already in place
                    ca, cb_ = _cs_of(h), _cs_of(s)
                    if ca is not None and cb_ is not None and not isinstance(ca, tuple) and not isinstance(cb_, tuple):
                        x._pcm_cs = (ca, cb_)       # statistics of the concatenation = the two producers', side by side
                else:
                    x = ops.concat_channels(h if h.is_contiguous() else h.contiguous(), s)
                h = x
                if save:
                    tape.append(("cat", None, dict(Ch=Ch, skip_index=len(skips))))
                last = j == lpb
                to_up = last and i < n - 1                                    # this layer's output feeds the upsampler conv, not a concat / norm
                tgt = None if (to_up or features) else cat_target(Cout)
                if cfg.up_attn(i):
                    lv = cfg.up_level(i)
                    h = self.resnet_fwd(f"up_blocks.{i}.resnets.{j}.", h, emb_act, B, H, Wd, tape)
                    h = self.transformer_fwd(f"up_blocks.{i}.attentions.{j}.", h, text, B, H, Wd, tape, cfg.transformer_depth[lv], cfg.heads_at(lv),
                                             out=tgt, want_stats=not to_up)
                else:
                    h = self.resnet_fwd(f"up_blocks.{i}.resnets.{j}.", h, emb_act, B, H, Wd, tape, out=tgt, want_stats=not to_up)
            if i < n - 1:
                geo = Geo(H, Wd, stride=1, src_mode=capi.SRC_UPSAMPLE2)   # nearest-2x fused into the conv loader
                sv = {} if save else None
                Mo = B * geo.Ho * geo.Wo
                st = (self._cs_arena, geo.Ho * geo.Wo) if (geo.Ho * geo.Wo) % 64 == 0 else None
                h = layer_fwd(W, lora, f"up_blocks.{i}.upsamplers.0.conv", h, Mo, geo, save=sv, out=None if features else cat_target(Cout), stats=st)
                if save:
                    tape.append(("up", f"up_blocks.{i}.upsamplers.0.conv", dict(sv=sv, B=B, H=H, W=Wd)))
                H, Wd = geo.Ho, geo.Wo
                h = _view(h, B, H * Wd, -1)
            if features:
                feats.append((h, H, Wd))
                if save:
                    tape.append(("feat", None, dict(k=len(feats) - 1)))
        if features:
            if save:
                tape.append(("feats_end", None, dict(B=B, H=H, W=Wd)))
                return feats, tape
            return feats
        sgn = {} if save else None
        hn = self._gn("conv_norm_out", h, capi.ACT_SILU, cfg.norm_eps, sgn)
        out = ops.conv_out_fwd(hn, W.conv_out[0], W.conv_out[1], B, H, Wd)
        if save:
            tape.append(("out", None, dict(sgn=sgn, B=B, H=H, W=Wd, n_skips=1 + sum(cfg.layers_per_block + (1 if i < n - 1 else 0) for i in range(n)))))
            return out, tape
        return out

    @staticmethod
    def tape_first_half(tape):
        """Tape of the first half of the batch of a ``forward(save=True)`` call.  Every saved tensor is batch-major
        ([B, ...] or [B*HW, ...] rows), so the first-half tape is the leading half of each tensor (views, no copies) with
        the recorded ``B`` / ``M`` halved.  Used to run the online (grad) and the target (no-grad) forward of the
        distillation step as ONE 2B-sample launch schedule and back-propagate through the online half only."""
        def half(v, key=None):
            if isinstance(v, HalfSaved):
                return v.t
            if isinstance(v, torch.Tensor):
                assert v.shape[0] % 2 == 0, (key, tuple(v.shape))
                return v[: v.shape[0] // 2]
            if isinstance(v, dict):
                return {k: half(x, k) for k, x in v.items()}
            if isinstance(v, int) and not isinstance(v, bool) and key in ("B", "M"):
                assert v % 2 == 0, (key, v)
                return v // 2
            return v
        return [(kind, p, half(sv)) for kind, p, sv in tape]

    def backward(self, d_eps, tape, d_feats=None, need_input_grad=False, on_late=None):
        """d_eps [B,4,H,W] fp32 -> LoRA grads accumulated in self.lora.grads (if any).
        Feature-tap tapes (``forward(features=True, save=True)``) take ``d_feats`` (list of 9 gradients, entries may
        be None) instead of d_eps.  ``need_input_grad`` also back-propagates through the first resnet and conv_in and
        returns d sample [B,4,H,W] fp32 (the generator step's path through the frozen teacher, sd15_adv.py:1414-1424)."""
        global _SIDE, _DEFER
        W, lora, cfg = self.W, self.lora, self.cfg
        dev_ = d_eps.device if d_eps is not None else self.W.conv_in[0].device
        if lora is not None and WGRAD_SIDE_STREAM and dev_.type == "cuda" and not ops.DETERMINISTIC:     # (reproducible reductions: one stream, one order)
            if self._side is None:
                self._side = WgradSide()
            _SIDE = self._side
        elif lora is not None and WGRAD_DEFER > 1:
            _DEFER = []
        try:
            out = self._backward(d_eps, tape, d_feats, need_input_grad, on_late)
            _flush_deferred()
            return out
        finally:
            if _SIDE is not None:
                _SIDE.join()
            _SIDE = _DEFER = None

    def _backward(self, d_eps, tape, d_feats, need_input_grad, on_late=None):
        W, lora, cfg = self.W, self.lora, self.cfg
        self._arena = ops.StatArena.for_pass(d_eps.device if d_eps is not None else self.W.conv_in[0].device, W, tape[-1][2]["B"], cfg.norm_num_groups)
        kind, _, sv = tape[-1]
        if kind == "out":
            d_hn = ops.conv_out_bwd(d_eps.contiguous(), W.conv_out[0], cfg.block_out_channels[0])
            d_h = self._gn_bwd("conv_norm_out", d_hn, capi.ACT_SILU, cfg.norm_eps, sv["sgn"])
        else:
            assert kind == "feats_end" and d_feats is not None
            d_h = None
        d_skips = {}
        first_resnet = "down_blocks.0.resnets.0."  # its input (conv_in output) has nothing trainable upstream
        idx = len(tape) - 2
        while idx >= 0:
            kind, p, sv = tape[idx]
            if on_late is not None and p is not None and p.startswith("down_blocks"):
                # every LoRA gradient of the up and mid blocks is final (lora.grads[lora.late_offset:]): the data-parallel trainer starts
                # that bucket's all-reduce here, behind the rest of the backward
                if _SIDE is not None:
                    _SIDE.join()
                _flush_deferred()
                on_late()
                on_late = None
            if kind == "feat":
                df = d_feats[sv["k"]]
                if df is not None:
                    d_h = df if d_h is None else ops.add(d_h, df.view_as(d_h))
            elif kind == "resnet":
                d_h = self.resnet_bwd(p, d_h, sv, need_dx=(need_input_grad or p != first_resnet))
            elif kind == "transformer":
                d_h = self.transformer_bwd(p, d_h, sv)
            elif kind == "cat":
                d_h, d_s = ops.split_channels(d_h, sv["Ch"])
                d_skips[sv["skip_index"]] = d_s
            elif kind == "up":
                geo = sv["sv"]["geo"]
                d_up = layer_bwd(W, lora, p, d_h.view(-1, d_h.shape[-1]), sv["sv"])
                d_h = ops.pool2x_sum(d_up.view(sv["B"], 4 * geo.Hs * geo.Ws, -1), sv["B"], geo.Hs, geo.Ws)
            elif kind == "down":
                d_h = layer_bwd(W, lora, p, d_h.view(-1, d_h.shape[-1]), sv["sv"])
                geo = sv["sv"]["geo"]
                d_h = d_h.view(sv["B"], geo.Hs * geo.Ws, -1)
            # the input of this op may also have fed a skip connection: add that gradient
            if kind in ("resnet", "down") and d_h is not None:
                si = self._skip_index_of_input(tape, idx)
                if si is not None and si in d_skips:
                    d_h = ops.add(d_h, d_skips.pop(si).view_as(d_h))
            idx -= 1
        if need_input_grad:
            # d sample = input gradient of conv_in (4 <- C0): the conv_out kernel with in/out-transposed, tap-flipped weights
            sv = tape[0][2]
            w = W.conv_in[0]                                              # [C0, 4, 3, 3]
            wt = w.flip(2, 3).permute(1, 0, 2, 3).contiguous()            # [4, C0, 3, 3]
            return ops.conv_out_fwd(d_h, wt, None, sv["B"], sv["H"], sv["W"])
        return None

    def _skip_index_of_input(self, tape, idx):
        """If the input tensor of tape[idx] is a down-path skip tensor, return its index in the skip
        stack (0 = conv_in output), else None."""
        kind, p, _ = tape[idx]
        if p is None or not p.startswith("down_blocks"):
            # mid_block.resnets.0 consumes the last skip
            if p == "mid_block.resnets.0.":
                return self._n_down_skips() - 1
            return None
        cfg = self.cfg
        parts = p.split(".")
        i = int(parts[1])
        # skip stack: 0 = conv_in; block i pushes layers_per_block layer outputs (+1 downsampler output)
        base = 1 + i * (cfg.layers_per_block + 1)
        if parts[2] == "resnets":
            j = int(parts[3])
            # input of resnet j: previous skip (block output j-1, or previous block's last / conv_in)
            return base + j - 1
        if parts[2] == "attentions":
            return None  # input is the resnet output inside the same layer (not a skip)
        if parts[2] == "downsamplers":
            return base + cfg.layers_per_block - 1
        return None

    def _n_down_skips(self):
        cfg = self.cfg
        n = len(cfg.block_out_channels)
        return 1 + sum(cfg.layers_per_block + (1 if i < n - 1 else 0) for i in range(n))


def __getattr__(name):
    # ``<module>.BF16`` = "the library's 16-bit dtype" for external readers (tests, tools): a call-time lookup, never a captured constant
    if name == "BF16":
        return _act_dtype()
    raise AttributeError(name)
