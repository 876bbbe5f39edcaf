"""SD3 transformer (MMDiT) runner on the HIP kernels: forward, and the explicit backward that accumulates the LoRA gradients
(SURVEY §8f rank 4).  Interface of the reference's calls (train_pcm_lora_sd3.py:1304-1310, :1336-1366):

    transformer(hidden_states=[B,16,H,W], timestep=[B] float, encoder_hidden_states=[B,Lc,4096], pooled_projections=[B,2048]).sample

Wiring: the reference's copied forward, discriminator_sd3.py:73-137; block internals = diffusers' JointTransformerBlock (see
oracle/mmdit_sd3.py for the restated semantics and what pins them).  Every contraction runs in ``pcm_gemm_bf16`` (LoRA as the
second K-segment, rank 32 zero-padded to the kernels' 64), joint attention in the flash kernels with 64-wide heads, adaLN in
``pcm_layernorm_mod_*``, gates in ``pcm_rowgate_fma``, tanh-GELU in ``pcm_gelu_tanh_*``, the adaLN parameter gradients in
``pcm_mod_grad``.  What torch does here is memory movement and bookkeeping on per-sample vectors: the token-axis concat / split of the
two streams around attention, slicing the [B, 6, D] modulation outputs into (1 + scale, shift, gate) rows, stacking their gradients
back, and the fp32 running sum of the [B, D] conditioning gradient over the blocks.

First version of this path: q/k/v fused per stream, no hipGraph capture; measured next round.
"""
import os

import torch

from . import capi, ops
from .mmdit_spec import MMDiTConfig, buffer_spec, lora_target_modules, param_spec
from .precision import act_dtype as _act_dtype
from . import model as _model
from .model import LoraState, PackedLayer, layer_bwd, layer_fwd
from .ops import Seg
from .precision import precision

# opt-in until measured (tools/jobs/r06_o_*): LoRA weight gradients of the backward on a second stream
WGRAD_SIDE_STREAM = os.environ.get("PCM_SD3_WGRAD_SIDE", "0") == "1"

# debug hook: PCM_MMDIT_QKV=0 runs the six q/k/v projections of a block as separate layers (A/B measurement)
FUSE_QKV = os.environ.get("PCM_MMDIT_QKV", "1") != "0"


class MMDiTWeights:
    """Frozen SD3 transformer weights (diffusers key names in, packed bf16 MFMA operands out)."""

    def __init__(self, cfg: MMDiTConfig, state_dict, device, need_bwd=True):
        self.cfg, self.device = cfg, torch.device(device)
        self.format = precision()            # the 16-bit format the operands below are packed in (pcm_amd/precision.py)
        spec = param_spec(cfg) + buffer_spec(cfg)
        missing = [k for k, _ in spec if k not in state_dict]
        if missing:
            raise KeyError(f"MMDiTWeights: {len(missing)} missing keys, e.g. {missing[:3]}")
        for k, shp in spec:
            if tuple(state_dict[k].shape) != tuple(shp):
                raise ValueError(f"MMDiTWeights: {k} has shape {tuple(state_dict[k].shape)}, expected {shp}")
        self.layers, self.qkv, self.qkv_bwd = {}, {}, {}
        lora_paths = {p for p, _ in lora_target_modules(cfg)}
        for k, _ in param_spec(cfg):
            if not k.endswith(".weight"):
                continue
            path = k[:-7]
            # dgrad operands where a gradient may have to pass: inside the blocks, proj_out, and (for the adversarial trainers' LoRA
            # list, which adapts the conditioning path too) the adaLN projections and the second embedder linears
            bwd = need_bwd and (path.startswith("transformer_blocks.") or path in lora_paths or path.endswith("linear_2") or
                                path in ("pos_embed.proj", "norm_out.linear"))
            self.layers[path] = PackedLayer(state_dict[k], state_dict[path + ".bias"], self.device, bwd)
        self.pos_embed = state_dict["pos_embed.pos_embed"].to(self.device, torch.float32).reshape(cfg.pos_embed_max_size, cfg.pos_embed_max_size, cfg.inner_dim)
        self._pos_cache = {}
        # q/k/v of each stream as ONE projection: rows [Wq; Wk; Wv] (+ concatenated bias), the normalised activation is read once and
        # attention reads q/k/v in place with row stride 3D; dgrad operand [K][3N] = [Wq^T | Wk^T | Wv^T]
        self.qkv_bias, self.cqkv, self.cqkv_bias, self.cqkv_bwd = {}, {}, {}, {}
        for i in range(cfg.num_layers):
            pa = f"transformer_blocks.{i}.attn."
            for names, wf, wb, bs in ((("to_q", "to_k", "to_v"), self.qkv, self.qkv_bwd, self.qkv_bias),
                                      (("add_q_proj", "add_k_proj", "add_v_proj"), self.cqkv, self.cqkv_bwd, self.cqkv_bias)):
                Ls = [self.layers[pa + n] for n in names]
                wf[pa] = torch.cat([l.w_fwd.view(l.N, l.K) for l in Ls]).contiguous()
                bs[pa] = torch.cat([l.bias for l in Ls]).contiguous()
                if all(l.w_bwd is not None for l in Ls):
                    wb[pa] = torch.cat([l.w_bwd.view(l.K, l.N) for l in Ls], dim=1).contiguous()

    def pos_crop(self, hp, wp):
        """centre crop of the positional table for an hp x wp token grid, bf16 [hp*wp, D] (cached)."""
        key = (hp, wp)
        if key not in self._pos_cache:
            S = self.cfg.pos_embed_max_size
            if hp > S or wp > S:
                raise ValueError(f"MMDiTWeights: token grid {hp}x{wp} exceeds pos_embed_max_size {S}")
            top, left = (S - hp) // 2, (S - wp) // 2
            self._pos_cache[key] = self.pos_embed[top:top + hp, left:left + wp].reshape(hp * wp, -1).to(_act_dtype()).contiguous()
        return self._pos_cache[key]


def sd3_lora_state(cfg: MMDiTConfig, rank=32, lora_alpha=8.0, device="cuda", seed=1, b_std=0.0, targets=None, init="gaussian"):
    """LoRA factors for the reference's SD3 LoraConfig.  Default: train_pcm_lora_sd3.py:975-988 (8 suffixes, gaussian init, B = 0);
    ``targets=mmdit_spec.LORA_TARGETS_SD3_ADV, init="kaiming"``: the adversarial trainers' 22-entry list with peft's default init
    (train_pcm_lora_sd3_adv.py:987-1016, where init_lora_weights="gaussian" is commented out)."""
    tl = lora_target_modules(cfg) if targets is None else lora_target_modules(cfg, targets)
    return LoraState(cfg, rank, lora_alpha, device, seed=seed, b_std=b_std, targets=tl, init=init)


class MMDiT:
    """Runner bound to frozen weights and (optionally) LoRA factors.  Whatever subset of the Linear / patch-conv modules the LoRA
    state adapts is honoured: the 8-suffix list of the base trainer touches the image stream only; with the adversarial trainers'
    list the context stream, the adaLN projections (-> gradients of the modulation vectors, ``pcm_mod_grad``), the time / text
    embedders, context_embedder and the patch embedding receive gradients as well."""

    def __init__(self, weights: MMDiTWeights, lora: LoraState = None):
        self.W, self.lora, self.cfg = weights, lora, weights.cfg
        mods = lora.modules if lora is not None else {}
        self.emb_lora = any(p.startswith("time_text_embed.") for p in mods)
        # gradients of the modulation vectors are needed when anything on the conditioning path is trainable: the adaLN projections
        # themselves, or the embedders upstream of silu(temb) (then they also flow through the frozen projections incl. norm_out.linear)
        self.mod_lora = self.emb_lora or any(p.endswith("norm1.linear") or p.endswith("norm1_context.linear") for p in mods)
        self.ctx_in_lora = "context_embedder" in mods
        self.pos_lora = "pos_embed.proj" in mods

    def _fusable(self, pa, save):
        """q/k/v of block ``pa`` can run as fused projections: the context-stream projections carry no LoRA (true for both reference
        lists) and the image stream's three either all do (concatenated operands exist) or none does."""
        if not FUSE_QKV:
            return False
        W, lora = self.W, self.lora
        if save and (pa not in W.qkv_bwd or pa not in W.cqkv_bwd):
            return False
        if lora is None:
            return True
        mods = lora.modules
        if any((pa + n) in mods for n in ("add_q_proj", "add_k_proj", "add_v_proj")):
            return False
        have = [(pa + n) in mods for n in ("to_q", "to_k", "to_v")]
        return (all(have) and pa in lora.qkv) or not any(have)

    @staticmethod
    def _aff(scale, shift):
        return (1.0 + scale).contiguous(), shift.contiguous()

    def forward(self, hidden_states, timestep, encoder_hidden_states, pooled_projections, save=False, features=False):
        """-> [B,16,H,W] fp32 (``.sample``).  ``save``: also the tape for ``backward``.  ``features``: return the image-stream
        hidden states after every block instead (the discriminator's ``modified_forward``, discriminator_sd3.py:36-137)."""
        cfg, W, lora = self.cfg, self.W, self.lora
        B, Cin, H, Wd = hidden_states.shape
        D, nh, hd = cfg.inner_dim, cfg.num_attention_heads, cfg.attention_head_dim
        hp, wp = H // 2, Wd // 2
        Lx, Lc = hp * wp, encoder_hidden_states.shape[1]
        Mx, Mc = B * Lx, B * Lc
        tape = [] if save else None
        S = (lambda: {}) if save else (lambda: None)
        # PatchEmbed: Conv2d(k=2, s=2) as a K=64 GEMM + bias + cropped positional table (as the GEMM's residual operand)
        tok = ops.patchify2x2(hidden_states.float().contiguous(), 0)
        spos = S()
        x = layer_fwd(W, lora, "pos_embed.proj", tok, Mx, save=spos, residual=W.pos_crop(hp, wp).repeat(B, 1))
        # CombinedTimestepTextProjEmbeddings: linear_2(silu(linear_1(.))) for the timestep projection and the pooled text embedding
        tp = ops.timestep_embedding_f32(timestep.float().contiguous(), 256)
        pooled = pooled_projections if pooled_projections.dtype == _act_dtype() else ops.cast_bf16(pooled_projections.float().contiguous())
        emb = {}
        for name, inp in (("timestep_embedder", tp), ("text_embedder", pooled)):
            s1, s2 = S(), S()
            pre = "time_text_embed." + name
            if self.emb_lora and save:      # the pre-activation is needed for the backward: no fused SiLU epilogue
                z1 = layer_fwd(W, lora, pre + ".linear_1", inp, B, save=s1)
                h1 = ops.silu(z1)
            else:
                z1, h1 = None, layer_fwd(W, lora, pre + ".linear_1", inp, B, save=s1, act=capi.ACT_SILU)
            emb[name] = (layer_fwd(W, lora, pre + ".linear_2", h1, B, save=s2), s1, s2, z1)
        temb = ops.add(emb["timestep_embedder"][0], emb["text_embedder"][0])
        semb = ops.silu(temb)
        ctx = encoder_hidden_states if encoder_hidden_states.dtype == _act_dtype() else ops.cast_bf16(encoder_hidden_states.float().contiguous())
        sctx = S()
        c = layer_fwd(W, lora, "context_embedder", ctx.view(Mc, -1), Mc, save=sctx)
        feats = []
        for i in range(cfg.num_layers):
            b = f"transformer_blocks.{i}."
            last = i == cfg.num_layers - 1
            rec = {} if save else None
            sm, smc = S(), S()
            m = layer_fwd(W, lora, b + "norm1.linear", semb, B, save=sm, out_dtype=torch.float32).view(B, 6, D)
            # shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
            gam_a, sh_a = self._aff(m[:, 1], m[:, 0])
            g_a = m[:, 2].contiguous()
            gam_m, sh_m = self._aff(m[:, 4], m[:, 3])
            g_m = m[:, 5].contiguous()
            xn, mu1, rs1 = ops.layernorm_mod_fwd(x, gam_a, sh_a, Lx)
            mc = layer_fwd(W, lora, b + "norm1_context.linear", semb, B, save=smc, out_dtype=torch.float32).view(B, 2 if last else 6, D)
            if last:                                                   # AdaLayerNormContinuous: scale, shift
                cgam_a, csh_a = self._aff(mc[:, 0], mc[:, 1])
            else:
                cgam_a, csh_a = self._aff(mc[:, 1], mc[:, 0])
                cg_a = mc[:, 2].contiguous()
                cgam_m, csh_m = self._aff(mc[:, 4], mc[:, 3])
                cg_m = mc[:, 5].contiguous()
            cn, cmu1, crs1 = ops.layernorm_mod_fwd(c, cgam_a, csh_a, Lc)
            sq, sk, sv, so, sf0, sf2 = (S() for _ in range(6))
            cq, ck, cv, co, cf0, cf2 = (S() for _ in range(6))
            pa = b + "attn."
            fuse = self._fusable(pa, save)
            fq = t3 = None
            if fuse:
                # both streams' q/k/v as one projection each; LoRA (image stream): concatenated rank-192 down-projection + block-diagonal
                # s*B K-segment (LoraState.qkv); one token-axis concat of the [., 3D] rows instead of three
                segs = [Seg(xn, W.qkv[pa])]
                fq = lora.qkv.get(pa) if lora is not None else None
                if fq is not None:
                    t3 = torch.empty(Mx, fq.r3, dtype=_act_dtype(), device=xn.device)
                    ops.gemm([Seg(xn, fq.A_cat_fwd)], Mx, fq.r3, t3)
                    segs.append(Seg(t3, fq.Bs_cat_fwd, k_algo=lora.rank))
                qkv_x = torch.empty(Mx, 3 * D, dtype=_act_dtype(), device=xn.device)
                ops.gemm(segs, Mx, 3 * D, qkv_x, bias=W.qkv_bias[pa])
                qkv_c = torch.empty(Mc, 3 * D, dtype=_act_dtype(), device=xn.device)
                ops.gemm([Seg(cn, W.cqkv[pa])], Mc, 3 * D, qkv_c, bias=W.cqkv_bias[pa])
                j3 = torch.cat([qkv_x.view(B, Lx, 3 * D), qkv_c.view(B, Lc, 3 * D)], 1)
                q, k, v = j3[:, :, :D], j3[:, :, D:2 * D], j3[:, :, 2 * D:]
            else:
                q = torch.cat([layer_fwd(W, lora, b + "attn.to_q", xn, Mx, save=sq).view(B, Lx, D),
                               layer_fwd(W, lora, b + "attn.add_q_proj", cn, Mc, save=cq).view(B, Lc, D)], 1)
                k = torch.cat([layer_fwd(W, lora, b + "attn.to_k", xn, Mx, save=sk).view(B, Lx, D),
                               layer_fwd(W, lora, b + "attn.add_k_proj", cn, Mc, save=ck).view(B, Lc, D)], 1)
                v = torch.cat([layer_fwd(W, lora, b + "attn.to_v", xn, Mx, save=sv).view(B, Lx, D),
                               layer_fwd(W, lora, b + "attn.add_v_proj", cn, Mc, save=cv).view(B, Lc, D)], 1)
            o, lse = ops.attn_fwd(q, k, v, nh, hd)
            ox = o[:, :Lx].contiguous().view(Mx, D)
            a_x = layer_fwd(W, lora, b + "attn.to_out.0", ox, Mx, save=so)
            x1 = ops.rowgate_fma(a_x, g_a, Lx, res=x)
            xn2, mu2, rs2 = ops.layernorm_mod_fwd(x1, gam_m, sh_m, Lx)
            h = layer_fwd(W, lora, b + "ff.net.0.proj", xn2, Mx, save=sf0)
            f_x = layer_fwd(W, lora, b + "ff.net.2", ops.gelu_tanh_fwd(h), Mx, save=sf2)
            x2 = ops.rowgate_fma(f_x, g_m, Lx, res=x1)
            if save:
                rec.update(x=x, c=c, gam_a=gam_a, g_a=g_a, gam_m=gam_m, g_m=g_m, mu1=mu1, rs1=rs1, cgam_a=cgam_a, cmu1=cmu1, crs1=crs1,
                           sq=sq, sk=sk, sv=sv, so=so, sf0=sf0, sf2=sf2, cq=cq, ck=ck, cv=cv, q=q, k=k, v=v, o=o, lse=lse,
                           x1=x1, mu2=mu2, rs2=rs2, h=h, last=last, sm=sm, smc=smc, fuse=fuse, xn=xn, cn=cn, t3=t3)
                if self.mod_lora:
                    rec.update(a_x=a_x, f_x=f_x)
            if not last:
                oc = o[:, Lx:].contiguous().view(Mc, D)
                a_c = layer_fwd(W, lora, b + "attn.to_add_out", oc, Mc, save=co)
                c1 = ops.rowgate_fma(a_c, cg_a, Lc, res=c)
                cn2, cmu2, crs2 = ops.layernorm_mod_fwd(c1, cgam_m, csh_m, Lc)
                hc = layer_fwd(W, lora, b + "ff_context.net.0.proj", cn2, Mc, save=cf0)
                f_c = layer_fwd(W, lora, b + "ff_context.net.2", ops.gelu_tanh_fwd(hc), Mc, save=cf2)
                c2 = ops.rowgate_fma(f_c, cg_m, Lc, res=c1)
                if save:
                    rec.update(co=co, cf0=cf0, cf2=cf2, cg_a=cg_a, cgam_m=cgam_m, cg_m=cg_m, c1=c1, cmu2=cmu2, crs2=crs2, hc=hc)
                    if self.mod_lora:
                        rec.update(a_c=a_c, f_c=f_c)
                c = c2
            x = x2
            if features:
                feats.append(x.view(B, Lx, D))
            if save:
                tape.append(rec)
        head = dict(final=True, B=B, H=H, W=Wd, Lx=Lx, Lc=Lc, spos=spos, sctx=sctx, emb=emb, temb=temb, features=features)
        if features:            # discriminator feature taps: no norm_out / proj_out
            if save:
                tape.append(head)
                return feats, tape
            return feats
        smo = S()
        mo = layer_fwd(W, None, "norm_out.linear", semb, B, save=smo, out_dtype=torch.float32).view(B, 2, D)   # AdaLayerNormContinuous: scale, shift
        gam_o, sh_o = self._aff(mo[:, 0], mo[:, 1])
        xo, muo, rso = ops.layernorm_mod_fwd(x, gam_o, sh_o, Lx)
        spo = S()
        y = layer_fwd(W, lora, "proj_out", xo, Mx, save=spo, out_dtype=torch.float32)
        out = ops.unpatchify2x2(y, B, cfg.out_channels, H, Wd)
        if save:
            head.update(x=x, gam_o=gam_o, muo=muo, rso=rso, spo=spo, smo=smo)
            tape.append(head)
            return out, tape
        return out

    @staticmethod
    def tape_first_half(tape):
        """Tape of the first half of the batch of a ``forward(save=True)`` call (as model.UNet.tape_first_half): every saved tensor is batch-major
        ([B, ...], [B*L, ...] rows or [B, heads, L]), so the first-half tape is the leading half of each (views) with the recorded ``B`` / ``M``
        halved.  Lets the distillation step run its online (grad) and target (no-grad) forwards as ONE 2B-sample launch schedule."""
        def half(v, key=None):
            if isinstance(v, torch.Tensor):
                assert v.shape[0] % 2 == 0, (key, tuple(v.shape))
                return v[: v.shape[0] // 2]
            if isinstance(v, dict):
                return {k: half(x, k) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return type(v)(half(x, key) for x in v)
            if isinstance(v, int) and not isinstance(v, bool) and key in ("B", "M"):
                assert v % 2 == 0, (key, v)
                return v // 2
            return v
        return [half(r) for r in tape]

    def backward(self, d_out, tape, d_feats=None, need_input_grad=False):
        """d_out [B,16,H,W] fp32 -> LoRA gradients accumulated into ``self.lora.grads`` (if any).  Feature-tap tapes
        (``forward(features=True, save=True)``) take ``d_feats`` (one gradient per block output, entries may be None) instead.
        ``need_input_grad``: also return d hidden_states [B,16,H,W] fp32 (the generator step's path through the frozen teacher of
        the discriminator, train_pcm_lora_sd3_adv.py:1492-1506)."""
        # LoRA weight gradients beside the input-gradient chain on a second stream (model.WgradSide; PCM_SD3_WGRAD_SIDE=1)
        dev_ = d_out.device if d_out is not None else self.W.device
        if self.lora is not None and WGRAD_SIDE_STREAM and dev_.type == "cuda" and not ops.DETERMINISTIC:
            if getattr(self, "_side", None) is None:
                self._side = _model.WgradSide()
            _model._SIDE = self._side
        elif self.lora is not None and _model.WGRAD_DEFER > 1:      # weight-gradient jobs collected across modules (model._wgrad / _flush_deferred)
            _model._DEFER = []
        try:
            out = self._backward(d_out, tape, d_feats, need_input_grad)
            _model._flush_deferred()
            return out
        finally:
            if _model._SIDE is not None:
                _model._SIDE.join()
            _model._SIDE = _model._DEFER = None

    def _backward(self, d_out, tape, d_feats, need_input_grad):
        cfg, W, lora = self.cfg, self.W, self.lora
        fin = tape[-1]
        B, H, Wd, Lx, Lc = fin["B"], fin["H"], fin["W"], fin["Lx"], fin["Lc"]
        D, nh, hd = cfg.inner_dim, cfg.num_attention_heads, cfg.attention_head_dim
        Mx, Mc = B * Lx, B * Lc
        d_c = None
        d_semb = None            # fp32 [B, D]: gradient of silu(temb), fed by every adaLN projection

        def mod_bwd(path, parts, saved):
            nonlocal d_semb
            dm = torch.stack(parts, 1).reshape(B, -1).to(_act_dtype()).contiguous()
            g = layer_bwd(W, lora, path, dm, saved).float()
            d_semb = g if d_semb is None else d_semb + g
        if fin["features"]:
            assert d_feats is not None and len(d_feats) == cfg.num_layers
            d_x = None
        else:
            d_tok = ops.patchify2x2(d_out.float().contiguous(), 1)                       # (p, q, c) columns of proj_out
            d_xo = layer_bwd(W, lora, "proj_out", d_tok, fin["spo"])
            d_x = ops.layernorm_mod_bwd(fin["x"], d_xo, fin["gam_o"], fin["muo"], fin["rso"], Lx)
            if self.mod_lora:
                dgam_o, dsh_o = ops.mod_grad(fin["x"], d_xo, B, fin["muo"], fin["rso"])
                mod_bwd("norm_out.linear", [dgam_o, dsh_o], fin["smo"])                  # (scale, shift)
        for i in range(cfg.num_layers - 1, -1, -1):
            b = f"transformer_blocks.{i}."
            r = tape[i]
            if d_feats is not None and d_feats[i] is not None:
                df = d_feats[i].reshape(Mx, D)
                d_x = df if d_x is None else ops.add(d_x, df)
            if d_x is None:           # nothing downstream of this block reaches the loss
                continue
            # image stream: x2 = x1 + g_m * ff(adaLN(x1)) ; x1 = x + g_a * to_out(attn)
            d_h = ops.gelu_tanh_bwd(r["h"], layer_bwd(W, lora, b + "ff.net.2", ops.rowgate_fma(d_x, r["g_m"], Lx), r["sf2"]))
            d_xn2 = layer_bwd(W, lora, b + "ff.net.0.proj", d_h, r["sf0"])
            d_x1 = ops.layernorm_mod_bwd(r["x1"], d_xn2, r["gam_m"], r["mu2"], r["rs2"], Lx, dres=d_x)
            d_ox = layer_bwd(W, lora, b + "attn.to_out.0", ops.rowgate_fma(d_x1, r["g_a"], Lx), r["so"])
            if self.mod_lora:
                dg_m, _ = ops.mod_grad(r["f_x"], d_x, B, want_b=False)
                dgam_m, dsh_m = ops.mod_grad(r["x1"], d_xn2, B, r["mu2"], r["rs2"])
                dg_a, _ = ops.mod_grad(r["a_x"], d_x1, B, want_b=False)
            # context stream (absent in the last block: its attention output for the text tokens is dropped)
            if r["last"] or d_c is None:
                d_oc = torch.zeros(B, Lc, D, dtype=_act_dtype(), device=d_ox.device)
                d_c1 = None
                cparts = None
            else:
                d_hc = ops.gelu_tanh_bwd(r["hc"], layer_bwd(W, lora, b + "ff_context.net.2", ops.rowgate_fma(d_c, r["cg_m"], Lc), r["cf2"]))
                d_cn2 = layer_bwd(W, lora, b + "ff_context.net.0.proj", d_hc, r["cf0"])
                d_c1 = ops.layernorm_mod_bwd(r["c1"], d_cn2, r["cgam_m"], r["cmu2"], r["crs2"], Lc, dres=d_c)
                d_oc = layer_bwd(W, lora, b + "attn.to_add_out", ops.rowgate_fma(d_c1, r["cg_a"], Lc), r["co"]).view(B, Lc, D)
                if self.mod_lora:
                    cdg_m, _ = ops.mod_grad(r["f_c"], d_c, B, want_b=False)
                    cdgam_m, cdsh_m = ops.mod_grad(r["c1"], d_cn2, B, r["cmu2"], r["crs2"])
                    cdg_a, _ = ops.mod_grad(r["a_c"], d_c1, B, want_b=False)
                    cparts = (cdg_a, cdsh_m, cdgam_m, cdg_m)
            d_o = torch.cat([d_ox.view(B, Lx, D), d_oc], 1)
            pa = b + "attn."
            if r["fuse"]:
                d3 = torch.empty(B, Lx + Lc, 3 * D, dtype=_act_dtype(), device=d_o.device)      # [dq | dk | dv], written in place by attention
                ops.attn_bwd(r["q"], r["k"], r["v"], r["o"], d_o, r["lse"], nh, hd, out=(d3[:, :, :D], d3[:, :, D:2 * D], d3[:, :, 2 * D:]))
                d3x = d3[:, :Lx].contiguous().view(Mx, 3 * D)
                segs = [Seg(d3x, W.qkv_bwd[pa])]
                fq = lora.qkv.get(pa) if lora is not None else None
                if fq is not None:
                    rk, t3, xn_ = lora.rank, r["t3"], r["xn"]
                    u3 = torch.empty(Mx, fq.r3, dtype=_act_dtype(), device=d_o.device)
                    ops.gemm([Seg(d3x, fq.Bs_cat_bwd, k_algo=D)], Mx, fq.r3, u3)
                    def wg(d3x=d3x, t3=t3, xn_=xn_, u3=u3, fq=fq, rk=rk):
                        with ops.wgrad_batch():
                            for jj, lm in enumerate((fq.q, fq.k, fq.v)):
                                ops.lora_wgrad(d3x[:, jj * D:(jj + 1) * D], t3[:, jj * rk:(jj + 1) * rk], lm.gB, lora.scaling, Mx, G=D, g_stride=rk, r_stride=1,
                                               ldb=3 * D, lds=fq.r3)
                                ops.lora_wgrad(xn_, u3[:, jj * rk:(jj + 1) * rk], lm.gA, 1.0, Mx, G=fq.K, g_stride=1, r_stride=fq.K, lds=fq.r3)
                    _model._wgrad(wg, d3x, t3, xn_, u3)
                    segs.append(Seg(u3, fq.A_cat_bwd))
                d_xn = torch.empty(Mx, D, dtype=_act_dtype(), device=d_o.device)
                ops.gemm(segs, Mx, D, d_xn)
                dq = dk = dv = None
            else:
                dq, dk, dv = ops.attn_bwd(r["q"], r["k"], r["v"], r["o"], d_o, r["lse"], nh, hd)
                d_xn = layer_bwd(W, lora, b + "attn.to_q", dq[:, :Lx].contiguous().view(Mx, D), r["sq"])
                d_xn = layer_bwd(W, lora, b + "attn.to_k", dk[:, :Lx].contiguous().view(Mx, D), r["sk"], residual=d_xn)
                d_xn = layer_bwd(W, lora, b + "attn.to_v", dv[:, :Lx].contiguous().view(Mx, D), r["sv"], residual=d_xn)
            d_x = ops.layernorm_mod_bwd(r["x"], d_xn, r["gam_a"], r["mu1"], r["rs1"], Lx, dres=d_x1)
            if self.mod_lora:
                dgam_a, dsh_a = ops.mod_grad(r["x"], d_xn, B, r["mu1"], r["rs1"])
                mod_bwd(b + "norm1.linear", [dsh_a, dgam_a, dg_a, dsh_m, dgam_m, dg_m], r["sm"])
            # block 0's text input comes from context_embedder: its input gradient matters only if that layer (or something
            # upstream of the modulation) is adapted
            need_dc = i > 0 or self.ctx_in_lora or self.mod_lora
            if need_dc:
                if r["fuse"]:
                    d_cn = torch.empty(Mc, D, dtype=_act_dtype(), device=d_o.device)
                    ops.gemm([Seg(d3[:, Lx:].contiguous().view(Mc, 3 * D), W.cqkv_bwd[pa])], Mc, D, d_cn)
                else:
                    d_cn = layer_bwd(W, lora, b + "attn.add_q_proj", dq[:, Lx:].contiguous().view(Mc, D), r["cq"])
                    d_cn = layer_bwd(W, lora, b + "attn.add_k_proj", dk[:, Lx:].contiguous().view(Mc, D), r["ck"], residual=d_cn)
                    d_cn = layer_bwd(W, lora, b + "attn.add_v_proj", dv[:, Lx:].contiguous().view(Mc, D), r["cv"], residual=d_cn)
                d_c = ops.layernorm_mod_bwd(r["c"], d_cn, r["cgam_a"], r["cmu1"], r["crs1"], Lc, dres=d_c1)
                if self.mod_lora:
                    cdgam_a, cdsh_a = ops.mod_grad(r["c"], d_cn, B, r["cmu1"], r["crs1"])
                    if r["last"]:
                        mod_bwd(b + "norm1_context.linear", [cdgam_a, cdsh_a], r["smc"])                   # (scale, shift)
                    else:
                        z = torch.zeros_like(cdgam_a)
                        cdg_a, cdsh_m, cdgam_m, cdg_m = cparts if cparts is not None else (z, z, z, z)
                        mod_bwd(b + "norm1_context.linear", [cdsh_a, cdgam_a, cdg_a, cdsh_m, cdgam_m, cdg_m], r["smc"])
        # ---- below the blocks ----
        if d_c is not None and self.ctx_in_lora:
            layer_bwd(W, lora, "context_embedder", d_c, fin["sctx"], need_dx=False)
        if d_semb is not None and self.emb_lora:
            d_temb = ops.silu_bwd(fin["temb"], d_semb.to(_act_dtype()).contiguous())          # semb = silu(temb), temb = te + pe
            for name in ("timestep_embedder", "text_embedder"):
                _, s1, s2, z1 = fin["emb"][name]
                pre = "time_text_embed." + name
                d_h1 = layer_bwd(W, lora, pre + ".linear_2", d_temb, s2)
                layer_bwd(W, lora, pre + ".linear_1", ops.silu_bwd(z1, d_h1), s1, need_dx=False)
        if d_x is None:
            return None
        if not (need_input_grad or self.pos_lora):
            return None
        d_tok0 = layer_bwd(W, lora, "pos_embed.proj", d_x, fin["spos"], need_dx=need_input_grad)
        if not need_input_grad:
            return None
        return ops.unpatchify2x2(d_tok0.float(), B, cfg.in_channels, H, Wd, order=0)


def __getattr__(name):
    # ``<module>.BF16`` = "the library's 16-bit dtype" for external readers (tests, tools): a call-time lookup, never a captured constant
    if name == "BF16":
        return _act_dtype()
    raise AttributeError(name)
