"""SD3 transformer (MMDiT) runner on the HIP kernels: forward, and the explicit backward that accumulates the LoRA gradients
(SURVEY §8f rank 4).  Interface of the reference's calls (train_pcm_lora_sd3.py:1304-1310, :1336-1366):

    transformer(hidden_states=[B,16,H,W], timestep=[B] float, encoder_hidden_states=[B,Lc,4096], pooled_projections=[B,2048]).sample

Wiring: the reference's copied forward, discriminator_sd3.py:73-137; block internals = diffusers' JointTransformerBlock (see
oracle/mmdit_sd3.py for the restated semantics and what pins them).  Every contraction runs in ``pcm_gemm_bf16`` (LoRA as the
second K-segment, rank 32 zero-padded to the kernels' 64), joint attention in the flash kernels with 64-wide heads, adaLN in
``pcm_layernorm_mod_*``, gates in ``pcm_rowgate_fma``, tanh-GELU in ``pcm_gelu_tanh_*``.  torch is used for memory only: the
token-axis concat / split of the two streams around attention and the [B, 6D] modulation slices.

First version of this path: no fused QKV, no hipGraph capture; measured next round.
"""
import torch

from . import capi, ops
from .mmdit_spec import MMDiTConfig, buffer_spec, lora_target_modules, param_spec
from .model import BF16, LoraState, PackedLayer, layer_bwd, layer_fwd


class MMDiTWeights:
    """Frozen SD3 transformer weights (diffusers key names in, packed bf16 MFMA operands out)."""

    def __init__(self, cfg: MMDiTConfig, state_dict, device, need_bwd=True):
        self.cfg, self.device = cfg, torch.device(device)
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
            # dgrad operands only where a gradient has to pass: everything inside the blocks and proj_out
            bwd = need_bwd and (path.startswith("transformer_blocks.") or path in lora_paths) and not path.endswith(".linear")
            self.layers[path] = PackedLayer(state_dict[k], state_dict[path + ".bias"], self.device, bwd)
        self.pos_embed = state_dict["pos_embed.pos_embed"].to(self.device, torch.float32).reshape(cfg.pos_embed_max_size, cfg.pos_embed_max_size, cfg.inner_dim)
        self._pos_cache = {}

    def pos_crop(self, hp, wp):
        """centre crop of the positional table for an hp x wp token grid, bf16 [hp*wp, D] (cached)."""
        key = (hp, wp)
        if key not in self._pos_cache:
            S = self.cfg.pos_embed_max_size
            if hp > S or wp > S:
                raise ValueError(f"MMDiTWeights: token grid {hp}x{wp} exceeds pos_embed_max_size {S}")
            top, left = (S - hp) // 2, (S - wp) // 2
            self._pos_cache[key] = self.pos_embed[top:top + hp, left:left + wp].reshape(hp * wp, -1).to(BF16).contiguous()
        return self._pos_cache[key]


def sd3_lora_state(cfg: MMDiTConfig, rank=32, lora_alpha=8.0, device="cuda", seed=1, b_std=0.0):
    """LoRA factors for the reference's SD3 LoraConfig (train_pcm_lora_sd3.py:975-988): gaussian init, B = 0."""
    return LoraState(cfg, rank, lora_alpha, device, seed=seed, b_std=b_std, targets=lora_target_modules(cfg), init="gaussian")


class MMDiT:
    """Runner bound to frozen weights and (optionally) LoRA factors."""

    def __init__(self, weights: MMDiTWeights, lora: LoraState = None):
        self.W, self.lora, self.cfg = weights, lora, weights.cfg

    # ---- helpers ----
    def _mod(self, path, semb, B, n):
        """adaLN projection of silu(temb): fp32 [B, n, D]."""
        m = layer_fwd(self.W, None, path, semb, B, out_dtype=torch.float32)
        return m.view(B, n, self.cfg.inner_dim)

    @staticmethod
    def _aff(scale, shift):
        return (1.0 + scale).contiguous(), shift.contiguous()

    def forward(self, hidden_states, timestep, encoder_hidden_states, pooled_projections, save=False):
        cfg, W, lora = self.cfg, self.W, self.lora
        B, Cin, H, Wd = hidden_states.shape
        D, nh, hd = cfg.inner_dim, cfg.num_attention_heads, cfg.attention_head_dim
        hp, wp = H // 2, Wd // 2
        Lx, Lc = hp * wp, encoder_hidden_states.shape[1]
        Mx, Mc = B * Lx, B * Lc
        dev = hidden_states.device
        tape = [] if save else None
        # PatchEmbed: Conv2d(k=2, s=2) as a K=64 GEMM + bias + cropped positional table (as the GEMM's residual operand)
        tok = ops.patchify2x2(hidden_states.float().contiguous(), 0)
        x = layer_fwd(W, None, "pos_embed.proj", tok, Mx, residual=W.pos_crop(hp, wp).repeat(B, 1))
        # CombinedTimestepTextProjEmbeddings
        tp = ops.timestep_embedding_f32(timestep.float().contiguous(), 256)
        te = layer_fwd(W, None, "time_text_embed.timestep_embedder.linear_2",
                       layer_fwd(W, None, "time_text_embed.timestep_embedder.linear_1", tp, B, act=capi.ACT_SILU), B)
        pooled = pooled_projections if pooled_projections.dtype == BF16 else ops.cast_bf16(pooled_projections.float().contiguous())
        pe = layer_fwd(W, None, "time_text_embed.text_embedder.linear_2",
                       layer_fwd(W, None, "time_text_embed.text_embedder.linear_1", pooled, B, act=capi.ACT_SILU), B)
        semb = ops.silu(ops.add(te, pe))
        ctx = encoder_hidden_states if encoder_hidden_states.dtype == BF16 else ops.cast_bf16(encoder_hidden_states.float().contiguous())
        c = layer_fwd(W, None, "context_embedder", ctx.view(Mc, -1), Mc)
        for i in range(cfg.num_layers):
            b = f"transformer_blocks.{i}."
            last = i == cfg.num_layers - 1
            rec = {} if save else None
            m = self._mod(b + "norm1.linear", semb, B, 6)          # shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
            gam_a, sh_a = self._aff(m[:, 1], m[:, 0])
            g_a = m[:, 2].contiguous()
            gam_m, sh_m = self._aff(m[:, 4], m[:, 3])
            g_m = m[:, 5].contiguous()
            xn, mu1, rs1 = ops.layernorm_mod_fwd(x, gam_a, sh_a, Lx)
            if last:
                mc = self._mod(b + "norm1_context.linear", semb, B, 2)     # AdaLayerNormContinuous: scale, shift
                cgam_a, csh_a = self._aff(mc[:, 0], mc[:, 1])
            else:
                mc = self._mod(b + "norm1_context.linear", semb, B, 6)
                cgam_a, csh_a = self._aff(mc[:, 1], mc[:, 0])
                cg_a = mc[:, 2].contiguous()
                cgam_m, csh_m = self._aff(mc[:, 4], mc[:, 3])
                cg_m = mc[:, 5].contiguous()
            cn, cmu1, crs1 = ops.layernorm_mod_fwd(c, cgam_a, csh_a, Lc)
            sq, sk, sv, so, sf0, sf2 = ({} if save else None for _ in range(6))
            cq, ck, cv, co, cf0, cf2 = ({} if save else None for _ in range(6))
            q = torch.cat([layer_fwd(W, lora, b + "attn.to_q", xn, Mx, save=sq).view(B, Lx, D),
                           layer_fwd(W, None, b + "attn.add_q_proj", cn, Mc, save=cq).view(B, Lc, D)], 1)
            k = torch.cat([layer_fwd(W, lora, b + "attn.to_k", xn, Mx, save=sk).view(B, Lx, D),
                           layer_fwd(W, None, b + "attn.add_k_proj", cn, Mc, save=ck).view(B, Lc, D)], 1)
            v = torch.cat([layer_fwd(W, lora, b + "attn.to_v", xn, Mx, save=sv).view(B, Lx, D),
                           layer_fwd(W, None, b + "attn.add_v_proj", cn, Mc, save=cv).view(B, Lc, D)], 1)
            o, lse = ops.attn_fwd(q, k, v, nh, hd)
            ox = o[:, :Lx].contiguous().view(Mx, D)
            x1 = ops.rowgate_fma(layer_fwd(W, lora, b + "attn.to_out.0", ox, Mx, save=so), g_a, Lx, res=x)
            xn2, mu2, rs2 = ops.layernorm_mod_fwd(x1, gam_m, sh_m, Lx)
            h = layer_fwd(W, lora, b + "ff.net.0.proj", xn2, Mx, save=sf0)
            x2 = ops.rowgate_fma(layer_fwd(W, lora, b + "ff.net.2", ops.gelu_tanh_fwd(h), Mx, save=sf2), g_m, Lx, res=x1)
            if save:
                rec.update(x=x, c=c, gam_a=gam_a, g_a=g_a, gam_m=gam_m, g_m=g_m, mu1=mu1, rs1=rs1, cgam_a=cgam_a, cmu1=cmu1, crs1=crs1,
                           sq=sq, sk=sk, sv=sv, so=so, sf0=sf0, sf2=sf2, cq=cq, ck=ck, cv=cv, q=q, k=k, v=v, o=o, lse=lse,
                           x1=x1, mu2=mu2, rs2=rs2, h=h, last=last)
            if not last:
                oc = o[:, Lx:].contiguous().view(Mc, D)
                c1 = ops.rowgate_fma(layer_fwd(W, None, b + "attn.to_add_out", oc, Mc, save=co), cg_a, Lc, res=c)
                cn2, cmu2, crs2 = ops.layernorm_mod_fwd(c1, cgam_m, csh_m, Lc)
                hc = layer_fwd(W, None, b + "ff_context.net.0.proj", cn2, Mc, save=cf0)
                c2 = ops.rowgate_fma(layer_fwd(W, None, b + "ff_context.net.2", ops.gelu_tanh_fwd(hc), Mc, save=cf2), cg_m, Lc, res=c1)
                if save:
                    rec.update(co=co, cf0=cf0, cf2=cf2, cg_a=cg_a, cgam_m=cgam_m, cg_m=cg_m, c1=c1, cmu2=cmu2, crs2=crs2, hc=hc)
                c = c2
            x = x2
            if save:
                tape.append(rec)
        mo = self._mod("norm_out.linear", semb, B, 2)               # AdaLayerNormContinuous: scale, shift
        gam_o, sh_o = self._aff(mo[:, 0], mo[:, 1])
        xo, muo, rso = ops.layernorm_mod_fwd(x, gam_o, sh_o, Lx)
        spo = {} if save else None
        y = layer_fwd(W, lora, "proj_out", xo, Mx, save=spo, out_dtype=torch.float32)
        out = ops.unpatchify2x2(y, B, cfg.out_channels, H, Wd)
        if save:
            tape.append(dict(final=True, x=x, gam_o=gam_o, muo=muo, rso=rso, spo=spo, B=B, H=H, W=Wd, Lx=Lx, Lc=Lc))
            return out, tape
        return out

    def backward(self, d_out, tape):
        """d_out [B,16,H,W] fp32 -> LoRA gradients accumulated into ``self.lora.grads``."""
        cfg, W, lora = self.cfg, self.W, self.lora
        fin = tape[-1]
        B, H, Wd, Lx, Lc = fin["B"], fin["H"], fin["W"], fin["Lx"], fin["Lc"]
        D, nh, hd = cfg.inner_dim, cfg.num_attention_heads, cfg.attention_head_dim
        Mx, Mc = B * Lx, B * Lc
        d_tok = ops.patchify2x2(d_out.float().contiguous(), 1)                       # (p, q, c) columns of proj_out
        d_xo = layer_bwd(W, lora, "proj_out", d_tok, fin["spo"])
        d_x = ops.layernorm_mod_bwd(fin["x"], d_xo, fin["gam_o"], fin["muo"], fin["rso"], Lx)
        d_c = None
        for i in range(cfg.num_layers - 1, -1, -1):
            b = f"transformer_blocks.{i}."
            r = tape[i]
            # image stream: x2 = x1 + g_m * ff(adaLN(x1)) ; x1 = x + g_a * to_out(attn)
            d_h = ops.gelu_tanh_bwd(r["h"], layer_bwd(W, lora, b + "ff.net.2", ops.rowgate_fma(d_x, r["g_m"], Lx), r["sf2"]))
            d_xn2 = layer_bwd(W, lora, b + "ff.net.0.proj", d_h, r["sf0"])
            d_x1 = ops.layernorm_mod_bwd(r["x1"], d_xn2, r["gam_m"], r["mu2"], r["rs2"], Lx, dres=d_x)
            d_ox = layer_bwd(W, lora, b + "attn.to_out.0", ops.rowgate_fma(d_x1, r["g_a"], Lx), r["so"])
            # context stream (absent in the last block: its attention output for the text tokens is dropped)
            if r["last"]:
                d_oc = torch.zeros(B, Lc, D, dtype=BF16, device=d_ox.device)
                d_c1 = None
            else:
                d_hc = ops.gelu_tanh_bwd(r["hc"], layer_bwd(W, None, b + "ff_context.net.2", ops.rowgate_fma(d_c, r["cg_m"], Lc), r["cf2"]))
                d_cn2 = layer_bwd(W, None, b + "ff_context.net.0.proj", d_hc, r["cf0"])
                d_c1 = ops.layernorm_mod_bwd(r["c1"], d_cn2, r["cgam_m"], r["cmu2"], r["crs2"], Lc, dres=d_c)
                d_oc = layer_bwd(W, None, b + "attn.to_add_out", ops.rowgate_fma(d_c1, r["cg_a"], Lc), r["co"]).view(B, Lc, D)
            d_o = torch.cat([d_ox.view(B, Lx, D), d_oc], 1)
            dq, dk, dv = ops.attn_bwd(r["q"], r["k"], r["v"], r["o"], d_o, r["lse"], nh, hd)
            d_xn = layer_bwd(W, lora, b + "attn.to_q", dq[:, :Lx].contiguous().view(Mx, D), r["sq"])
            d_xn = layer_bwd(W, lora, b + "attn.to_k", dk[:, :Lx].contiguous().view(Mx, D), r["sk"], residual=d_xn)
            d_xn = layer_bwd(W, lora, b + "attn.to_v", dv[:, :Lx].contiguous().view(Mx, D), r["sv"], residual=d_xn)
            d_x = ops.layernorm_mod_bwd(r["x"], d_xn, r["gam_a"], r["mu1"], r["rs1"], Lx, dres=d_x1)
            if i > 0:       # block 0's text input comes from the (unadapted) context_embedder: nothing trainable upstream
                d_cn = layer_bwd(W, None, b + "attn.add_q_proj", dq[:, Lx:].contiguous().view(Mc, D), r["cq"])
                d_cn = layer_bwd(W, None, b + "attn.add_k_proj", dk[:, Lx:].contiguous().view(Mc, D), r["ck"], residual=d_cn)
                d_cn = layer_bwd(W, None, b + "attn.add_v_proj", dv[:, Lx:].contiguous().view(Mc, D), r["cv"], residual=d_cn)
                d_c = ops.layernorm_mod_bwd(r["c"], d_cn, r["cgam_a"], r["cmu1"], r["crs1"], Lc, dres=d_c1)
        return None
