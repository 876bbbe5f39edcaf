"""Restatement (plain torch, CPU, fp32, autograd) of the SD3 transformer the SD3 trainer distils.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED for the block internals: the model is diffusers' ``SD3Transformer2DModel`` (+ peft 0.9.0 LoRA), an UN-VENDORED
dependency pinned by /root/reference/code/text_to_image_sd3/environment.sd3.yaml and not installed here, so there is no golden
vector from the reference for it.  What anchors this restatement:
  * the call order pos_embed -> time_text_embed(timestep, pooled) -> context_embedder -> blocks (returning
    (encoder_hidden_states, hidden_states)) -> norm_out(hidden_states, temb) -> proj_out -> unpatchify einsum "nhwpqc->nchpwq"
    is the reference's own copied forward, code/text_to_image_sd3/discriminator_sd3.py:73-137;
  * the key / shape enumeration below reproduces the published SD3-medium parameter count 2 028 328 000
    (tests/test_emu_mmdit.py::test_spec_counts);
  * LoRA placement follows the reference's LoraConfig (train_pcm_lora_sd3.py:975-988: r = --lora_rank (run.sh: 32),
    init_lora_weights="gaussian", target suffixes to_k,to_q,to_v,to_out.0,proj_in,proj_out,ff.net.0.proj,ff.net.2; peft matches
    ``name == t or name.endswith("." + t)``, so the context stream's add_*_proj / to_add_out / ff_context are NOT adapted and the final
    proj_out is); lora_alpha is peft's default 8.
Block semantics restated from the pinned diffusers release: JointTransformerBlock with AdaLayerNormZero (chunk order shift_msa,
scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp), AdaLayerNormContinuous for the last block's context stream and for norm_out
(chunk order scale, shift), LayerNorm(eps=1e-6, no affine), joint attention over [image tokens ; text tokens] with 64-wide heads,
FeedForward with tanh-GELU, PatchEmbed(patch 2) + centre-cropped positional table, CombinedTimestepTextProjEmbeddings.
"""
import math
from collections import OrderedDict
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class MMDiTConfig:
    sample_size: int = 128
    patch_size: int = 2
    in_channels: int = 16
    num_layers: int = 24
    attention_head_dim: int = 64
    num_attention_heads: int = 24
    joint_attention_dim: int = 4096
    caption_projection_dim: int = 1536
    pooled_projection_dim: int = 2048
    out_channels: int = 16
    pos_embed_max_size: int = 192

    @property
    def inner_dim(self):
        return self.num_attention_heads * self.attention_head_dim

    @staticmethod
    def sd3_medium():
        return MMDiTConfig()


def param_spec(cfg: MMDiTConfig):
    """(key, shape) of every parameter, diffusers state-dict order; ``pos_embed.pos_embed`` is a persistent buffer (listed last)."""
    D, p = cfg.inner_dim, cfg.patch_size
    out = [("pos_embed.proj.weight", (D, cfg.in_channels, p, p)), ("pos_embed.proj.bias", (D,))]

    def lin(name, n, k):
        out.append((name + ".weight", (n, k)))
        out.append((name + ".bias", (n,)))
    lin("time_text_embed.timestep_embedder.linear_1", D, 256)
    lin("time_text_embed.timestep_embedder.linear_2", D, D)
    lin("time_text_embed.text_embedder.linear_1", D, cfg.pooled_projection_dim)
    lin("time_text_embed.text_embedder.linear_2", D, D)
    lin("context_embedder", cfg.caption_projection_dim, cfg.joint_attention_dim)
    for i in range(cfg.num_layers):
        b = f"transformer_blocks.{i}."
        last = i == cfg.num_layers - 1
        lin(b + "norm1.linear", 6 * D, D)
        lin(b + "norm1_context.linear", (2 if last else 6) * D, D)
        for n in ("to_q", "to_k", "to_v", "add_k_proj", "add_v_proj", "add_q_proj"):
            lin(b + "attn." + n, D, D)
        lin(b + "attn.to_out.0", D, D)
        if not last:
            lin(b + "attn.to_add_out", D, D)
        lin(b + "ff.net.0.proj", 4 * D, D)
        lin(b + "ff.net.2", D, 4 * D)
        if not last:
            lin(b + "ff_context.net.0.proj", 4 * D, D)
            lin(b + "ff_context.net.2", D, 4 * D)
    lin("norm_out.linear", 2 * D, D)
    lin("proj_out", p * p * cfg.out_channels, D)
    return out


def buffer_spec(cfg: MMDiTConfig):
    return [("pos_embed.pos_embed", (1, cfg.pos_embed_max_size ** 2, cfg.inner_dim))]


LORA_SUFFIXES = ("to_k", "to_q", "to_v", "to_out.0", "proj_in", "proj_out", "ff.net.0.proj", "ff.net.2")   # train_pcm_lora_sd3.py:978-987
# train_pcm_lora_sd3_adv.py:992-1015 (verbatim, incl. the three leading-dot entries that peft's suffix rule can never match)
LORA_SUFFIXES_ADV = ("to_k", "to_q", "to_v", ".add_q_proj", ".add_k_proj", ".add_v_proj", "to_add_out", "to_out.0", "proj_in", "proj_out",
                     "ff.net.0.proj", "ff.net.2", "ff_context.net.0.proj", "ff_context.net.2", "norm1.linear", "norm1_context.linear",
                     "context_embedder", "text_embedder.linear_1", "text_embedder.linear_2", "timestep_embedder.linear_1",
                     "timestep_embedder.linear_2", "pos_embed.proj")


def lora_target_modules(cfg: MMDiTConfig, targets=LORA_SUFFIXES):
    """[(module path, weight shape)] in module order -- peft 0.9's rule ``name == t or name.endswith("." + t)``."""
    out = []
    for k, shp in param_spec(cfg):
        if not k.endswith(".weight"):
            continue
        name = k[:-len(".weight")]
        if any(name == t or name.endswith("." + t) for t in targets):
            out.append((name, shp))
    return out


def sincos_pos_embed(cfg: MMDiTConfig):
    """get_2d_sincos_pos_embed(D, pos_embed_max_size, base_size=sample_size // patch_size, interpolation_scale=1): [1, S*S, D]."""
    D, S = cfg.inner_dim, cfg.pos_embed_max_size
    base = cfg.sample_size // cfg.patch_size
    g = torch.arange(S, dtype=torch.float64) / (S / base)
    gw, gh = torch.meshgrid(g, g, indexing="xy")           # np.meshgrid(grid_w, grid_h): w goes first

    def one(pos, dim):
        omega = 1.0 / 10000 ** (torch.arange(dim // 2, dtype=torch.float64) / (dim / 2.0))
        o = pos.reshape(-1)[:, None] * omega[None, :]
        return torch.cat([torch.sin(o), torch.cos(o)], 1)
    emb = torch.cat([one(gh, D // 2), one(gw, D // 2)], 1)
    return emb.float().unsqueeze(0)


def init_state_dict(cfg: MMDiTConfig, seed=0, std=None):
    """deterministic random weights (there is no checkpoint here): N(0, 1/sqrt(fan_in)) matrices, small biases, sincos table."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for k, shp in param_spec(cfg):
        if k.endswith(".bias"):
            sd[k] = torch.randn(shp, generator=g) * 0.02
        else:
            fan_in = math.prod(shp[1:])
            sd[k] = torch.randn(shp, generator=g) * ((std if std is not None else 1.0) / math.sqrt(fan_in))
    sd["pos_embed.pos_embed"] = sincos_pos_embed(cfg)
    return sd


def _lin(sd, name, x, lora=None, alpha=8.0):
    y = F.linear(x, sd[name + ".weight"], sd[name + ".bias"])
    if lora is not None and name in lora:
        A, Bm = lora[name]
        y = y + (alpha / A.shape[0]) * F.linear(F.linear(x, A.reshape(A.shape[0], -1)), Bm.reshape(Bm.shape[0], -1))
    return y


def timestep_proj(t, dim=256):
    """Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0)."""
    half = dim // 2
    f = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    a = t.float()[:, None] * f[None, :]
    return torch.cat([torch.cos(a), torch.sin(a)], -1)


def crop_pos_embed(cfg, sd, hp, wp):
    S = cfg.pos_embed_max_size
    top, left = (S - hp) // 2, (S - wp) // 2
    pe = sd["pos_embed.pos_embed"].reshape(1, S, S, -1)[:, top:top + hp, left:left + wp, :]
    return pe.reshape(1, hp * wp, -1)


def mmdit_forward(cfg: MMDiTConfig, sd, hidden_states, timestep, encoder_hidden_states, pooled_projections, lora=None, lora_alpha=8.0,
                  return_features=False):
    """hidden_states [B, 16, H, W], timestep [B] float, encoder_hidden_states [B, Lc, 4096], pooled [B, 2048] -> [B, 16, H, W]."""
    B, _, H, W = hidden_states.shape
    p, D, nh = cfg.patch_size, cfg.inner_dim, cfg.num_attention_heads
    hp, wp = H // p, W // p
    L = dict(lora=lora, alpha=lora_alpha)
    x = F.conv2d(hidden_states, sd["pos_embed.proj.weight"], sd["pos_embed.proj.bias"], stride=p)
    if lora is not None and "pos_embed.proj" in lora:      # peft Conv2d LoRA: lora_A = Conv2d(16, r, k=p, s=p), lora_B = Conv2d(r, D, 1)
        A, Bm = lora["pos_embed.proj"]
        r = A.shape[0]
        x = x + (lora_alpha / r) * F.conv2d(F.conv2d(hidden_states, A.reshape(r, cfg.in_channels, p, p), stride=p), Bm.reshape(-1, r, 1, 1))
    x = x.flatten(2).transpose(1, 2)
    x = x + crop_pos_embed(cfg, sd, hp, wp)
    te = _lin(sd, "time_text_embed.timestep_embedder.linear_2", F.silu(_lin(sd, "time_text_embed.timestep_embedder.linear_1", timestep_proj(timestep), **L)), **L)
    pe = _lin(sd, "time_text_embed.text_embedder.linear_2", F.silu(_lin(sd, "time_text_embed.text_embedder.linear_1", pooled_projections, **L)), **L)
    temb = te + pe
    c = _lin(sd, "context_embedder", encoder_hidden_states, **L)
    semb = F.silu(temb)
    feats = []

    def ln(v):
        return F.layer_norm(v, (D,), eps=1e-6)
    for i in range(cfg.num_layers):
        b = f"transformer_blocks.{i}."
        last = i == cfg.num_layers - 1
        sh_a, sc_a, g_a, sh_m, sc_m, g_m = _lin(sd, b + "norm1.linear", semb, **L).chunk(6, dim=1)
        xn = ln(x) * (1 + sc_a[:, None]) + sh_a[:, None]
        if last:
            csc, csh = _lin(sd, b + "norm1_context.linear", semb, **L).chunk(2, dim=1)
            cn = ln(c) * (1 + csc[:, None]) + csh[:, None]
        else:
            csh_a, csc_a, cg_a, csh_m, csc_m, cg_m = _lin(sd, b + "norm1_context.linear", semb, **L).chunk(6, dim=1)
            cn = ln(c) * (1 + csc_a[:, None]) + csh_a[:, None]
        q = torch.cat([_lin(sd, b + "attn.to_q", xn, **L), _lin(sd, b + "attn.add_q_proj", cn, **L)], 1)
        k = torch.cat([_lin(sd, b + "attn.to_k", xn, **L), _lin(sd, b + "attn.add_k_proj", cn, **L)], 1)
        v = torch.cat([_lin(sd, b + "attn.to_v", xn, **L), _lin(sd, b + "attn.add_v_proj", cn, **L)], 1)

        def heads(t):
            return t.view(B, -1, nh, cfg.attention_head_dim).transpose(1, 2)
        o = F.scaled_dot_product_attention(heads(q), heads(k), heads(v)).transpose(1, 2).reshape(B, -1, D)
        Lx = x.shape[1]
        ox, oc = o[:, :Lx], o[:, Lx:]
        x = x + g_a[:, None] * _lin(sd, b + "attn.to_out.0", ox, **L)
        xn2 = ln(x) * (1 + sc_m[:, None]) + sh_m[:, None]
        ff = _lin(sd, b + "ff.net.2", F.gelu(_lin(sd, b + "ff.net.0.proj", xn2, **L), approximate="tanh"), **L)
        x = x + g_m[:, None] * ff
        if not last:
            c = c + cg_a[:, None] * _lin(sd, b + "attn.to_add_out", oc, **L)
            cn2 = ln(c) * (1 + csc_m[:, None]) + csh_m[:, None]
            c = c + cg_m[:, None] * _lin(sd, b + "ff_context.net.2", F.gelu(_lin(sd, b + "ff_context.net.0.proj", cn2, **L), approximate="tanh"), **L)
        feats.append(x)
    sc, sh = _lin(sd, "norm_out.linear", semb).chunk(2, dim=1)
    x = ln(x) * (1 + sc[:, None]) + sh[:, None]
    x = _lin(sd, "proj_out", x, **L)
    x = x.reshape(B, hp, wp, p, p, cfg.out_channels)
    x = torch.einsum("nhwpqc->nchpwq", x).reshape(B, cfg.out_channels, hp * p, wp * p)
    return (x, feats) if return_features else x
