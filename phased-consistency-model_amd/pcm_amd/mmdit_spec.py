"""``SD3Transformer2DModel`` (MMDiT) topology and parameter naming (diffusers key names), so a diffusers
``transformer/diffusion_pytorch_model.safetensors`` loads unchanged and the LoRA checkpoints written here load in peft / diffusers.

Reference wiring witness: /root/reference/code/text_to_image_sd3/discriminator_sd3.py:73-137 (the reference's copy of the forward);
LoRA target list: train_pcm_lora_sd3.py:975-988 (r = --lora_rank, run.sh: 32; init_lora_weights="gaussian").
The enumeration reproduces the published SD3-medium parameter count, 2 028 328 000.
"""
import math
from collections import OrderedDict

import torch

LORA_TARGETS_SD3 = ("to_k", "to_q", "to_v", "to_out.0", "proj_in", "proj_out", "ff.net.0.proj", "ff.net.2")
# the adversarial trainers' list (train_pcm_lora_sd3_adv.py:992-1015, 22 entries; the stochastic script drops "pos_embed.proj").
# peft 0.9 matches ``name == t or name.endswith("." + t)``: the three entries written with a LEADING DOT (".add_q_proj", ...) can
# never match, so add_q/k/v_proj stay unadapted -- kept verbatim, the quirk is part of the behaviour.
LORA_TARGETS_SD3_ADV = ("to_k", "to_q", "to_v", ".add_q_proj", ".add_k_proj", ".add_v_proj", "to_add_out", "to_out.0", "proj_in", "proj_out",
                        "ff.net.0.proj", "ff.net.2", "ff_context.net.0.proj", "ff_context.net.2", "norm1.linear", "norm1_context.linear",
                        "context_embedder", "text_embedder.linear_1", "text_embedder.linear_2", "timestep_embedder.linear_1",
                        "timestep_embedder.linear_2", "pos_embed.proj")


class MMDiTConfig:
    def __init__(self, sample_size=128, patch_size=2, in_channels=16, num_layers=24, attention_head_dim=64, num_attention_heads=24,
                 joint_attention_dim=4096, caption_projection_dim=1536, pooled_projection_dim=2048, out_channels=16, pos_embed_max_size=192):
        if patch_size != 2:
            raise ValueError("pcm_amd.mmdit: patch_size 2 (the SD3 configuration) is the one built")
        self.sample_size, self.patch_size, self.in_channels, self.num_layers = sample_size, patch_size, in_channels, num_layers
        self.attention_head_dim, self.num_attention_heads = attention_head_dim, num_attention_heads
        self.joint_attention_dim, self.caption_projection_dim = joint_attention_dim, caption_projection_dim
        self.pooled_projection_dim, self.out_channels, self.pos_embed_max_size = pooled_projection_dim, out_channels, pos_embed_max_size
        self.inner_dim = num_attention_heads * attention_head_dim
        if caption_projection_dim != self.inner_dim:
            raise ValueError("pcm_amd.mmdit: caption_projection_dim must equal heads * head_dim (joint attention shares the width)")

    @staticmethod
    def sd3_medium():
        return MMDiTConfig()


def param_spec(cfg: MMDiTConfig):
    """(key, shape), diffusers state-dict order."""
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
        lin(b + "norm1_context.linear", (2 if last else 6) * D, D)      # last block: context_pre_only (AdaLayerNormContinuous)
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
    """persistent buffers of the checkpoint."""
    return [("pos_embed.pos_embed", (1, cfg.pos_embed_max_size ** 2, cfg.inner_dim))]


def lora_target_modules(cfg: MMDiTConfig, targets=LORA_TARGETS_SD3):
    """[(module path, weight shape)] -- peft's rule ``name == t or name.endswith('.' + t)`` over the Linear / Conv2d modules.
    Default list (train_pcm_lora_sd3.py:978-987): the image stream's to_q/k/v/to_out.0 and ff of every block plus the final proj_out
    (145 modules at 24 layers).  ``LORA_TARGETS_SD3_ADV``: additionally the context stream's to_add_out / ff_context, the adaLN
    projections norm1(.context).linear, context_embedder, the four time/text embedder linears and the patch-embedding conv."""
    out = []
    for k, shp in param_spec(cfg):
        if not k.endswith(".weight"):
            continue
        name = k[:-len(".weight")]
        if any(name == t or name.endswith("." + t) for t in targets):
            out.append((name, shp))
    return out


def sincos_pos_embed(cfg: MMDiTConfig):
    """PatchEmbed's fixed table: 2-D sin/cos over a pos_embed_max_size grid scaled to base_size = sample_size // patch_size."""
    D, S = cfg.inner_dim, cfg.pos_embed_max_size
    base = cfg.sample_size // cfg.patch_size
    g = torch.arange(S, dtype=torch.float64) / (S / base)
    gw, gh = torch.meshgrid(g, g, indexing="xy")

    def one(pos, dim):
        omega = 1.0 / 10000 ** (torch.arange(dim // 2, dtype=torch.float64) / (dim / 2.0))
        o = pos.reshape(-1)[:, None] * omega[None, :]
        return torch.cat([torch.sin(o), torch.cos(o)], 1)
    return torch.cat([one(gh, D // 2), one(gw, D // 2)], 1).float().unsqueeze(0)


def random_state_dict(cfg: MMDiTConfig, seed=0, device="cpu", std=None):
    """Seeded stand-in for the (offline-unavailable) SD3 checkpoint: N(0, 1/fan_in) matrices, small biases, the sincos table."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd = OrderedDict()
    for k, shp in param_spec(cfg):
        if k.endswith(".bias"):
            sd[k] = torch.randn(shp, generator=g, device=device) * 0.02
        else:
            sd[k] = torch.randn(shp, generator=g, device=device) * ((std if std is not None else 1.0) / math.sqrt(math.prod(shp[1:])))
    sd["pos_embed.pos_embed"] = sincos_pos_embed(cfg).to(device)
    return sd
