"""SD1.5 ``UNet2DConditionModel`` topology and parameter naming (diffusers 0.26.3 key names), so a
diffusers ``unet/diffusion_pytorch_model.safetensors`` loads unchanged and LoRA checkpoints written
here load in diffusers / peft / the reference's demo.

Reference wiring witness: /root/reference/code/text_to_image_sd15/discriminator_sd15.py:84-345
(a verbatim copy of diffusers' forward); LoRA target list: train_pcm_lora_sd15.py:868-883.
"""
import math
from collections import OrderedDict

import torch

LORA_TARGETS = ("to_q", "to_k", "to_v", "to_out.0", "proj_in", "proj_out", "ff.net.0.proj", "ff.net.2",
                "conv1", "conv2", "conv_shortcut", "downsamplers.0.conv", "upsamplers.0.conv", "time_emb_proj")


class UNetConfig:
    """``UNet2DConditionModel`` topology.  Defaults = SD1.5; ``sdxl()`` = stabilityai/stable-diffusion-xl-base-1.0 (SURVEY §8f rank 3:
    no attention at the first level, transformer depth (1, 2, 10), 64-wide heads, linear proj_in/proj_out, text_time added conditioning)."""

    def __init__(self, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                 cross_attention_dim=768, heads=8, norm_num_groups=32, norm_eps=1e-5, down_attn=None, transformer_depth=None,
                 mid_depth=None, use_linear_projection=False, addition_time_embed_dim=None, projection_class_embeddings_input_dim=None):
        self.in_channels, self.out_channels = in_channels, out_channels
        self.block_out_channels = tuple(block_out_channels)
        n = len(self.block_out_channels)
        self.layers_per_block = layers_per_block
        self.cross_attention_dim = cross_attention_dim
        self.heads = heads  # unet/config.json `attention_head_dim` is the head COUNT (SD1.5: 8; SDXL: (5, 10, 20) per level)
        self.norm_num_groups, self.norm_eps = norm_num_groups, norm_eps
        self.time_embed_dim = 4 * block_out_channels[0]
        self.down_attn = tuple(down_attn) if down_attn is not None else tuple(i < n - 1 for i in range(n))
        self.transformer_depth = tuple(transformer_depth) if transformer_depth is not None else (1,) * n
        self.mid_depth = mid_depth if mid_depth is not None else self.transformer_depth[-1]
        self.use_linear_projection = use_linear_projection
        self.addition_time_embed_dim = addition_time_embed_dim
        self.projection_class_embeddings_input_dim = projection_class_embeddings_input_dim

    def heads_at(self, level):
        return self.heads if isinstance(self.heads, int) else self.heads[level]

    def up_level(self, i):
        """up block i works at the resolution level of down block n-1-i."""
        return len(self.block_out_channels) - 1 - i

    def heads_of_path(self, path):
        """head count of the attention a parameter path (``down_blocks.1.attentions.0. ... attn1.to_q``) belongs to"""
        parts = path.split(".")
        if parts[0] == "down_blocks":
            return self.heads_at(int(parts[1]))
        if parts[0] == "up_blocks":
            return self.heads_at(self.up_level(int(parts[1])))
        return self.heads_at(len(self.block_out_channels) - 1)            # mid block

    def q_scale_of_path(self, path, n_out):
        """What the UNet's attention kernels expect folded into a ``to_q`` projection (csrc/attention_ps.hip): softmax scale x log2(e) for
        the head dim n_out / heads; None for every other path."""
        if not path.endswith(".to_q"):
            return None
        d = n_out // self.heads_of_path(path)
        return d ** -0.5 * 1.4426950408889634

    def up_attn(self, i):
        return self.down_attn[self.up_level(i)]

    @staticmethod
    def sd15():
        return UNetConfig()

    @staticmethod
    def sdxl():
        return UNetConfig(block_out_channels=(320, 640, 1280), cross_attention_dim=2048, heads=(5, 10, 20), down_attn=(False, True, True),
                          transformer_depth=(1, 2, 10), use_linear_projection=True, addition_time_embed_dim=256,
                          projection_class_embeddings_input_dim=2816)


def _resnet(p, cin, cout, temb):
    s = [(p + "norm1.weight", (cin,)), (p + "norm1.bias", (cin,)), (p + "conv1.weight", (cout, cin, 3, 3)),
         (p + "conv1.bias", (cout,)), (p + "time_emb_proj.weight", (cout, temb)), (p + "time_emb_proj.bias", (cout,)),
         (p + "norm2.weight", (cout,)), (p + "norm2.bias", (cout,)), (p + "conv2.weight", (cout, cout, 3, 3)),
         (p + "conv2.bias", (cout,))]
    if cin != cout:
        s += [(p + "conv_shortcut.weight", (cout, cin, 1, 1)), (p + "conv_shortcut.bias", (cout,))]
    return s


def _attn(p, c, ctx, depth=1, linear=False):
    pw = (c, c) if linear else (c, c, 1, 1)
    s = [(p + "norm.weight", (c,)), (p + "norm.bias", (c,)), (p + "proj_in.weight", pw), (p + "proj_in.bias", (c,))]
    for k in range(depth):
        b = p + f"transformer_blocks.{k}."
        s += [(b + "norm1.weight", (c,)), (b + "norm1.bias", (c,)), (b + "attn1.to_q.weight", (c, c)),
              (b + "attn1.to_k.weight", (c, c)), (b + "attn1.to_v.weight", (c, c)), (b + "attn1.to_out.0.weight", (c, c)),
              (b + "attn1.to_out.0.bias", (c,)), (b + "norm2.weight", (c,)), (b + "norm2.bias", (c,)),
              (b + "attn2.to_q.weight", (c, c)), (b + "attn2.to_k.weight", (c, ctx)), (b + "attn2.to_v.weight", (c, ctx)),
              (b + "attn2.to_out.0.weight", (c, c)), (b + "attn2.to_out.0.bias", (c,)), (b + "norm3.weight", (c,)),
              (b + "norm3.bias", (c,)), (b + "ff.net.0.proj.weight", (8 * c, c)), (b + "ff.net.0.proj.bias", (8 * c,)),
              (b + "ff.net.2.weight", (c, 4 * c)), (b + "ff.net.2.bias", (c,))]
    return s + [(p + "proj_out.weight", pw), (p + "proj_out.bias", (c,))]


def up_resnet_in_channels(cfg):
    boc = cfg.block_out_channels
    n = len(boc)
    rev = list(reversed(boc))
    rows, prev = [], rev[0]
    for i in range(n):
        out, inp = rev[i], rev[min(i + 1, n - 1)]
        rows.append([(prev if j == 0 else out) + (inp if j == cfg.layers_per_block else out)
                     for j in range(cfg.layers_per_block + 1)])
        prev = out
    return rows


def param_spec(cfg):
    boc, temb, ctx, n = cfg.block_out_channels, cfg.time_embed_dim, cfg.cross_attention_dim, len(cfg.block_out_channels)
    s = [("conv_in.weight", (boc[0], cfg.in_channels, 3, 3)), ("conv_in.bias", (boc[0],)),
         ("time_embedding.linear_1.weight", (temb, boc[0])), ("time_embedding.linear_1.bias", (temb,)),
         ("time_embedding.linear_2.weight", (temb, temb)), ("time_embedding.linear_2.bias", (temb,))]
    if cfg.addition_time_embed_dim:
        pin = cfg.projection_class_embeddings_input_dim
        s += [("add_embedding.linear_1.weight", (temb, pin)), ("add_embedding.linear_1.bias", (temb,)),
              ("add_embedding.linear_2.weight", (temb, temb)), ("add_embedding.linear_2.bias", (temb,))]
    lin = cfg.use_linear_projection
    cin = boc[0]
    for i in range(n):
        cout = boc[i]
        for j in range(cfg.layers_per_block):
            s += _resnet(f"down_blocks.{i}.resnets.{j}.", cin if j == 0 else cout, cout, temb)
        if cfg.down_attn[i]:
            for j in range(cfg.layers_per_block):
                s += _attn(f"down_blocks.{i}.attentions.{j}.", cout, ctx, cfg.transformer_depth[i], lin)
        if i < n - 1:
            s += [(f"down_blocks.{i}.downsamplers.0.conv.weight", (cout, cout, 3, 3)),
                  (f"down_blocks.{i}.downsamplers.0.conv.bias", (cout,))]
        cin = cout
    c = boc[-1]
    s += _resnet("mid_block.resnets.0.", c, c, temb) + _attn("mid_block.attentions.0.", c, ctx, cfg.mid_depth, lin) + _resnet("mid_block.resnets.1.", c, c, temb)
    rin, rev = up_resnet_in_channels(cfg), list(reversed(boc))
    for i in range(n):
        cout = rev[i]
        for j in range(cfg.layers_per_block + 1):
            s += _resnet(f"up_blocks.{i}.resnets.{j}.", rin[i][j], cout, temb)
        if cfg.up_attn(i):
            for j in range(cfg.layers_per_block + 1):
                s += _attn(f"up_blocks.{i}.attentions.{j}.", cout, ctx, cfg.transformer_depth[cfg.up_level(i)], lin)
        if i < n - 1:
            s += [(f"up_blocks.{i}.upsamplers.0.conv.weight", (cout, cout, 3, 3)), (f"up_blocks.{i}.upsamplers.0.conv.bias", (cout,))]
    s += [("conv_norm_out.weight", (boc[0],)), ("conv_norm_out.bias", (boc[0],)),
          ("conv_out.weight", (cfg.out_channels, boc[0], 3, 3)), ("conv_out.bias", (cfg.out_channels,))]
    return s


def lora_target_modules(cfg):
    """[(module path, base weight shape)] matched by peft's ``key == t or key.endswith('.'+t)`` rule."""
    out = []
    for k, shp in param_spec(cfg):
        if k.endswith(".weight"):
            path = k[:-7]
            if any(path == t or path.endswith("." + t) for t in LORA_TARGETS):
                out.append((path, shp))
    return out


def random_state_dict(cfg, seed=0, device="cpu"):
    """Seeded stand-in for the (offline-unavailable) SD1.5 checkpoint: PyTorch-default layer init.
    device="cpu" reproduces the oracle's init bit for bit; a GPU device draws on-device (bench)."""
    g = torch.Generator(device=device).manual_seed(seed)
    spec = param_spec(cfg)
    shapes = dict(spec)
    sd = OrderedDict()
    for k, shp in spec:
        leaf = k.rsplit(".", 1)[0].rsplit(".", 1)[-1]
        if leaf.startswith("norm") or leaf == "conv_norm_out":
            sd[k] = torch.ones(shp, device=device) if k.endswith("weight") else torch.zeros(shp, device=device)
            continue
        w = shapes[k.rsplit(".", 1)[0] + ".weight"]
        bound = 1.0 / math.sqrt(math.prod(w[1:]))
        sd[k] = (torch.rand(shp, generator=g, device=device) * 2 - 1) * bound
    return sd
