"""Algorithmic work of the distillation step from a walk over the model configuration (the counts BASELINE.md section 2 / SURVEY section 8d
quote for SD1.5, generalised to the SDXL UNet and the SD3 MMDiT).  2 FLOP per MAC.  Used by bench.py for ``roofline`` and by
tests/test_flops.py, which pins the SD1.5 numbers of the survey (401.64 GMAC base forward, 47.16 GMAC LoRA, 339.8 GMAC heads).

Conventions: forward = every contraction of the module tree + the attention cores (QK^T and PV); LoRA r adds
r*(K + N) MACs per row of a wrapped layer; backward = base dgrad (= base forward) + one more attention core (recompute excluded)
+ LoRA dgrad and wgrad (2 x the LoRA forward extra); step = 2 student forwards + 2 teacher forwards + backward."""
import math

from .unet_spec import lora_target_modules, param_spec


def _unet_rows(cfg, path, H, W, ctx_len):
    """rows (per sample) the contraction at ``path`` runs on"""
    n = len(cfg.block_out_channels)
    parts = path.split(".")
    if parts[0] in ("time_embedding", "add_embedding") or parts[-1] == "time_emb_proj":
        return 1
    if path.endswith("attn2.to_k") or path.endswith("attn2.to_v"):
        return ctx_len
    if parts[0] == "conv_in" or parts[0] == "conv_out":
        return H * W
    if parts[0] == "mid_block":
        lvl = n - 1
    elif parts[0] == "down_blocks":
        lvl = int(parts[1])
        if parts[2] == "downsamplers":
            lvl += 1
    else:   # up_blocks.i runs at level n-1-i; its upsampler conv runs on the 2x image (level n-2-i)
        lvl = n - 1 - int(parts[1])
        if parts[2] == "upsamplers":
            lvl -= 1
    return (H >> lvl) * (W >> lvl)


def unet_macs(cfg, H=64, W=64, ctx_len=77, lora_rank=64):
    """per-sample MACs of one UNet forward: dict(base, lora, attn_core, by_kind)"""
    spec = dict(param_spec(cfg))
    lora = {p for p, _ in lora_target_modules(cfg)} if lora_rank else set()
    base = lo = 0
    kinds = {}
    for k, shp in spec.items():
        if not k.endswith(".weight") or len(shp) < 2:
            continue
        path = k[:-7]
        rows = _unet_rows(cfg, path, H, W, ctx_len)
        kin = math.prod(shp[1:])
        m = rows * shp[0] * kin
        base += m
        leaf = path.rsplit(".", 1)[-1] if not path.endswith("to_out.0") else "to_out.0"
        kind = ("conv3x3" if len(shp) == 4 and shp[-1] == 3 else "linear/1x1") + ":" + leaf
        kinds[kind] = kinds.get(kind, 0) + m
        if path in lora:
            lo += rows * lora_rank * (kin + shp[0])
    # attention cores: one self + one cross attention per transformer block
    core = 0
    n = len(cfg.block_out_channels)

    def blocks():
        for i in range(n):
            if cfg.down_attn[i]:
                yield i, cfg.layers_per_block * cfg.transformer_depth[i]
        yield n - 1, cfg.mid_depth
        for i in range(n):
            if cfg.up_attn(i):
                lv = cfg.up_level(i)
                yield lv, (cfg.layers_per_block + 1) * cfg.transformer_depth[lv]
    for lvl, count in blocks():
        L, C = (H >> lvl) * (W >> lvl), cfg.block_out_channels[lvl]
        core += count * (2 * L * L * C + 2 * L * ctx_len * C)
    return dict(base=base + core, contractions=base, attn_core=core, lora=lo, by_kind=kinds)


def mmdit_macs(cfg, hw=128, ctx_len=154, lora_rank=32, lora_targets=None):
    """per-sample MACs of one SD3 MMDiT forward (joint attention over hw/patch squared image tokens + ctx_len text tokens)"""
    from . import mmdit_spec as S
    spec = dict(S.param_spec(cfg))
    lora = {p for p, _ in S.lora_target_modules(cfg, lora_targets or S.LORA_TARGETS_SD3)} if lora_rank else set()
    Li = (hw // cfg.patch_size) ** 2
    D = cfg.num_attention_heads * cfg.attention_head_dim
    base = lo = 0
    for k, shp in spec.items():
        if not k.endswith(".weight") or len(shp) < 2:
            continue
        path = k[:-7]
        leaf = path.split(".")
        if path.startswith("time_text_embed") or ".norm1.linear" in path or ".norm1_context.linear" in path or path.startswith("norm_out"):
            rows = 1
        elif "add_" in leaf[-1] or "to_add_out" in path or "ff_context" in path or path.startswith("context_embedder"):
            rows = ctx_len
        else:
            rows = Li
        kin = math.prod(shp[1:])
        base += rows * shp[0] * kin
        if path in lora:
            lo += rows * lora_rank * (kin + shp[0])
    Lt = Li + ctx_len
    core = cfg.num_layers * 2 * Lt * Lt * D
    return dict(base=base + core, contractions=base, attn_core=core, lora=lo)


def heads_macs(dims, hw, nh=4, ksize=3):
    """per-sample MACs of one forward of the discriminator heads (two k x k convs + the 1x1 logit conv per head)"""
    return sum(nh * h * h * (2 * ksize * ksize * C * C + C) for C, h in zip(dims, hw))


def step_tflop(m):
    """per-sample TFLOP of one distillation step from a forward walk: (student fwd, teacher fwd, backward, step)"""
    stu = 2e-12 * (m["base"] + m["lora"])
    tea = 2e-12 * m["base"]
    bwd = 2e-12 * (m["base"] + m["attn_core"] + 2 * m["lora"])
    return dict(student_fwd=stu, teacher_fwd=tea, backward=bwd, step=2 * stu + 2 * tea + bwd)
