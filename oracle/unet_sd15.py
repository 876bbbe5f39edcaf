"""Plain-torch fp32 restatement of diffusers-0.26.3 ``UNet2DConditionModel`` (SD1.5 config)
with peft-0.9.0 LoRA semantics.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference does not vendor diffusers/peft (environment.yaml:40,:87) and neither is
installable here, so this file restates the published module semantics; the only in-repo
witness of the wiring is the copied forward at
/root/reference/code/text_to_image_sd15/discriminator_sd15.py:84-345 (time-emb :122-141,
conv_in :248, down loop :268-290, mid :293-309, up loop :312-342) which this follows.
Anchors: 686 tensors / 859 520 964 parameters; LoRA r=64 on the 14 target patterns of
train_pcm_lora_sd15.py:868-883 -> 278 wrapped modules / 67 252 224 parameters.
**parity unpinned** (no reference outputs exist for this part).

Functional style: ``unet_forward(sd, x, t, ctx, lora=None)`` takes a flat state dict with
diffusers key names and an optional ``{module_path: (A, B)}`` LoRA dict (peft layouts:
Linear A [r,in], B [out,r]; Conv2d A [r,in,k,k], B [out,r,1,1]).
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

LORA_TARGETS = ["to_q", "to_k", "to_v", "to_out.0", "proj_in", "proj_out", "ff.net.0.proj",
                "ff.net.2", "conv1", "conv2", "conv_shortcut", "downsamplers.0.conv",
                "upsamplers.0.conv", "time_emb_proj"]  # train_pcm_lora_sd15.py:868-883


class UNetConfig:
    """SD1.5 unet/config.json semantics.  ``tiny()`` keeps the topology and
    shrinks widths so CPU tests finish in seconds."""

    def __init__(self, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                 layers_per_block=2, cross_attention_dim=768, heads=8, norm_num_groups=32,
                 norm_eps=1e-5, temb_mult=4, down_attn=None, transformer_depth=None, mid_depth=None,
                 use_linear_projection=False, addition_time_embed_dim=None, projection_class_embeddings_input_dim=None):
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.block_out_channels = tuple(block_out_channels)
        self.layers_per_block = layers_per_block
        self.cross_attention_dim = cross_attention_dim
        self.heads = heads  # config key `attention_head_dim: 8` is used as the head COUNT
        self.norm_num_groups = norm_num_groups
        self.norm_eps = norm_eps
        self.time_embed_dim = block_out_channels[0] * temb_mult
        # SDXL-style topology switches (restated from the pinned diffusers UNet2DConditionModel: the
        # down_block_types / transformer_layers_per_block / attention_head_dim / addition_embed_type="text_time" config keys)
        n = len(self.block_out_channels)
        self.down_attn = tuple(down_attn) if down_attn is not None else tuple(i < n - 1 for i in range(n))
        self.transformer_depth = tuple(transformer_depth) if transformer_depth is not None else (1,) * n
        self.mid_depth = mid_depth if mid_depth is not None else self.transformer_depth[-1]
        self.use_linear_projection = use_linear_projection
        self.addition_time_embed_dim = addition_time_embed_dim
        self.projection_class_embeddings_input_dim = projection_class_embeddings_input_dim

    def heads_at(self, level):
        return self.heads if isinstance(self.heads, int) else self.heads[level]

    @staticmethod
    def sd15():
        return UNetConfig()

    @staticmethod
    def sdxl():
        """stabilityai/stable-diffusion-xl-base-1.0 unet/config.json (2 567 463 684 parameters)."""
        return UNetConfig(block_out_channels=(320, 640, 1280), cross_attention_dim=2048, heads=(5, 10, 20),
                          down_attn=(False, True, True), transformer_depth=(1, 2, 10), use_linear_projection=True,
                          addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816)

    @staticmethod
    def tiny(c=32, ctx=64, heads=2, groups=8):
        return UNetConfig(block_out_channels=(c, 2 * c, 4 * c, 4 * c), cross_attention_dim=ctx,
                          heads=heads, norm_num_groups=groups)


# ----------------------------------------------------------------------------- parameter spec
def _resnet_spec(p, cin, cout, temb):
    s = [(p + "norm1.weight", (cin,)), (p + "norm1.bias", (cin,)),
         (p + "conv1.weight", (cout, cin, 3, 3)), (p + "conv1.bias", (cout,)),
         (p + "time_emb_proj.weight", (cout, temb)), (p + "time_emb_proj.bias", (cout,)),
         (p + "norm2.weight", (cout,)), (p + "norm2.bias", (cout,)),
         (p + "conv2.weight", (cout, cout, 3, 3)), (p + "conv2.bias", (cout,))]
    if cin != cout:
        s += [(p + "conv_shortcut.weight", (cout, cin, 1, 1)), (p + "conv_shortcut.bias", (cout,))]
    return s


def _attn_spec(p, c, ctx, depth=1, linear=False):
    pw = (c, c) if linear else (c, c, 1, 1)
    s = [(p + "norm.weight", (c,)), (p + "norm.bias", (c,)),
         (p + "proj_in.weight", pw), (p + "proj_in.bias", (c,))]
    for k in range(depth):
        b = p + f"transformer_blocks.{k}."
        s += [(b + "norm1.weight", (c,)), (b + "norm1.bias", (c,)),
              (b + "attn1.to_q.weight", (c, c)), (b + "attn1.to_k.weight", (c, c)),
              (b + "attn1.to_v.weight", (c, c)),
              (b + "attn1.to_out.0.weight", (c, c)), (b + "attn1.to_out.0.bias", (c,)),
              (b + "norm2.weight", (c,)), (b + "norm2.bias", (c,)),
              (b + "attn2.to_q.weight", (c, c)), (b + "attn2.to_k.weight", (c, ctx)),
              (b + "attn2.to_v.weight", (c, ctx)),
              (b + "attn2.to_out.0.weight", (c, c)), (b + "attn2.to_out.0.bias", (c,)),
              (b + "norm3.weight", (c,)), (b + "norm3.bias", (c,)),
              (b + "ff.net.0.proj.weight", (8 * c, c)), (b + "ff.net.0.proj.bias", (8 * c,)),
              (b + "ff.net.2.weight", (c, 4 * c)), (b + "ff.net.2.bias", (c,))]
    s += [(p + "proj_out.weight", pw), (p + "proj_out.bias", (c,))]
    return s


def up_resnet_in_channels(cfg):
    """Input channel count of every up-block resnet: cat([h, skip])."""
    boc = cfg.block_out_channels
    n = len(boc)
    rev = list(reversed(boc))
    res = []
    prev_out = rev[0]
    for i in range(n):
        out = rev[i]
        inp = rev[min(i + 1, n - 1)]
        row = []
        for j in range(cfg.layers_per_block + 1):
            skip = inp if j == cfg.layers_per_block else out
            rin = prev_out if j == 0 else out
            row.append(rin + skip)
        res.append(row)
        prev_out = out
    return res


def param_spec(cfg):
    """Ordered [(diffusers key, shape)] for the whole UNet."""
    boc = cfg.block_out_channels
    temb = cfg.time_embed_dim
    ctx = cfg.cross_attention_dim
    n = len(boc)
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
        has_attn = cfg.down_attn[i]
        for j in range(cfg.layers_per_block):
            s += _resnet_spec(f"down_blocks.{i}.resnets.{j}.", cin if j == 0 else cout, cout, temb)
        if has_attn:
            for j in range(cfg.layers_per_block):
                s += _attn_spec(f"down_blocks.{i}.attentions.{j}.", cout, ctx, cfg.transformer_depth[i], lin)
        if i < n - 1:
            s += [(f"down_blocks.{i}.downsamplers.0.conv.weight", (cout, cout, 3, 3)),
                  (f"down_blocks.{i}.downsamplers.0.conv.bias", (cout,))]
        cin = cout
    c = boc[-1]
    s += _resnet_spec("mid_block.resnets.0.", c, c, temb)
    s += _attn_spec("mid_block.attentions.0.", c, ctx, cfg.mid_depth, lin)
    s += _resnet_spec("mid_block.resnets.1.", c, c, temb)
    rin = up_resnet_in_channels(cfg)
    rev = list(reversed(boc))
    for i in range(n):
        cout = rev[i]
        has_attn = cfg.down_attn[n - 1 - i]
        for j in range(cfg.layers_per_block + 1):
            s += _resnet_spec(f"up_blocks.{i}.resnets.{j}.", rin[i][j], cout, temb)
        if has_attn:
            for j in range(cfg.layers_per_block + 1):
                s += _attn_spec(f"up_blocks.{i}.attentions.{j}.", cout, ctx, cfg.transformer_depth[n - 1 - i], lin)
        if i < n - 1:
            s += [(f"up_blocks.{i}.upsamplers.0.conv.weight", (cout, cout, 3, 3)),
                  (f"up_blocks.{i}.upsamplers.0.conv.bias", (cout,))]
    s += [("conv_norm_out.weight", (boc[0],)), ("conv_norm_out.bias", (boc[0],)),
          ("conv_out.weight", (cfg.out_channels, boc[0], 3, 3)), ("conv_out.bias", (cfg.out_channels,))]
    return s


def init_state_dict(cfg, seed=0, dtype=torch.float32):
    """Seeded random init standing in for the (unavailable) SD1.5 checkpoint: PyTorch-default
    layer init (kaiming-uniform(a=sqrt5) weights, U(+-1/sqrt(fan_in)) biases, norms 1/0).
    Deterministic across machines for a given torch build (CPU generator)."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    spec = param_spec(cfg)
    shapes = dict(spec)
    for k, shp in spec:
        leaf = k.rsplit(".", 1)[0].rsplit(".", 1)[-1]
        if leaf.startswith("norm") or leaf == "conv_norm_out":
            sd[k] = torch.ones(shp, dtype=dtype) if k.endswith("weight") else torch.zeros(shp, dtype=dtype)
            continue
        wshape = shapes[k.rsplit(".", 1)[0] + ".weight"]
        fan_in = 1
        for d in wshape[1:]:
            fan_in *= d
        bound = 1.0 / math.sqrt(fan_in)
        sd[k] = ((torch.rand(shp, generator=g, dtype=torch.float32) * 2 - 1) * bound).to(dtype)
    return sd


def lora_target_modules(cfg):
    """[(module path, weight shape)] of every module matched by the peft rule
    ``key == t or key.endswith('.' + t)`` for the reference's 14 targets."""
    out = []
    for k, shp in param_spec(cfg):
        if not k.endswith(".weight"):
            continue
        path = k[:-len(".weight")]
        if any(path == t or path.endswith("." + t) for t in LORA_TARGETS):
            out.append((path, shp))
    return out


def init_lora(cfg, rank=64, seed=1, b_std=0.0):
    """peft-0.9 init: A kaiming_uniform(a=sqrt5), B zeros (b_std>0 -> N(0,b_std) so that the
    LoRA branch is exercised in kernel-parity tests)."""
    g = torch.Generator().manual_seed(seed)
    lora = OrderedDict()
    for path, shp in lora_target_modules(cfg):
        if len(shp) == 4:
            a_shape = (rank, shp[1], shp[2], shp[3])
            b_shape = (shp[0], rank, 1, 1)
        else:
            a_shape = (rank, shp[1])
            b_shape = (shp[0], rank)
        fan_in = 1
        for d in a_shape[1:]:
            fan_in *= d
        bound = 1.0 / math.sqrt(fan_in)  # kaiming_uniform(a=sqrt5) == U(+-1/sqrt(fan_in))
        A = (torch.rand(a_shape, generator=g) * 2 - 1) * bound
        B = torch.randn(b_shape, generator=g) * b_std if b_std > 0 else torch.zeros(b_shape)
        lora[path] = (A, B)
    return lora


# ----------------------------------------------------------------------------- forward
def timestep_embedding(t, dim):
    """diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)  (SURVEY B.3)."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32) / half
    arg = t.float()[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


class _RoundBF16(torch.autograd.Function):
    """x -> bf16 -> fp32 (a STORE in bf16).  Backward: the cotangent passes straight through and is itself rounded -- the HIP path
    keeps the gradient of a bf16-stored activation in bf16 as well."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


_WCACHE = {}


def _derived(w, rounded, dtype):
    """(bf16-rounded and/or dtype-converted) copy of a frozen parameter, cached by tensor identity (the HIP path packs weights once)."""
    if not rounded and w.dtype == dtype:
        return w
    key = (id(w), w._version, tuple(w.shape), rounded, dtype)
    hit = _WCACHE.get(key)
    if hit is None or hit[0] is not w:
        if len(_WCACHE) > 8192:
            _WCACHE.clear()
        v = w.detach()
        if rounded:
            v = v.to(torch.bfloat16)
        hit = (w, v.to(dtype))
        _WCACHE[key] = hit
    return hit[1]


class _Net:
    """``storage="bf16"``: the ROUNDING-POINT-MATCHED oracle.  fp32 compute everywhere, but every tensor DESIGN.md section 3 says the HIP
    path STORES in bf16 is rounded to bf16 at the same point: frozen weights and the LoRA operands (A and s*B) as MFMA operands, the output
    of every GEMM / conv epilogue (after bias, time-embedding row vector, residual add, activation -- those run on the fp32 accumulator),
    GroupNorm / LayerNorm outputs, the rank-64 projection t = x A^T, q/k/v, the softmax probabilities fed to the PV product, the
    attention output, the GEGLU product.  What remains between this oracle and the HIP path is accumulation order (and the
    probabilities' rounding happening on unnormalised values in the kernel), so the comparison can be asserted at the north-star 1e-3.
    ``storage=None`` is the plain fp32 restatement.
    ``compute``: arithmetic dtype between the rounding points (default fp32).  ``storage="bf16", compute=torch.float64`` is the same
    network with the same rounding points and a different accumulation precision -- its distance from the fp32-compute run measures how
    far two correct evaluations of the SAME bf16-storage network drift apart (the yardstick of tests/test_*_rounding_matched.py)."""

    def __init__(self, cfg, sd, lora, lora_alpha=8.0, storage=None, compute=torch.float32):
        assert storage in (None, "bf16")
        self.cfg, self.sd, self.lora = cfg, sd, lora or {}
        self.alpha = lora_alpha
        self.bf16 = storage == "bf16"
        self.dt = compute

    def q(self, x):
        return _RoundBF16.apply(x) if self.bf16 else x

    def w(self, path):
        return _derived(self.sd[path + ".weight"], self.bf16, self.dt)

    def par(self, key):
        """a parameter the HIP path keeps in fp32 (biases, norm affine, conv_in / conv_out weights)"""
        t = self.sd.get(key)
        return None if t is None else _derived(t, False, self.dt)

    def _lora_ops(self, path, fold=1.0):
        A, B = self.lora[path]
        s = self.alpha / A.shape[0]
        if self.bf16:        # operands the kernels read: bf16(A) and bf16(s * B), both rounded from the fp32 master values
            return self.q(A).to(self.dt), self.q(B * (s * fold)).to(self.dt), 1.0
        return A.to(self.dt), B.to(self.dt), s * fold

    def linear(self, path, x, keep_f32=False, weight_f32=False, fold=1.0):
        """``fold``: a constant folded into the projection's operands (the attention query scale, see attention): in the matched mode the
        bf16 operands are rounded from fold * (fp32 master value), as pcm_amd/model.py packs them"""
        if fold != 1.0:
            w0, b0 = self.sd[path + ".weight"], self.sd.get(path + ".bias")
            w = _RoundBF16.apply(w0 * fold).to(self.dt) if self.bf16 else (w0 * fold).to(self.dt)
            y = F.linear(x, w, None if b0 is None else (b0 * fold).to(self.dt))
        else:
            y = F.linear(x, self.par(path + ".weight") if weight_f32 else self.w(path), self.par(path + ".bias"))
        if path in self.lora:
            A, B, s = self._lora_ops(path, fold)
            y = y + F.linear(self.q(F.linear(x, A)), B) * s
        return y if keep_f32 else self.q(y)

    def conv(self, path, x, stride=1, keep_f32=False, weight_f32=False):
        w = self.par(path + ".weight") if weight_f32 else self.w(path)
        pad = w.shape[-1] // 2
        y = F.conv2d(x, w, self.par(path + ".bias"), stride=stride, padding=pad)
        if path in self.lora:
            A, B, s = self._lora_ops(path)
            y = y + F.conv2d(self.q(F.conv2d(x, A, None, stride=stride, padding=pad)), B) * s
        return y if keep_f32 else self.q(y)

    def gn(self, path, x, eps):
        return F.group_norm(x, self.cfg.norm_num_groups, self.par(path + ".weight"), self.par(path + ".bias"), eps)

    def ln(self, path, x):
        return F.layer_norm(x, (x.shape[-1],), self.par(path + ".weight"), self.par(path + ".bias"), 1e-5)

    def resnet(self, p, x, emb):
        # epilogue order of the HIP path: fp32 accumulator + bias + time-embedding row (itself a bf16-stored projection) -> ONE rounding
        h = self.conv(p + "conv1", self.q(F.silu(self.gn(p + "norm1", x, self.cfg.norm_eps))), keep_f32=True)
        h = self.q(h + self.linear(p + "time_emb_proj", self.q(F.silu(emb)))[:, :, None, None])
        h = self.conv(p + "conv2", self.q(F.silu(self.gn(p + "norm2", h, self.cfg.norm_eps))), keep_f32=True)
        if (p + "conv_shortcut.weight") in self.sd:
            x = self.conv(p + "conv_shortcut", x)
        return self.q(x + h)             # residual added on the fp32 accumulator, then stored

    def attention(self, p, x, ctx, H):
        B, L, C = x.shape
        d = C // H
        if self.bf16:
            # rounding points of the HIP path (round 5): the softmax scale AND the base change live in the query projection -- q is stored once,
            # already in the log2 domain (weights packed from d^-1/2 * log2(e) * W_q, LoRA copy from that times s * B_q); the scores are 2^(q'k)
            q = self.linear(p + "to_q", x, fold=d ** -0.5 * 1.4426950408889634)
            post = 0.6931471805599453
        else:
            q = self.linear(p + "to_q", x)
            post = d ** -0.5
        k = self.linear(p + "to_k", ctx)
        v = self.linear(p + "to_v", ctx)
        q = q.view(B, L, H, d).transpose(1, 2)
        k = k.view(B, -1, H, d).transpose(1, 2)
        v = v.view(B, -1, H, d).transpose(1, 2)
        s = torch.softmax(q @ k.transpose(-1, -2) * post, dim=-1)
        if self.bf16:
            # the kernel feeds bf16 probabilities to the PV MFMA and takes the row sum of the SAME rounded values as the denominator
            s = self.q(s)
            o = self.q((s @ v) / s.sum(-1, keepdim=True)).transpose(1, 2).reshape(B, L, C)
        else:
            o = (s @ v).transpose(1, 2).reshape(B, L, C)
        return self.linear(p + "to_out.0", o, keep_f32=True)      # the caller adds the residual before the store

    def transformer(self, p, x, ctx, depth=1, heads=None):
        heads = heads if heads is not None else self.cfg.heads_at(0)
        B, C, Hh, Ww = x.shape
        r = x
        h = self.q(self.gn(p + "norm", x, 1e-6))
        if self.cfg.use_linear_projection:       # Transformer2DModel: reshape first, then a Linear proj_in
            h = self.linear(p + "proj_in", h.permute(0, 2, 3, 1).reshape(B, Hh * Ww, C))
        else:
            h = self.conv(p + "proj_in", h).permute(0, 2, 3, 1).reshape(B, Hh * Ww, C)
        for k in range(depth):
            b = p + f"transformer_blocks.{k}."
            n = self.q(self.ln(b + "norm1", h))
            h = self.q(h + self.attention(b + "attn1.", n, n, heads))
            h = self.q(h + self.attention(b + "attn2.", self.q(self.ln(b + "norm2", h)), ctx, heads))
            n = self.q(self.ln(b + "norm3", h))
            # GEGLU runs in the projection's epilogue on the fp32 accumulator (fused path); only the product is stored
            a, g = self.linear(b + "ff.net.0.proj", n, keep_f32=self.geglu_fused(B * Hh * Ww)).chunk(2, dim=-1)
            h = self.q(h + self.linear(b + "ff.net.2", self.q(a * F.gelu(g)), keep_f32=True))
        if self.cfg.use_linear_projection:
            h = self.linear(p + "proj_out", h, keep_f32=True).reshape(B, Hh, Ww, C).permute(0, 3, 1, 2)
            return self.q(h + r)
        h = h.reshape(B, Hh, Ww, C).permute(0, 3, 1, 2)
        return self.q(self.conv(p + "proj_out", h, keep_f32=True) + r)

    @staticmethod
    def geglu_fused(M):
        """the HIP schedule fuses GEGLU into the projection from 128 rows up (pcm_amd/model.py transformer_fwd); below that the
        pre-activation makes a bf16 round trip"""
        return M >= 128


def unet_forward(cfg, sd, sample, timesteps, encoder_hidden_states, lora=None, lora_alpha=8.0,
                 return_features=False, added_cond=None, storage=None, compute=torch.float32):
    """UNet2DConditionModel.forward(sample, timestep, encoder_hidden_states).sample in fp32.
    ``return_features`` mimics discriminator_sd15.py modified_forward (features after every down
    block, mid, every up block; no conv_norm_out/conv_out).
    ``storage="bf16"``: the rounding-point-matched variant (see _Net)."""
    net = _Net(cfg, sd, lora, lora_alpha, storage, compute)
    q = net.q
    boc = cfg.block_out_channels
    n = len(boc)
    out_dtype = sample.dtype
    sample = sample.to(compute)
    encoder_hidden_states = q(encoder_hidden_states.to(compute))
    t_emb = q(timestep_embedding(timesteps, boc[0]).to(sample.dtype))
    # SiLU of both embedding projections runs in the GEMM epilogue (fp32) before the store
    emb = net.linear("time_embedding.linear_2", q(F.silu(net.linear("time_embedding.linear_1", t_emb, keep_f32=True))), keep_f32=True)
    if cfg.addition_time_embed_dim:
        # addition_embed_type="text_time" (get_aug_embed): emb += add_embedding(cat(text_embeds, add_time_proj(time_ids.flatten())))
        B = sample.shape[0]
        tid = q(timestep_embedding(added_cond["time_ids"].flatten(), cfg.addition_time_embed_dim).reshape(B, -1).to(sample.dtype))
        add_in = torch.cat([q(added_cond["text_embeds"].to(sample.dtype)), tid], dim=-1)
        a1 = q(F.silu(net.linear("add_embedding.linear_1", add_in, keep_f32=True)))
        emb = q(q(emb) + net.linear("add_embedding.linear_2", a1, keep_f32=True))      # emb_t stored, then the residual add in the epilogue
    h = net.conv("conv_in", sample, weight_f32=True)                                # conv_in / conv_out keep fp32 weights
    skips = [h]
    feats = []
    for i in range(n):
        for j in range(cfg.layers_per_block):
            h = net.resnet(f"down_blocks.{i}.resnets.{j}.", h, emb)
            if cfg.down_attn[i]:
                h = net.transformer(f"down_blocks.{i}.attentions.{j}.", h, encoder_hidden_states, cfg.transformer_depth[i], cfg.heads_at(i))
            skips.append(h)
        if i < n - 1:
            h = net.conv(f"down_blocks.{i}.downsamplers.0.conv", h, stride=2)
            skips.append(h)
        feats.append(h)
    h = net.resnet("mid_block.resnets.0.", h, emb)
    h = net.transformer("mid_block.attentions.0.", h, encoder_hidden_states, cfg.mid_depth, cfg.heads_at(n - 1))
    h = net.resnet("mid_block.resnets.1.", h, emb)
    feats.append(h)
    if return_features == "down_mid":      # discriminator_sdxl.py:311: "do not use up blocks to save memory"
        return feats
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = net.resnet(f"up_blocks.{i}.resnets.{j}.", h, emb)
            if cfg.down_attn[n - 1 - i]:
                h = net.transformer(f"up_blocks.{i}.attentions.{j}.", h, encoder_hidden_states, cfg.transformer_depth[n - 1 - i],
                                    cfg.heads_at(n - 1 - i))
        if i < n - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = net.conv(f"up_blocks.{i}.upsamplers.0.conv", h)
        feats.append(h)
    if return_features:
        return feats
    h = q(F.silu(net.gn("conv_norm_out", h, cfg.norm_eps)))
    return net.conv("conv_out", h, keep_f32=True, weight_f32=True).to(out_dtype)


def peft_state_dict(lora):
    """get_peft_model_state_dict key names (the `.default` segment dropped), as consumed by
    train_pcm_lora_sd15.py:56-58,:921-923."""
    out = OrderedDict()
    for path, (A, B) in lora.items():
        out[f"base_model.model.{path}.lora_A.weight"] = A
        out[f"base_model.model.{path}.lora_B.weight"] = B
    return out
