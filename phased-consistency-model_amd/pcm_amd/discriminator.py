"""Latent adversarial discriminator of PCM (reference: code/text_to_image_sd15/discriminator_sd15.py).

``Discriminator(unet)`` there = the frozen teacher UNet used as a feature extractor
(``modified_forward``, :16-345 -> pcm_amd.model.UNet.forward(features=True)) + 9 x 4 trainable
``DiscriminatorHead``s (:348-393): conv3x3 -> GroupNorm(32) -> LeakyReLU, conv3x3 -> GN -> LeakyReLU (+ skip),
conv1x1 -> 1 logit map; hinge losses :412-434.

All head parameters live in ONE flat fp32 buffer (+ grad + Adam moments).  conv3x3 weights are stored
[co][kh][kw][ci] internally (contiguous weight-gradient atomics, direct MFMA operand packing); the
reference / torch layout [co][ci][kh][kw] is produced on export.
"""
import math
from collections import OrderedDict

import torch

from . import capi, ops
from .ops import Seg

from .precision import act_dtype as _act_dtype  # noqa: E402

ADAPTER_DIMS = (320, 640, 1280, 1280, 1280, 1280, 1280, 640, 320)   # discriminator_sd15.py:377
ADAPTER_DIMS_SDXL = (320, 640, 1280, 1280)                            # discriminator_sdxl.py:377-386 (down blocks + mid only)


class Head:
    __slots__ = ("C", "p", "g", "wf", "wb", "off0", "off1")


class Discriminator:
    def __init__(self, adapter_channel_dims=ADAPTER_DIMS, num_h_per_head=4, device="cuda", seed=2, groups=32, ksize=3, taps=True):
        """``ksize`` 3: the SD1.5 heads (conv3x3); 1: the SDXL heads (discriminator_sdxl.py:348-369, "1x1 to save memory").
        ``taps``: feature mode handed to ``UNet.forward(features=...)`` (True: 9 taps; "down_mid": the 4 SDXL taps)."""
        assert ksize in (1, 3)
        self.dims, self.nh, self.device, self.G = tuple(adapter_channel_dims), num_h_per_head, torch.device(device), groups
        self.ksize, self.taps = ksize, taps
        self.head_num = len(self.dims)
        layout, total = [], 0
        for k, C in enumerate(self.dims):
            for h in range(num_h_per_head):
                wshape = (C, 3, 3, C) if ksize == 3 else (C, C)
                names = OrderedDict([("conv1.0.weight", wshape), ("conv1.0.bias", (C,)), ("conv1.1.weight", (C,)),
                                     ("conv1.1.bias", (C,)), ("conv2.0.weight", wshape), ("conv2.0.bias", (C,)),
                                     ("conv2.1.weight", (C,)), ("conv2.1.bias", (C,)), ("conv_out.weight", (C,)),
                                     ("conv_out.bias", (1,))])
                offs = {}
                for n, shp in names.items():
                    offs[n] = (total, shp)
                    total += (math.prod(shp) + 3) // 4 * 4          # keep every tensor 16-byte aligned
                layout.append((k, h, C, offs))
        self.numel = total
        f32 = dict(dtype=torch.float32, device=self.device)
        self.params = torch.zeros(total, **f32)
        self.grads = torch.zeros(total, **f32)
        self.exp_avg = torch.zeros(total, **f32)
        self.exp_avg_sq = torch.zeros(total, **f32)
        self.gradsq = torch.zeros(1, dtype=torch.float64, device=self.device)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.heads = []
        g = torch.Generator().manual_seed(seed)
        for k, h, C, offs in layout:
            hd = Head()
            hd.C = C
            hd.off0 = min(o for o, _ in offs.values())                    # this head's range of the flat buffers
            hd.off1 = max(o + (math.prod(shp) + 3) // 4 * 4 for o, shp in offs.values())
            hd.p = {n: self.params[o:o + math.prod(shp)].view(shp) for n, (o, shp) in offs.items()}
            hd.g = {n: self.grads[o:o + math.prod(shp)].view(shp) for n, (o, shp) in offs.items()}
            # torch default init of nn.Conv2d / nn.GroupNorm (DiscriminatorHead.__init__, :349-362)
            for cv in ("conv1.0", "conv2.0"):
                bound = 1.0 / math.sqrt(ksize * ksize * C)
                w = (torch.rand(C, C, ksize, ksize, generator=g) * 2 - 1) * bound
                hd.p[cv + ".weight"].copy_((w.permute(0, 2, 3, 1) if ksize == 3 else w.view(C, C)).to(self.device))
                hd.p[cv + ".bias"].copy_(((torch.rand(C, generator=g) * 2 - 1) * bound).to(self.device))
            for gn in ("conv1.1", "conv2.1"):
                hd.p[gn + ".weight"].fill_(1.0)
            bound = 1.0 / math.sqrt(C)
            hd.p["conv_out.weight"].copy_(((torch.rand(C, generator=g) * 2 - 1) * bound).to(self.device))
            hd.p["conv_out.bias"].copy_(((torch.rand(1, generator=g) * 2 - 1) * bound).to(self.device))
            hd.wf = {cv: torch.empty(C, ksize * ksize * C, dtype=_act_dtype(), device=self.device) for cv in ("conv1.0", "conv2.0")}
            hd.wb = {cv: torch.empty(C, ksize * ksize * C, dtype=_act_dtype(), device=self.device) for cv in ("conv1.0", "conv2.0")}
            self.heads.append((k, hd))
        self.repack()

    def repack(self):
        for _, hd in self.heads:
            for cv in ("conv1.0", "conv2.0"):
                if self.ksize == 3:
                    ops.pack_conv3x3(hd.p[cv + ".weight"], True, True, 1.0, hd.wf[cv], hd.wb[cv], khwc=True)
                else:
                    ops.pack_linear(hd.p[cv + ".weight"], True, True, 1.0, hd.wf[cv], hd.wb[cv])

    def state_dict(self):
        """reference names / torch layouts: heads.{k}.{h}.conv1.0.weight [C,C,3,3], ... conv_out.weight [1,C,1,1]"""
        out, cnt = OrderedDict(), {}
        for k, hd in self.heads:
            h = cnt.get(k, 0)
            cnt[k] = h + 1
            for n, t in hd.p.items():
                v = t.detach().clone()
                if n in ("conv1.0.weight", "conv2.0.weight"):
                    v = v.permute(0, 3, 1, 2).contiguous() if self.ksize == 3 else v.view(hd.C, hd.C, 1, 1)
                elif n == "conv_out.weight":
                    v = v.view(1, hd.C, 1, 1)
                out[f"heads.{k}.{h}.{n}"] = v
        return out

    def load_state_dict(self, sd):
        cnt = {}
        for k, hd in self.heads:
            h = cnt.get(k, 0)
            cnt[k] = h + 1
            for n, t in hd.p.items():
                v = sd[f"heads.{k}.{h}.{n}"].to(self.device)
                if n in ("conv1.0.weight", "conv2.0.weight") and self.ksize == 3:
                    v = v.permute(0, 2, 3, 1)
                t.copy_(v.reshape(t.shape))
        self.repack()

    # ------------------------------------------------------------------ forward / backward of the heads
    def forward(self, feats, save=False):
        """feats: list of 9 (tensor [B, HW, C] bf16, H, W).  Returns list of 36 logit maps fp32 [B*HW] (+ tape)."""
        logits, tape = [], ([] if save else None)
        for k, hd in self.heads:
            f, H, W = feats[k]
            B, C = f.shape[0], hd.C
            M = B * H * W
            geo = dict(Hs=H, Ws=W) if self.ksize == 3 else None
            a1 = torch.empty(M, C, dtype=_act_dtype(), device=f.device)
            ops.gemm([Seg(f if geo else f.reshape(M, C), hd.wf["conv1.0"], conv=geo)], M, C, a1, bias=hd.p["conv1.0.bias"], Ho=H, Wo=W)
            n1, st1 = ops.groupnorm_fwd(a1.view(B, H * W, C), hd.p["conv1.1.weight"], hd.p["conv1.1.bias"], self.G, 1e-5, capi.ACT_LEAKY)
            a2 = torch.empty(M, C, dtype=_act_dtype(), device=f.device)
            ops.gemm([Seg(n1 if geo else n1.reshape(M, C), hd.wf["conv2.0"], conv=geo)], M, C, a2, bias=hd.p["conv2.0.bias"], Ho=H, Wo=W)
            n2, st2 = ops.groupnorm_fwd(a2.view(B, H * W, C), hd.p["conv2.1.weight"], hd.p["conv2.1.bias"], self.G, 1e-5, capi.ACT_LEAKY)
            h2 = ops.add(n2, n1)                                                         # x = conv2(x) + x   (:366)
            logits.append(ops.rowdot_fwd(h2, hd.p["conv_out.weight"], hd.p["conv_out.bias"]))
            if save:
                tape.append(dict(f=f, H=H, W=W, B=B, a1=a1, st1=st1, n1=n1, a2=a2, st2=st2, h2=h2))
        return (logits, tape) if save else logits

    def backward(self, d_logits, tape, param_grads=True, feature_grads=False, on_bucket=None):
        """d_logits: 36 fp32 [M] gradients.  Accumulates parameter gradients (discriminator step) and / or returns
        the 9 feature gradients (generator step).  ``on_bucket(off0, off1)``: called when the parameter gradients of ALL heads of one
        feature are final (flat range of self.grads) -- the data-parallel trainer starts that bucket's all-reduce behind the rest."""
        d_feats = [None] * self.head_num
        first_of = {}
        for i, (k, hd) in enumerate(self.heads):
            first_of.setdefault(k, hd.off0)
        for i, ((k, hd), dl, sv) in enumerate(zip(self.heads, d_logits, tape)):
            B, H, W, C = sv["B"], sv["H"], sv["W"], hd.C
            M = B * H * W
            geo = dict(Hs=H, Ws=W) if self.ksize == 3 else None
            wg = dict(Hs=H, Ws=W, Ho=H, Wo=W)
            if param_grads:
                d_h2 = ops.rowdot_bwd(sv["h2"], hd.p["conv_out.weight"], dl, hd.g["conv_out.weight"], hd.g["conv_out.bias"])
            else:
                scratch = torch.zeros(C + 4, dtype=torch.float32, device=dl.device)
                d_h2 = ops.rowdot_bwd(sv["h2"], hd.p["conv_out.weight"], dl, scratch, scratch[C:])
            # GN2 + LeakyReLU
            gam2, bet2 = hd.p["conv2.1.weight"], hd.p["conv2.1.bias"]
            a2 = sv["a2"].view(B, H * W, C)
            if param_grads:
                ops.groupnorm_param_grad(a2, d_h2, sv["st2"], gam2, bet2, hd.g["conv2.1.weight"], hd.g["conv2.1.bias"], self.G, 1e-5, capi.ACT_LEAKY)
            d_a2 = ops.groupnorm_bwd(a2, d_h2, sv["st2"], gam2, bet2, self.G, 1e-5, capi.ACT_LEAKY)
            if param_grads:
                self._conv_param_grads(hd, "conv2.0", sv["n1"], d_a2.view(M, C), M, wg)
            d_n1 = torch.empty(M, C, dtype=_act_dtype(), device=dl.device)                     # dgrad(conv2) + skip branch
            ops.gemm([Seg(d_a2 if geo else d_a2.reshape(M, C), hd.wb["conv2.0"], conv=geo)], M, C, d_n1, residual=d_h2.view(M, C), Ho=H, Wo=W)
            gam1, bet1 = hd.p["conv1.1.weight"], hd.p["conv1.1.bias"]
            a1 = sv["a1"].view(B, H * W, C)
            if param_grads:
                ops.groupnorm_param_grad(a1, d_n1.view(B, H * W, C), sv["st1"], gam1, bet1, hd.g["conv1.1.weight"], hd.g["conv1.1.bias"], self.G, 1e-5, capi.ACT_LEAKY)
            d_a1 = ops.groupnorm_bwd(a1, d_n1.view(B, H * W, C), sv["st1"], gam1, bet1, self.G, 1e-5, capi.ACT_LEAKY)
            if param_grads:
                self._conv_param_grads(hd, "conv1.0", sv["f"], d_a1.view(M, C), M, wg)
            if feature_grads:
                d_f = torch.empty(M, C, dtype=_act_dtype(), device=dl.device)
                ops.gemm([Seg(d_a1 if geo else d_a1.reshape(M, C), hd.wb["conv1.0"], conv=geo)], M, C, d_f, residual=None if d_feats[k] is None else d_feats[k].view(M, C),
                         Ho=H, Wo=W)                                                   # 4 heads share one feature: sum in the epilogue
                d_feats[k] = d_f.view(B, H * W, C)
            if on_bucket is not None and param_grads and (i + 1 == len(self.heads) or self.heads[i + 1][0] != k):
                on_bucket(first_of[k], hd.off1)
        return d_feats

    def _conv_param_grads(self, hd, name, x, dy, M, wg):
        """full conv3x3 weight gradient dW[co][tap][ci] = sum_m dy[m][co] * im2col(x)[m][(tap,ci)]: the dense kernel
        (csrc/wgrad_dense.hip) where the geometry allows, else Cout/64 launches of the rank-64 wgrad kernel; bias gradient = pixel sum of dy."""
        C = hd.C
        if self.ksize == 1:      # dW[co][ci] = sum_m dy[m][co] * x[m][ci]: plain rank-64 wgrad per 64 output channels
            gW = hd.g[name + ".weight"].view(C, C)
            with ops.wgrad_batch():       # the Cout/64 chunks share launches
                for co in range(0, C, 64):
                    ops.lora_wgrad(x.reshape(M, C), dy[:, co:co + 64], gW[co:co + 64], 1.0, M, G=C, g_stride=1, r_stride=C, lds=C)
            ops.colsum_into(dy, hd.g[name + ".bias"], M, C)
            return
        gW = hd.g[name + ".weight"].view(C, 9 * C)
        if ops.conv3x3_wgrad_ok(wg["Hs"], wg["Ws"], C, C):          # dense kernel: one launch, x staged once per 128 input channels
            ops.conv3x3_wgrad(x.reshape(-1, wg["Hs"], wg["Ws"], C), dy, gW, M // (wg["Hs"] * wg["Ws"]), wg["Hs"], wg["Ws"])
        else:
            with ops.wgrad_batch():       # (3x3 view: the jobs the multi-launch kernel does not take run one by one inside the call)
                for co in range(0, C, 64):
                    ops.lora_wgrad(x, dy[:, co:co + 64], gW[co:co + 64], 1.0, M, conv=wg, g_stride=1, r_stride=9 * C, lds=C)
        ops.colsum_into(dy, hd.g[name + ".bias"], M, C)

    # ------------------------------------------------------------------ losses (discriminator_sd15.py:412-434)
    def d_loss_backward(self, logits_fake_real, tape, B_half, weight=1.0, on_bucket=None, loss_scale_dev=None):
        """logits of the batched [fake; real] pass.  Returns loss (fp64 [1]); accumulates head gradients.
        ``loss_scale_dev`` (half build): the fp32 logit gradients are multiplied by the device-side loss scale before the 16-bit backward."""
        loss = torch.zeros(1, dtype=torch.float64, device=self.device)
        n_heads = self.head_num * self.nh
        d_logits = []
        for lg in logits_fake_real:
            half = lg.numel() // 2
            df, dr = ops.hinge_loss(lg[:half], lg[half:], 0, weight / n_heads, loss)
            d_logits.append(torch.cat([df, dr]))
            if loss_scale_dev is not None:
                ops.scale_by_dev(d_logits[-1], loss_scale_dev)
        self.backward(d_logits, tape, param_grads=True, feature_grads=False, on_bucket=on_bucket)
        return loss

    def g_loss_backward(self, logits_fake, tape, weight=1.0, grad_scale=1.0, loss_scale_dev=None):
        """Returns (loss fp64 [1], d_feats): gradient of grad_scale * g_loss (times the device-side loss scale, half build) wrt the 9 teacher features."""
        loss = torch.zeros(1, dtype=torch.float64, device=self.device)
        n_heads = self.head_num * self.nh
        d_logits = []
        for lg in logits_fake:
            df, _ = ops.hinge_loss(lg, None, 1, weight / n_heads, loss, grad_scale=grad_scale)
            if loss_scale_dev is not None:
                ops.scale_by_dev(df, loss_scale_dev)
            d_logits.append(df)
        d_feats = self.backward(d_logits, tape, param_grads=False, feature_grads=True)
        return loss, d_feats


def __getattr__(name):
    # ``<module>.BF16`` = "the library's 16-bit dtype" for external readers (tests, tools): a call-time lookup, never a captured constant
    if name == "BF16":
        return _act_dtype()
    raise AttributeError(name)
