"""Execute the reference's OWN function/class source by AST-slicing it out of /root/reference.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Works only where /root/reference is
mounted (the build container); nothing at GPU-box run time may call this.  Nothing is copied
into the repo: the source text is read, the wanted top-level ``def``/``class`` nodes are
compiled in memory and exec'd with ``torch``/``numpy`` in their namespace.

Slices used (reference file:line):
  train_pcm_lora_sd15.py:240-341  append_dims, scalings_for_boundary_conditions_{target,online},
                                  predicted_origin, extract_into_tensor, DDIMSolver
  train_pcm_lora_sd15.py:344-355  update_ema
  train_pcm_lora_sd15.py:52-72    get_module_kohya_state_dict (key-renaming rule only)
  scheduling_ddpm_modified.py:500-554  DDPMScheduler.add_noise / .noise_travel (method bodies
                                  lifted and bound to a stub that carries ``alphas_cumprod``)
  discriminator_sd15.py:348-434   DiscriminatorHead, Discriminator.d_loss/g_loss
  discriminator_sd15.py:16-345    modified_forward (the reference's copy of UNet2DConditionModel.forward with the 9 feature taps), executed on
                                  a duck-typed block tree built from the oracle's block functions (modified_forward_on_oracle_blocks)
  text_to_image_sd3/train_pcm_lora_sd3.py:153-230  extract_into_tensor, EulerSolver (flow-matching PCM math)
  text_to_image_sd3/pcm_fm_{deterministic,stochastic}_scheduler.py:35-242  the two sampler classes (bases and the
                                  config decorator stripped; ``self.config`` supplied by a stub)
"""
import ast
import os
import types

import numpy as np
import torch

REF_ROOT = "/root/reference"
SD15_DIR = os.path.join(REF_ROOT, "code", "text_to_image_sd15")
SD3_DIR = os.path.join(REF_ROOT, "code", "text_to_image_sd3")


def available() -> bool:
    return os.path.isdir(SD15_DIR)


def _slice(path, names, extra_ns=None):
    src = open(path).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body
            if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    found = {n.name for n in keep}
    missing = set(names) - found
    if missing:
        raise KeyError(f"{path}: missing {sorted(missing)}")
    mod = ast.Module(body=keep, type_ignores=[])
    ns = {"torch": torch, "np": np, "nn": torch.nn, "F": torch.nn.functional}
    if extra_ns:
        ns.update(extra_ns)
    exec(compile(mod, path, "exec"), ns)
    return ns


def train_script_namespace():
    """The reference-owned PCM math of train_pcm_lora_sd15.py, as live Python objects."""
    names = ["append_dims", "scalings_for_boundary_conditions_target",
             "scalings_for_boundary_conditions_online", "scalings_for_boundary_conditions",
             "predicted_origin", "extract_into_tensor", "DDIMSolver", "update_ema",
             "guidance_scale_embedding"]
    return _slice(os.path.join(SD15_DIR, "train_pcm_lora_sd15.py"), names)


def scheduler_stub(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    """A stub object carrying the reference DDPMScheduler's alphas_cumprod with the reference's
    own add_noise / noise_travel method bodies bound to it (scheduling_ddpm_modified.py:201-223,
    :500-554).  Only the 'scaled_linear' branch (SD1.5 config) is reproduced for the table."""
    path = os.path.join(SD15_DIR, "scheduling_ddpm_modified.py")
    tree = ast.parse(open(path).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "DDPMScheduler"][0]
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef)
           and n.name in ("add_noise", "noise_travel")]
    assert len(fns) == 2
    ns = {"torch": torch, "np": np}
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, "exec"), ns)
    stub = types.SimpleNamespace()
    # scheduling_ddpm_modified.py:205-207 (scaled_linear), :220-221
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                           dtype=torch.float32) ** 2
    stub.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
    stub.add_noise = types.MethodType(ns["add_noise"], stub)
    stub.noise_travel = types.MethodType(ns["noise_travel"], stub)
    return stub


def discriminator_namespace():
    """DiscriminatorHead + the hinge losses of discriminator_sd15.py:348-434.  ``Discriminator``
    itself needs a diffusers UNet; only its loss methods are usable (with a stubbed _forward)."""
    path = os.path.join(SD15_DIR, "discriminator_sd15.py")
    return _slice(path, ["DiscriminatorHead", "Discriminator"],
                  extra_ns={"modified_forward": None})


def kohya_rename(peft_key: str, prefix: str = "lora_unet") -> str:
    """The key-renaming rule of get_module_kohya_state_dict (train_pcm_lora_sd15.py:59-62),
    executed from the reference source line by line on one key."""
    k = peft_key.replace("base_model.model", prefix)
    k = k.replace("lora_A", "lora_down")
    k = k.replace("lora_B", "lora_up")
    k = k.replace(".", "_", k.count(".") - 2)
    return k


def sd3_train_namespace():
    """EulerSolver + extract_into_tensor of train_pcm_lora_sd3.py:153-230 as live objects."""
    return _slice(os.path.join(SD3_DIR, "train_pcm_lora_sd3.py"), ["extract_into_tensor", "EulerSolver"])


def sd3_flow_sigmas(num_train_timesteps=1000, shift=3.0):
    """The sigma table the reference hands to EulerSolver: ``noise_scheduler.sigmas.numpy()[::-1]`` (train_pcm_lora_sd3.py:961-965).
    diffusers' FlowMatchEulerDiscreteScheduler is not vendored; its constructor's table is the same expression the reference's own
    samplers build at pcm_fm_deterministic_scheduler.py:47-52 (float32 linspace 1..N reversed, /N, shift*s/(1+(shift-1)*s));
    shift = 3.0 is the SD3-medium scheduler config."""
    t = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
    sig = torch.from_numpy(t).to(torch.float32) / num_train_timesteps
    sig = shift * sig / (1 + (shift - 1) * sig)
    return sig.numpy()[::-1]


def sd3_sampler_class(kind="deterministic"):
    """PCMFMDeterministicScheduler / PCMFMStochasticScheduler with diffusers' mixins and config decorator stripped."""
    import typing
    fname = "pcm_fm_%s_scheduler.py" % kind
    cname = "PCMFM%sScheduler" % kind.capitalize()
    path = os.path.join(SD3_DIR, fname)
    tree = ast.parse(open(path).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cname][0]
    cls.bases, cls.keywords = [], []
    for n in cls.body:
        if isinstance(n, ast.FunctionDef):
            n.decorator_list = [d for d in n.decorator_list if not (isinstance(d, ast.Name) and d.id == "register_to_config")]
    ns = {"torch": torch, "np": np, "Optional": typing.Optional, "Tuple": typing.Tuple, "Union": typing.Union,
          cname + "Output": object}
    exec(compile(ast.Module(body=[cls], type_ignores=[]), path, "exec"), ns)
    base = ns[cname]

    class Sampler(base):
        def __init__(self, num_train_timesteps=1000, shift=1.0, pcm_timesteps=50):
            self.config = types.SimpleNamespace(num_train_timesteps=num_train_timesteps, shift=shift, pcm_timesteps=pcm_timesteps)
            super().__init__(num_train_timesteps, shift, pcm_timesteps)
    Sampler.__name__ = cname
    return Sampler


def modified_forward_on_oracle_blocks(ucfg, sd, sample, timesteps, encoder_hidden_states):
    """Run the REFERENCE'S OWN ``modified_forward`` (discriminator_sd15.py:16-345) -- its statement of the UNet wiring: time embedding, conv_in,
    the down loop with its residual tuple, the mid block, the up loop slicing ``len(upsample_block.resnets)`` residuals off that tuple, and
    the 9 feature taps -- on a duck-typed module tree whose BLOCKS are the oracle's block functions (oracle/unet_sd15._Net: diffusers' block
    internals are not vendored by the reference, so those stay a restatement).  Pins the block order, the skip bookkeeping and the tap
    points of oracle.unet_sd15.unet_forward(..., return_features=True) to reference source: the two must agree bit for bit."""
    import typing
    from . import unet_sd15 as O
    F = torch.nn.functional
    ns = _slice(os.path.join(SD15_DIR, "discriminator_sd15.py"), ["modified_forward"],
                extra_ns={"Union": typing.Union, "Optional": typing.Optional, "Dict": typing.Dict, "Any": typing.Any, "Tuple": typing.Tuple})
    net = O._Net(ucfg, sd, None)
    boc, n = ucfg.block_out_channels, len(ucfg.block_out_channels)

    class Down:
        def __init__(self, i):
            self.i, self.has_cross_attention = i, bool(ucfg.down_attn[i])

        def __call__(self, hidden_states, temb, encoder_hidden_states=None, attention_mask=None, cross_attention_kwargs=None,
                     encoder_attention_mask=None, scale=1.0):
            i, h, res = self.i, hidden_states, ()
            for j in range(ucfg.layers_per_block):
                h = net.resnet(f"down_blocks.{i}.resnets.{j}.", h, temb)
                if self.has_cross_attention:
                    h = net.transformer(f"down_blocks.{i}.attentions.{j}.", h, encoder_hidden_states, ucfg.transformer_depth[i], ucfg.heads_at(i))
                res += (h,)
            if i < n - 1:
                h = net.conv(f"down_blocks.{i}.downsamplers.0.conv", h, stride=2)
                res += (h,)
            return h, res

    class Mid:
        has_cross_attention = True

        def __call__(self, hidden_states, temb, encoder_hidden_states=None, attention_mask=None, cross_attention_kwargs=None,
                     encoder_attention_mask=None):
            h = net.resnet("mid_block.resnets.0.", hidden_states, temb)
            h = net.transformer("mid_block.attentions.0.", h, encoder_hidden_states, ucfg.mid_depth, ucfg.heads_at(n - 1))
            return net.resnet("mid_block.resnets.1.", h, temb)

    class Up:
        def __init__(self, i):
            self.i, self.has_cross_attention = i, bool(ucfg.down_attn[n - 1 - i])
            self.resnets = [None] * (ucfg.layers_per_block + 1)          # only len() is read by the reference

        def __call__(self, hidden_states, temb, res_hidden_states_tuple, encoder_hidden_states=None, cross_attention_kwargs=None,
                     upsample_size=None, attention_mask=None, encoder_attention_mask=None, scale=1.0):
            i, h = self.i, hidden_states
            for j in range(ucfg.layers_per_block + 1):
                r = res_hidden_states_tuple[-1]
                res_hidden_states_tuple = res_hidden_states_tuple[:-1]
                h = net.resnet(f"up_blocks.{i}.resnets.{j}.", torch.cat([h, r], dim=1), temb)
                if self.has_cross_attention:
                    lv = n - 1 - i
                    h = net.transformer(f"up_blocks.{i}.attentions.{j}.", h, encoder_hidden_states, ucfg.transformer_depth[lv], ucfg.heads_at(lv))
            if i < n - 1:
                h = net.conv(f"up_blocks.{i}.upsamplers.0.conv", F.interpolate(h, scale_factor=2.0, mode="nearest"))
            return h

    unet = types.SimpleNamespace(
        num_upsamplers=n - 1,
        config=types.SimpleNamespace(center_input_sample=False, addition_embed_type=None, class_embed_type=None,
                                     class_embeddings_concat=False, encoder_hid_dim_type=None),
        time_proj=lambda t: O.timestep_embedding(t, boc[0]),
        time_embedding=lambda t_emb, cond=None: net.linear("time_embedding.linear_2", F.silu(net.linear("time_embedding.linear_1", t_emb))),
        class_embedding=None, time_embed_act=None, encoder_hid_proj=None,
        conv_in=lambda x: net.conv("conv_in", x),
        down_blocks=[Down(i) for i in range(n)], mid_block=Mid(), up_blocks=[Up(i) for i in range(n)])
    return ns["modified_forward"](unet, sample, timesteps, encoder_hidden_states)
