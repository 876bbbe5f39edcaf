"""Checkpoint formats of the reference trainer, written without diffusers/peft installed:

* peft adapter dir   — ``adapter_model.safetensors`` + ``adapter_config.json``
                        (``unet.save_pretrained(output_dir)``, train_pcm_lora_sd15.py:928,:1378)
* diffusers LoRA     — ``unet_lora/pytorch_lora_weights.safetensors`` with the ``unet.`` prefix
                        (``StableDiffusionPipeline.save_lora_weights``, :924-926,:1380-1382)
* kohya-ss dict      — ``get_module_kohya_state_dict`` (:52-72): what ``log_validation`` / the demo's
                        ``pipe.load_lora_weights`` consume (``pcm_sd15_*_converted.safetensors``)
* trainer state      — optimizer moments + step (the reference leaves this to accelerate.save_state)
"""
import json
import os
import shutil

import torch
from safetensors.torch import load_file, save_file

from .unet_spec import LORA_TARGETS


def peft_state_dict(lora):
    return {k: v.detach().cpu().contiguous() for k, v in lora.peft_state_dict().items()}


def kohya_state_dict(peft_sd, lora_alpha, prefix="lora_unet", dtype=torch.float16):
    """train_pcm_lora_sd15.py:52-72."""
    out = {}
    for peft_key, weight in peft_sd.items():
        k = peft_key.replace("base_model.model", prefix)
        k = k.replace("lora_A", "lora_down").replace("lora_B", "lora_up")
        k = k.replace(".", "_", k.count(".") - 2)
        out[k] = weight.to(dtype)
        if "lora_down" in k:
            out[f'{k.split(".")[0]}.alpha'] = torch.tensor(lora_alpha).to(dtype)
    return out


def adapter_config(lora):
    """The fields peft 0.9 writes for LoraConfig(r, target_modules) (:866-884)."""
    return {"peft_type": "LORA", "task_type": None, "base_model_name_or_path": None, "r": lora.real_rank,
            "lora_alpha": lora.alpha, "lora_dropout": 0.0, "bias": "none", "fan_in_fan_out": False,
            "init_lora_weights": True, "inference_mode": True, "target_modules": list(LORA_TARGETS),
            "modules_to_save": None, "rank_pattern": {}, "alpha_pattern": {}, "use_rslora": False, "revision": None,
            "layers_pattern": None, "layers_to_transform": None, "megatron_config": None, "megatron_core": "megatron.core",
            "loftq_config": {}, "use_dora": False}


def save_lora(lora, output_dir, kohya=True):
    os.makedirs(os.path.join(output_dir, "unet_lora"), exist_ok=True)
    sd = peft_state_dict(lora)
    save_file(sd, os.path.join(output_dir, "adapter_model.safetensors"))
    with open(os.path.join(output_dir, "adapter_config.json"), "w") as f:
        json.dump(adapter_config(lora), f, indent=2)
    save_file({"unet." + k: v for k, v in sd.items()}, os.path.join(output_dir, "unet_lora", "pytorch_lora_weights.safetensors"))
    if kohya:
        save_file(kohya_state_dict(sd, lora.alpha), os.path.join(output_dir, "pcm_lora_kohya_converted.safetensors"))


def save_lora_sd3(lora, output_dir):
    """``StableDiffusion3Pipeline.save_lora_weights(output_dir, transformer_lora_layers=get_peft_model_state_dict(transformer))``
    (train_pcm_lora_sd3.py:1010-1012, :1495-1500): ``pytorch_lora_weights.safetensors`` with ``transformer.<module>.lora_A|B.weight``
    keys, at the real LoRA rank.  The peft adapter dir is written beside it for resuming."""
    os.makedirs(output_dir, exist_ok=True)
    sd = peft_state_dict(lora)
    save_file(sd, os.path.join(output_dir, "adapter_model.safetensors"))
    save_file({"transformer." + k[len("base_model.model."):]: v for k, v in sd.items()}, os.path.join(output_dir, "pytorch_lora_weights.safetensors"))


def load_transformer_state_dict(pretrained_dir):
    """diffusers SD3 layout: <dir>/transformer/diffusion_pytorch_model*.safetensors (possibly sharded)."""
    import glob
    files = sorted(glob.glob(os.path.join(pretrained_dir, "transformer", "diffusion_pytorch_model*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no transformer/diffusion_pytorch_model*.safetensors under {pretrained_dir}")
    sd = {}
    for f in files:
        sd.update(load_file(f))
    return sd


def unet_lora_from_file(cfg, path, device, lora_alpha=8.0, scale=1.0):
    """LoRA state for the UNet inferred FROM a LoRA file in any of the three formats the reference handles: kohya-ss
    (``lora_unet_<module_with_underscores>.lora_down|lora_up.weight`` + ``.alpha``: what get_module_kohya_state_dict writes,
    train_pcm_lora_sd15.py:52-72, and what demo/app.py loads), peft (``base_model.model.<module>.lora_A|B.weight``) or diffusers
    (``unet.<...>``).  Rank comes from the file, ``lora_alpha`` from the kohya ``.alpha`` entries when present."""
    from .model import LoraState
    from .unet_spec import lora_target_modules
    raw = load_file(path)
    all_targets = lora_target_modules(cfg)
    norm, alpha = {}, None
    if any(k.startswith("lora_unet_") for k in raw):
        inv = {"lora_unet_" + p_.replace(".", "_"): p_ for p_, _ in all_targets}
        for k, v in raw.items():
            mod, _, leaf = k.partition(".")
            if leaf == "alpha":
                alpha = float(v)
                continue
            if mod not in inv:
                raise KeyError(f"{path}: {mod} is not a LoRA-targeted module of this UNet")
            norm[inv[mod] + (".lora_A.weight" if leaf.startswith("lora_down") else ".lora_B.weight")] = v
    else:
        for k, v in raw.items():
            for pre in ("unet.", "base_model.model."):
                if k.startswith(pre):
                    k = k[len(pre):]
            norm[k] = v
    have = {k[:-len(".lora_A.weight")] for k in norm if k.endswith(".lora_A.weight")}
    targets = [(p_, shp) for p_, shp in all_targets if p_ in have]
    if not targets or have - {p_ for p_, _ in targets}:
        raise KeyError(f"{path}: no (or unknown) LoRA modules for this UNet: {sorted(have - {p_ for p_, _ in targets})[:3]}")
    rank = int(norm[targets[0][0] + ".lora_A.weight"].shape[0])
    lora = LoraState(cfg, rank, alpha if alpha is not None else lora_alpha, device, targets=targets)
    f = float(scale) ** 0.5
    lora.load_peft_state_dict({f"base_model.model.{p_}.lora_{ab}.weight": norm[f"{p_}.lora_{ab}.weight"].float() * f for p_, _ in targets for ab in "AB"})
    return lora


def sd3_lora_from_file(cfg, path, device, lora_alpha=8.0, scale=1.0):
    """LoRA state for the SD3 transformer inferred FROM a LoRA file: ``pytorch_lora_weights.safetensors`` (``transformer.<module>.lora_A|B
    .weight``, what the trainers and StableDiffusion3Pipeline.save_lora_weights write and ``pipe.load_lora_weights`` of
    code/text_to_image_sd3/sd3_test.py:13-20 reads) or a peft ``adapter_model.safetensors``.  Module set and rank come from the file, so
    adapters of the base trainer (8 suffixes) and of the adversarial trainers (22-entry list) both load.  ``scale``: sd3_test.py's ``alpha``
    (every tensor multiplied by sqrt(alpha))."""
    from .mmdit import sd3_lora_state  # noqa: F401  (documented entry point; the state is built directly below)
    from .mmdit_spec import param_spec
    from .model import LoraState
    raw = load_file(path)
    norm = {}
    for k, v in raw.items():
        for pre in ("transformer.", "base_model.model."):
            if k.startswith(pre):
                k = k[len(pre):]
        norm[k] = v
    have = {k[:-len(".lora_A.weight")] for k in norm if k.endswith(".lora_A.weight")}
    shapes = [(k[:-len(".weight")], shp) for k, shp in param_spec(cfg) if k.endswith(".weight")]
    targets = [(p_, shp) for p_, shp in shapes if p_ in have]
    unknown = have - {p_ for p_, _ in targets}
    if unknown or not targets:
        raise KeyError(f"{path}: LoRA modules not in the SD3 transformer: {sorted(unknown)[:3]}" if unknown else f"{path}: no lora_A tensors")
    rank = int(norm[targets[0][0] + ".lora_A.weight"].shape[0])
    lora = LoraState(cfg, rank, lora_alpha, device, targets=targets, init="gaussian")
    f = float(scale) ** 0.5
    lora.load_peft_state_dict({f"base_model.model.{p_}.lora_{ab}.weight": norm[f"{p_}.lora_{ab}.weight"].float() * f for p_, _ in targets for ab in "AB"})
    return lora


def load_lora(lora, input_dir):
    lora.load_peft_state_dict(load_file(os.path.join(input_dir, "adapter_model.safetensors")))


def save_state(distiller, output_dir, global_step):
    """checkpoint-N directory: adapter + diffusers LoRA (the reference's save hook) + optimizer state."""
    if type(getattr(getattr(distiller, "W", None), "cfg", None)).__name__ == "MMDiTConfig":
        save_lora_sd3(distiller.lora, output_dir)          # SD3: the transformer save hook's layout (train_pcm_lora_sd3.py:997-1012)
    else:
        save_lora(distiller.lora, output_dir, kohya=False)
    lo = distiller.lora
    save_file({"exp_avg": lo.exp_avg.detach().cpu(), "exp_avg_sq": lo.exp_avg_sq.detach().cpu()},
              os.path.join(output_dir, "optimizer.safetensors"))
    state = {"global_step": global_step, "optimizer_step": distiller.step_count}
    if getattr(distiller, "loss_scale_dev", None) is not None:      # fp16 build: the GradScaler state (accelerate saves scaler.pt, :1338)
        state["optimizer_step"] = int(distiller.step_dev.item())    # updates skipped on a non-finite norm are not optimizer steps
        state["loss_scale"] = float(distiller.loss_scale_dev.item())
        state["loss_scale_good_steps"] = int(distiller.loss_good_dev.item())
    with open(os.path.join(output_dir, "trainer_state.json"), "w") as f:
        json.dump(state, f)


def load_state(distiller, input_dir):
    load_lora(distiller.lora, input_dir)
    p = os.path.join(input_dir, "optimizer.safetensors")
    if os.path.exists(p):
        st = load_file(p)
        distiller.lora.exp_avg.copy_(st["exp_avg"].to(distiller.lora.device))
        distiller.lora.exp_avg_sq.copy_(st["exp_avg_sq"].to(distiller.lora.device))
    with open(os.path.join(input_dir, "trainer_state.json")) as f:
        st = json.load(f)
    distiller.step_count = st["optimizer_step"]
    distiller.step_dev.fill_(st["optimizer_step"])      # the AdamW kernel reads its bias-correction step from device memory
    if getattr(distiller, "loss_scale_dev", None) is not None and "loss_scale" in st:
        distiller.loss_scale_dev.fill_(st["loss_scale"])
        distiller.loss_good_dev.fill_(st["loss_scale_good_steps"])
    return st["global_step"]


def rotate_checkpoints(output_dir, total_limit):
    """train_pcm_lora_sd15.py:1311-1337: keep at most total_limit-1 before saving a new one."""
    if total_limit is None:
        return
    cks = sorted([d for d in os.listdir(output_dir) if d.startswith("checkpoint")], key=lambda x: int(x.split("-")[1]))
    if len(cks) >= total_limit:
        for d in cks[0:len(cks) - total_limit + 1]:
            shutil.rmtree(os.path.join(output_dir, d))


def latest_checkpoint(output_dir):
    """:1086-1091"""
    if not os.path.isdir(output_dir):
        return None
    dirs = sorted([d for d in os.listdir(output_dir) if d.startswith("checkpoint")], key=lambda x: int(x.split("-")[1]))
    return dirs[-1] if dirs else None


def load_unet_state_dict(pretrained_dir):
    """diffusers layout: <dir>/unet/diffusion_pytorch_model.safetensors (train_pcm_lora_sd15.py:840-851)."""
    for name in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors"):
        p = os.path.join(pretrained_dir, "unet", name)
        if os.path.exists(p):
            return load_file(p)
    raise FileNotFoundError(f"no unet/diffusion_pytorch_model*.safetensors under {pretrained_dir}")
