"""The 16-bit activation / weight format of this process: "bf16" (default; what BASELINE.json's configs and bench.py use) or "fp16".

The reference selects it with ``--mixed_precision`` (train_pcm_lora_sd15.py:1034 hands it to accelerate; every recipe in
train_pcm_lora_sd15.sh passes fp16, BASELINE.json's configs bf16).  Here the format is a BUILD variant of the kernel library
(csrc/pcm_common.h: one block of conversions + the two MFMA builtins), so ``set_precision("fp16")`` (1) loads lib/libpcm_hip_f16.so
instead of lib/libpcm_hip.so and (2) rebinds the ``BF16`` dtype constant of the host modules (it means "the library's 16-bit dtype")
to torch.float16.  Call it BEFORE building UNetWeights / LoraState / Distiller objects: packed operand buffers are allocated in the
current format.  fp16 needs loss scaling in the backward; trainer.Distiller does that with device-side GradScaler state.
"""
import sys

import torch

from . import capi

_NAME = "bf16"
_DTYPES = {"bf16": torch.bfloat16, "fp16": torch.float16}


def precision():
    return _NAME


def act_dtype():
    return _DTYPES[_NAME]


def set_precision(name, lib=None):
    """name: "bf16" | "fp16" (there is no fp32-storage build).  ``lib``: tests pass an emulator build of the matching variant."""
    global _NAME
    if name not in _DTYPES:
        raise ValueError("precision must be 'bf16' or 'fp16', got %r" % (name,))
    want = 1 if name == "fp16" else 0
    if lib is None:
        capi._LIB_PATH = capi.F16_LIB if name == "fp16" else capi.DEFAULT_LIB      # capi.set_lib(None) / lib() keep loading this variant
        lib = capi.Lib(capi._LIB_PATH)
    if lib.act_dtype != want:
        raise RuntimeError("%s was built for %s, not %s" % (lib.path, "fp16" if lib.act_dtype else "bf16", name))
    capi.set_lib(lib)
    _NAME = name
    for mod in ("pcm_amd.ops", "pcm_amd.model", "pcm_amd.discriminator", "pcm_amd.mmdit"):
        m = sys.modules.get(mod)
        if m is not None and hasattr(m, "BF16"):
            m.BF16 = _DTYPES[name]
