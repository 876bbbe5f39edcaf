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


def set_precision(name, lib=None, tools=False):
    """name: "bf16" | "fp16" (there is no fp32-storage build).  ``lib``: tests pass an emulator build of the matching variant;
    ``tools``: the TOOLS build of the variant (pcm_debug_* hooks; tools/ and hook-using tests only).
    Nothing is switched unless the library loads and reports the requested format: a missing / mismatched file leaves the process in
    its previous precision."""
    global _NAME
    if name not in _DTYPES:
        raise ValueError("precision must be 'bf16' or 'fp16', got %r" % (name,))
    want = 1 if name == "fp16" else 0
    path = None
    if lib is None:
        path = (capi.TOOLS_F16_LIB if tools else capi.F16_LIB) if name == "fp16" else (capi.TOOLS_LIB if tools else capi.DEFAULT_LIB)
        lib = capi.Lib(path)                                                       # raises when the file is missing: nothing switched yet
    if lib.act_dtype != want:
        raise RuntimeError("%s was built for %s, not %s" % (lib.path, "fp16" if lib.act_dtype else "bf16", name))
    if path is not None:
        capi._LIB_PATH = path                                                      # capi.set_lib(None) / lib() keep loading this variant
    capi.set_lib(lib)
    _NAME = name
    # every host module that holds the "library's 16-bit dtype" constant (BF16) -- found by attribute, not by a hard-coded list
    for mod_name, m in list(sys.modules.items()):
        if mod_name.startswith("pcm_amd.") and m is not None and isinstance(getattr(m, "BF16", None), torch.dtype):
            m.BF16 = _DTYPES[name]
