"""The 16-bit activation / weight format of this process: "bf16" (default; what BASELINE.json's configs and bench.py use) or "fp16".

The reference selects it with ``--mixed_precision`` (train_pcm_lora_sd15.py:1034 hands it to accelerate; every recipe in
train_pcm_lora_sd15.sh passes fp16, BASELINE.json's configs bf16).  Here the format is a BUILD variant of the kernel library
(csrc/pcm_common.h: one block of conversions + the two MFMA builtins), so ``set_precision("fp16")`` (1) loads lib/libpcm_hip_f16.so
instead of lib/libpcm_hip.so and (2) makes ``act_dtype()`` -- what every host module asks AT THE POINT OF USE for "the library's 16-bit
dtype" (no module holds it as a constant any more: round 6; ``ops.BF16`` etc. are module ``__getattr__`` lookups of the same function) --
return torch.float16.  Call it BEFORE building UNetWeights / LoraState / Distiller objects: packed operand buffers are allocated in the
current format.  fp16 needs loss scaling in the backward; trainer.Distiller does that with device-side GradScaler state.
"""
import torch

from . import capi

_NAME = "bf16"
_DTYPES = {"bf16": torch.bfloat16, "fp16": torch.float16}
_LIBS = {}        # format name -> the capi.Lib last validated for it (set_precision / register_lib); format_scope switches between them


def precision():
    return _NAME


def act_dtype():
    return _DTYPES[_NAME]


def set_precision(name, lib=None, tools=False):
    """name: "bf16" | "fp16" (there is no fp32-storage build).  ``lib``: tests pass an emulator build of the matching variant;
    ``tools``: the TOOLS build of the variant (pcm_debug_* hooks; tools/ and hook-using tests only).
    Nothing is switched unless the library loads and reports the requested format: a missing / mismatched file leaves the process in
    its previous precision."""
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
    _LIBS[name] = lib
    _activate(name, lib)


def _activate(name, lib):
    global _NAME
    capi.set_lib(lib)
    _NAME = name      # act_dtype() is looked up at every use site: nothing to rebind, a module imported later sees the same answer


def register_lib(name, lib):
    """the library ``format_scope(name)`` switches to (tests: an emulator build of that variant; default: the product library of the
    format, loaded on first use).  Validated like set_precision; does not switch anything."""
    if name not in _DTYPES:
        raise ValueError("precision must be 'bf16' or 'fp16', got %r" % (name,))
    if lib.act_dtype != (1 if name == "fp16" else 0):
        raise RuntimeError("%s was built for %s, not %s" % (lib.path, "fp16" if lib.act_dtype else "bf16", name))
    _LIBS[name] = lib


class format_scope:
    """``with format_scope("fp16"): ...`` -- run the enclosed host code (operand packing, a forward pass) against the OTHER 16-bit build of the
    kernel library, then switch back.  Both builds live in one process (distinct shared objects; a pointer handed to one is just memory
    to the other, fp32 tensors cross freely, 16-bit tensors must stay on their side -- ops.Seg checks the dtype).  This is how the
    reference's teacher pass is reproduced under --mixed_precision=bf16: its ``torch.autocast("cuda")`` (train_pcm_lora_sd15.py:1218)
    names no dtype and therefore runs the frozen teacher in IEEE half while the student runs in bfloat16 (trainer.Distiller
    ``teacher_weights``).  Switching is two attribute writes (the library handle and the format name); under hipGraph capture it happens at capture time only."""

    def __init__(self, name):
        if name not in _DTYPES:
            raise ValueError("precision must be 'bf16' or 'fp16', got %r" % (name,))
        self.name = name

    def __enter__(self):
        self.prev = (_NAME, capi.lib())
        if self.name != _NAME:
            lib = _LIBS.get(self.name)
            if lib is None:
                lib = capi.Lib(capi.F16_LIB if self.name == "fp16" else capi.DEFAULT_LIB)     # raises when the file is missing
                register_lib(self.name, lib)
            _LIBS.setdefault(self.prev[0], self.prev[1])
            _activate(self.name, lib)
        return self

    def __exit__(self, *exc):
        if self.prev[0] != _NAME:
            _activate(*self.prev)
        return False
