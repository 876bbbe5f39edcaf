"""TEST INFRASTRUCTURE: load the host-emulation build of the csrc kernels (tests/emu)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))
_cached = {}


def emu_lib(variant="bf16"):
    """variant "f16": the IEEE-half build (-DPCM_ACT_F16) of the same sources"""
    if variant not in _cached:
        import build_emu
        from pcm_amd import capi
        _cached[variant] = capi.Lib(build_emu.build(variant=variant))
    return _cached[variant]
