"""TEST INFRASTRUCTURE: load the host-emulation build of the csrc kernels (tests/emu)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))
_cached = None


def emu_lib():
    global _cached
    if _cached is None:
        import build_emu
        from pcm_amd import capi
        _cached = capi.Lib(build_emu.build())
    return _cached
