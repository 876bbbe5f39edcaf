"""TEST INFRASTRUCTURE ONLY: compile the csrc/*.hip kernels for the HOST with tests/emu/hip_emu.h
(wave64 fiber emulator) into tests/emu/libpcm_emu.so so index math can be checked without a GPU."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "phased-consistency-model_amd", "csrc")
OUT = os.path.join(HERE, "libpcm_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def build(force=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + [os.path.join(HERE, "hip_emu.cpp")]
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "hip_emu.h"),
                                                           os.path.join(ROOT, "include", "pcm_hip.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in srcs:
        o = os.path.join(HERE, "build", os.path.basename(s) + ".o")
        objs.append(o)
        cmd = [CLANG, "-x", "c++", "-DPCM_HOST_EMU", "-I", HERE, "-O2", "-std=c++17", "-fPIC",
               "-Wno-unused-value", "-Wno-deprecated-declarations", "-Wno-psabi", "-c", s, "-o", o]
        procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("emu compile failed: " + s)
    subprocess.check_call([CLANG, "-shared", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build(force="-f" in sys.argv))
