"""TEST INFRASTRUCTURE ONLY: compile the csrc/*.hip kernels for the HOST with tests/emu/hip_emu.h
(wave64 fiber emulator) into tests/emu/libpcm_emu.so so index math can be checked without a GPU."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "phased-consistency-model_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
# PCM_EMU_ASAN=1: AddressSanitizer build (separate objects / library).  Run the kernel tests under it with
#   LD_PRELOAD=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 \
#   PCM_EMU_ASAN=1 python -m pytest tests/test_emu_kernels.py
# torch's CPU allocator goes through the intercepted malloc, so every tensor gets redzones: out-of-bounds reads / writes of a kernel
# (tails, halos, clamped loads) that the GPU would silently tolerate or fault on are reported with the kernel's source line.
ASAN = os.environ.get("PCM_EMU_ASAN") == "1"
# PCM_EMU_UBSAN=1: -fsanitize=alignment,bounds (LD_PRELOAD libclang_rt.ubsan_standalone-x86_64.so, UBSAN_OPTIONS=halt_on_error=1):
# under-aligned vector accesses (hip_emu.h gives float4 / uint4 their 16-byte alignment) and out-of-range constant-size array indices
UBSAN = os.environ.get("PCM_EMU_UBSAN") == "1"
OUT = os.path.join(HERE, "libpcm_emu_asan.so" if ASAN else ("libpcm_emu_ubsan.so" if UBSAN else "libpcm_emu.so"))


def build(force=False, variant="bf16"):
    """variant "f16": the IEEE-half build of the same sources (-DPCM_ACT_F16, csrc/pcm_common.h) -> libpcm_emu_f16.so"""
    f16 = variant == "f16"
    OUT = globals()["OUT"].replace(".so", "_f16.so") if f16 else globals()["OUT"]
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + [os.path.join(HERE, "hip_emu.cpp")]
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "hip_emu.h"),
                                                           os.path.join(ROOT, "include", "pcm_hip.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in srcs:
        o = os.path.join(HERE, "build", os.path.basename(s) + (".f16" if f16 else "") + (".asan.o" if ASAN else (".ubsan.o" if UBSAN else ".o")))
        objs.append(o)
        cmd = [CLANG, "-x", "c++", "-DPCM_HOST_EMU", "-I", HERE, "-O1" if ASAN else "-O2", "-std=c++17", "-fPIC",
               "-Wno-unused-value", "-Wno-deprecated-declarations", "-Wno-psabi", "-c", s, "-o", o]
        if f16:
            cmd[3:3] = ["-DPCM_ACT_F16"]
        if ASAN:
            cmd[1:1] = ["-fsanitize=address", "-fno-omit-frame-pointer", "-g", "-shared-libasan"]
        elif UBSAN:
            cmd[1:1] = ["-fsanitize=alignment,bounds", "-fno-sanitize-recover=alignment,bounds", "-g", "-shared-libsan"]
        procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("emu compile failed: " + s)
    san = ["-fsanitize=address", "-shared-libasan"] if ASAN else (["-fsanitize=alignment,bounds", "-shared-libsan"] if UBSAN else [])
    subprocess.check_call([CLANG, "-shared", "-o", OUT] + san + objs)
    return OUT


if __name__ == "__main__":
    print(build(force="-f" in sys.argv))
