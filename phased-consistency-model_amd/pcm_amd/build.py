"""Build libpcm_hip.so for gfx950 with hipcc (cross-compiles without a GPU).  In-tree output so
the .so travels with the repo snapshot to the GPU box."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libpcm_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
# per-file extras are declared IN the source: a line `// pcm-build-flags: <flags>` in the first 40 lines (attention*.hip: keep MFMA
# results in VGPRs -- the softmax reads every accumulator, and the AGPR form costs a v_accvgpr_read/write per value per tile)
MARKER = "// pcm-build-flags:"


def extra_flags(src):
    out = []
    with open(src) as f:
        for i, line in enumerate(f):
            if i >= 40:
                break
            if line.startswith(MARKER):
                out += line[len(MARKER):].split()
    return out


def _digest(paths, extra=()):
    """sha256 over the CONTENTS of ``paths`` (sorted by file name) and the strings in ``extra``"""
    import hashlib
    h = hashlib.sha256()
    for p in sorted(paths, key=os.path.basename):
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    for e in extra:
        h.update(str(e).encode() + b"\0")
    return h.hexdigest()


def source_files():
    """everything a kernel library is a function of: csrc/*.hip, csrc/*.h, include/pcm_hip.h"""
    return (sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + sorted(glob.glob(os.path.join(CSRC, "*.h")))
            + [os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "pcm_hip.h")])


def source_id():
    """16 hex digits identifying the kernel sources of this tree (the same for all variants): what pcm_build_id() of a library built from them
    starts with, what tools/pmc_step_table.py stamps its tables with, what capi.lib() compares a loaded library against"""
    return _digest(source_files())[:16]


def _stale(out, stamp):
    """content-keyed staleness (round 6; was mtime): ``out`` is current iff ``out + '.hash'`` holds ``stamp``.  A snapshot that carries binaries
    older than its sources -- whatever the file times say -- is rebuilt, and touching a source without changing it is not."""
    try:
        with open(out + ".hash") as f:
            return (not os.path.exists(out)) or f.read().strip() != stamp
    except OSError:
        return True


def _mark(out, stamp):
    with open(out + ".hash", "w") as f:
        f.write(stamp + "\n")


# variants: the same sources, another 16-bit activation / weight format (csrc/pcm_common.h).  "bf16" is the product default and what
# bench.py measures; "f16" (-DPCM_ACT_F16 -> lib/libpcm_hip_f16.so) serves --mixed_precision=fp16 and the fp32-oracle loss validation.
VARIANTS = {"bf16": ("libpcm_hip.so", "obj", []), "f16": ("libpcm_hip_f16.so", "obj_f16", ["-DPCM_ACT_F16"]),
            "tools": ("libpcm_hip_tools.so", "obj_tools", ["-DPCM_TOOLS"]), "tools_f16": ("libpcm_hip_tools_f16.so", "obj_tools_f16", ["-DPCM_TOOLS", "-DPCM_ACT_F16"])}


def lib_path(variant="bf16"):
    return os.path.join(LIBDIR, VARIANTS[variant][0])


def build(force=False, verbose=False, variant="bf16"):
    libname, objname, vflags = VARIANTS[variant]
    lib = os.path.join(LIBDIR, libname)
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, objname)
    os.makedirs(objdir, exist_ok=True)
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "pcm_hip.h")]
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    sid = source_id()
    build_id = sid + "-" + variant
    jobs = []
    objs = []
    stamps = {}
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        flags = FLAGS + vflags + extra_flags(s)
        if os.path.basename(s) == "runtime.hip":       # pcm_build_id(): the identity of ALL sources is compiled into this one object
            flags = flags + ['-DPCM_BUILD_ID="%s"' % build_id]
        stamps[o] = _digest([s] + hdrs, flags)
        if force or _stale(o, stamps[o]):
            jobs.append(([HIPCC] + flags + ["-c", s, "-o", o], o))

    def run(job):
        cmd, out = job
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if out is not None:
            _mark(out, stamps[out])
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    lib_stamp = _digest([], [stamps[o] for o in objs])
    if force or jobs or _stale(lib, lib_stamp):
        run(([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, None))
        _mark(lib, lib_stamp)
    return lib


def build_all(force=False, verbose=False):
    return [build(force, verbose, v) for v in VARIANTS]


if __name__ == "__main__":
    for v in VARIANTS:
        if v == "bf16" or "--all" in sys.argv or ("--" + v) in sys.argv:
            print(build(force="-f" in sys.argv, verbose=True, variant=v))
