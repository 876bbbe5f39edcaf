"""Build a VARIANT of the kernel library for A/B timing: tools/probes/libpcm_<name>.so from a csrc directory (default: the tree's) with
extra compiler flags.  usage: build_variant.py <name> [--csrc DIR] [extra hipcc flags ...]   (outputs are git-ignored, travel with gpurun)"""
import glob, os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), 'phased-consistency-model_amd'))
from pcm_amd import build as _B  # noqa: E402
ROOT = os.path.dirname(os.path.dirname(HERE))
def build(name, csrc, extra):
    out, obj = os.path.join(HERE, "libpcm_%s.so" % name), os.path.join(HERE, "obj_%s" % name)
    os.makedirs(obj, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(csrc, "*.hip")))
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(obj, os.path.basename(s)[:-4] + ".o"); objs.append(o)
        per = _B.extra_flags(s)      # the in-source `// pcm-build-flags:` marker, as pcm_amd/build.py reads it
        jobs.append(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-I" + os.path.join(ROOT, "include")] + extra + per + ["-c", s, "-o", o])
    with ThreadPoolExecutor(8) as ex:
        list(ex.map(lambda c: subprocess.check_call(c), jobs))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out
if __name__ == "__main__":
    a = sys.argv[1:]
    name = a.pop(0)
    csrc = os.path.join(ROOT, "phased-consistency-model_amd", "csrc")
    if a and a[0] == "--csrc":
        a.pop(0); csrc = a.pop(0)
    print(build(name, csrc, a))
