"""Build the csrc kernels with -DPCM_ABLATE into tools/probes/libpcm_ablate.so (timing ablations for tools/gemm8p_ablate.py;
the product library is built without the flag and contains none of the ablation branches)."""
import glob, os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), 'phased-consistency-model_amd'))
from pcm_amd import build as _B  # noqa: E402
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "phased-consistency-model_amd", "csrc")
OUT = os.path.join(HERE, "libpcm_ablate.so")
OBJ = os.path.join(HERE, "obj_ablate")
def build():
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h"))
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o"); objs.append(o)
        if not os.path.exists(o) or any(os.path.getmtime(d) > os.path.getmtime(o) for d in [s] + hdrs):
            extra = _B.extra_flags(s)      # the in-source `// pcm-build-flags:` marker, as pcm_amd/build.py reads it
            jobs.append(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-DPCM_ABLATE"] + extra + ["-c", s, "-o", o])
    with ThreadPoolExecutor(8) as ex:
        list(ex.map(lambda c: subprocess.check_call(c), jobs))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    return OUT
if __name__ == "__main__":
    print(build())
