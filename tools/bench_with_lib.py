"""Run bench.py against ANOTHER build of the kernel library (A/B of whole steps between two commits on one box):
    python tools/bench_with_lib.py tools/probes/libpcm_base.so [bench.py arguments]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
sys.path.insert(0, ROOT)
from pcm_amd import capi  # noqa: E402

capi.set_lib(capi.Lib(os.path.abspath(sys.argv[1])))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
