"""Static resource limits of the gfx950 code objects (CPU-only: reads the kernel metadata notes hipcc wrote into lib/obj/*.o).

The host emulator checks DYNAMIC LDS requests at launch time (tests/emu/hip_emu.cpp); what it cannot see is what the compiler
allocated: static __shared__ arrays (host statics in the emulator), registers and scratch.  This test closes that gap: a kernel whose
static LDS exceeds a CU, or that would not launch with 64 KB of dynamic LDS on top of its static part, fails HERE and not on the
GPU box (round 1: pytest -m gpu went red on an LDS over-subscription no CPU test could see)."""
import glob
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "phased-consistency-model_amd", "pcm_amd", "lib", "obj")
LLVM = "/opt/rocm/lib/llvm/bin"
LDS_PER_CU = 160 * 1024


def kernel_metadata():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "phased-consistency-model_amd"))
    from pcm_amd import build
    build.build()
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for o in sorted(glob.glob(os.path.join(OBJ, "*.o"))):
            fat, co = os.path.join(td, "f.bin"), os.path.join(td, "f.co")
            r = subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", o], capture_output=True)
            if r.returncode != 0:
                continue      # host-only object (runtime.o)
            subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
            notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
            for blk in notes.split("  - .agpr_count:")[1:]:
                f = {k: int(v) for k, v in re.findall(r"\.(\w+):\s+(\d+)\s*$", "  .agpr_count:" + blk, flags=re.M)}
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                f["file"] = os.path.basename(o)
                out[name] = f
    return out


@pytest.fixture(scope="module")
def meta():
    if not os.path.exists(f"{LLVM}/llvm-readelf"):
        pytest.skip("no llvm-readelf")
    return kernel_metadata()


def test_every_kernel_fits_a_gfx950_cu(meta):
    assert len(meta) > 80, len(meta)
    bad = []
    for name, f in meta.items():
        lds, vg, ag = f["group_segment_fixed_size"], f["vgpr_count"], f["agpr_count"]
        if lds > LDS_PER_CU:      # a static segment may use the whole CU (MI355X_MICROARCH.md: 163 840 B static launches)
            bad.append(f"{name}: static LDS {lds} B > 160 KiB")
        if lds % 16 and any(k in name for k in DYN):
            bad.append(f"{name}: static LDS {lds} B not a multiple of 16 (misaligns the dynamic region: cdna guide G17)")
        if vg + ag > 512:
            bad.append(f"{name}: {vg}+{ag} registers")
        if f["max_flat_workgroup_size"] > 1024:
            bad.append(f"{name}: workgroup {f['max_flat_workgroup_size']}")
    assert not bad, "\n".join(bad)


# kernels that are launched with dynamic LDS: their (static + largest dynamic request) must fit the 160 KiB of a CU.
# The dynamic sizes restate the launchers' formulas at their maxima.
DYN = {
    "pcm_gemm8p_kernel": 2 * (256 + 64 * 5) * 128,      # pcm_gemm8p_lds_bytes(5)
    "pcm_gemm_kernel": 2 * (256 + 128) * 128,           # 256x128 tile
    "pcm_gemm_n64_kernel": 64 * 32 * 10 * 2,
    "gn_apply_kernel": 4 * 2560 * 4,                    # widest channel count of SDXL / SD1.5 (concatenated 2560)
}


def test_dynamic_lds_requests_fit(meta):
    seen = set()
    for name, f in meta.items():
        for key, dyn in DYN.items():
            if key in name:
                seen.add(key)
                assert f["group_segment_fixed_size"] + dyn <= LDS_PER_CU, (name, f["group_segment_fixed_size"], dyn)
    assert seen == set(DYN), seen


def test_hot_kernels_do_not_spill(meta):
    """scratch traffic in a hot loop is a >2x loss (cdna guide rule 20): the GEMM / attention / norm kernels must be spill-free"""
    hot = ("pcm_gemm8p_kernel", "pcm_gemm_kernel", "pcm_gemm_n64", "attn_fwd", "attn_bwd", "gn_stats", "gn_apply", "ln_fwd", "ln_bwd", "pcm_wgrad")
    report = {n: (f["vgpr_spill_count"], f["private_segment_fixed_size"]) for n, f in meta.items()
              if any(h in n for h in hot) and (f["vgpr_spill_count"] or f["private_segment_fixed_size"])}
    # known, measured exception: the 256x320 gemm8p variant spills in its segment-switch path
    unexpected = {n: v for n, v in report.items() if "pcm_gemm8p_kernelILi3" not in n}
    assert not unexpected, unexpected
