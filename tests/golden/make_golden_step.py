"""Writes the oracle fixtures tests/golden/step_*.safetensors that the ``-m gpu`` parity tests load (tests/golden_fixture.py).

Run in the BUILD CONTAINER (CPU only; needs the emulator library tests/emu for the host-side operand packing and /root/repo/oracle):

    python tests/golden/make_golden_step.py                 # every fixture, ~1.5 h on 8 cores
    python tests/golden/make_golden_step.py --only sd15_c2_m4_bs16 sd15_curve20_bs2
    python tests/golden/make_golden_step.py --list

Each fixture is the dictionary returned by a ``ref_*`` builder (tests/step_golden_cases.py, tests/adv_cases.py): the oracle
(oracle/pcm_step.py, oracle/unet_sd15.py, oracle/mmdit_sd3.py) evaluated on seeded CPU weights, LoRA factors and inputs.  Nothing under
/root/reference is read here: the oracle's own pinning against the reference source is tests/golden/make_golden.py +
tests/test_oracle_pinning.py."""
import argparse
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, os.path.join(ROOT, "phased-consistency-model_amd"), TESTS):
    if p not in sys.path:
        sys.path.insert(0, p)


def registry():
    import adv_cases as A
    import step_golden_cases as S
    from pcm_amd.discriminator import ADAPTER_DIMS
    reg = {
        S.step_name(0.0): lambda: S.ref_sd15_step(0.0),
        S.step_name(0.02): lambda: S.ref_sd15_step(0.02),
        "sd15_matched_m2_bs2": S.ref_sd15_matched,
        "sd15_c2_m4_bs16": S.ref_c2,
        "sd15_curve20_bs2": S.ref_curve20,
        "sdxl_fullsize_one_sample": S.ref_sdxl_one_sample,
        "sd3_fullsize_one_sample": S.ref_sd3_one_sample,
        "sdxl_fullsize_step_one_sample": S.ref_sdxl_step_fullsize,
        "sd3_fullsize_step_one_sample": S.ref_sd3_step_fullsize,
    }
    for gs in (0, 1):
        reg["sd15_adv_c3_bs2_step%d" % gs] = (lambda gs=gs: A.ref_adv_c3(S.SD15_KW, ADAPTER_DIMS, 2, 64, 77, 768, gs, nh=4, index=[30, 12]))
    return reg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*")
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--missing", action="store_true", help="only fixtures whose file does not exist yet")
    a = ap.parse_args()
    import golden_fixture as G
    reg = registry()
    if a.list:
        for n in reg:
            print(n, "(present)" if os.path.exists(G.path(n)) else "(missing)")
        return
    for name, fn in reg.items():
        if a.only and name not in a.only:
            continue
        if a.missing and os.path.exists(G.path(name)):
            continue
        t0 = time.time()
        G.save(name, fn())
        print("wrote %s  (%.0f s, %.2f MB)" % (G.path(name), time.time() - t0, os.path.getsize(G.path(name)) / 1e6), flush=True)


if __name__ == "__main__":
    main()
