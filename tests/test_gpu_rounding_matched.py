"""GPU run of the rounding-point-matched parity cases (tests/rounding_matched_cases.py) at the real SD1.5 widths: block level asserted at
the north-star 1e-3, end to end against the measured bf16-storage floor.  Writes gpurun_out/rounding_matched_sd15.json."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

KEYS = ("noise_pred", "cond_teacher_output", "target_noise_pred", "x_prev", "model_pred", "target")


@pytest.fixture(autouse=True)
def _hip():
    from pcm_amd import capi
    capi.set_lib(None)
    capi.lib()
    assert torch.cuda.is_available()


def _kw():
    return dict(block_out_channels=(320, 640, 1280, 1280), cross_attention_dim=768, heads=8, norm_num_groups=32)


@pytest.mark.parametrize("level,H", [(0, 32), (2, 16)])     # head dims 40 / 160
def test_blocks_vs_rounding_matched_oracle(level, H):
    import rounding_matched_cases as R
    rep = {}
    R.case_blocks("cuda", _kw(), 2, H, 77, level=level, report=rep)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rep, open("gpurun_out/rounding_matched_blocks_level%d.json" % level, "w"), indent=1)


def test_sd15_step_within_the_bf16_storage_floor():
    import rounding_matched_cases as R
    rep = {}
    R.case_step_floor("cuda", _kw(), 2, 64, 768, index=[13, 37], report=rep, with_fp32=False, golden_name="sd15_matched_m2_bs2")
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rep, open("gpurun_out/rounding_matched_sd15.json", "w"), indent=1)
    for k in KEYS:
        assert rep["hip_vs_matched"][k] <= 1.3 * rep["floor_matched_fp64_vs_fp32"][k] + 2e-4, (k, rep["hip_vs_matched"][k], rep["floor_matched_fp64_vs_fp32"][k])
    # the loss against the MATCHED oracle: the north-star 1e-3 (round-3 measurement on MI355X: 6.5e-4, with the oracle's own fp64-vs-fp32
    # floor at 3.1e-4; against the plain fp32 oracle both the HIP path and the matched oracle sit at 5.5e-3 / 6.1e-3 -- the bf16 storage
    # itself moves the loss by that much, in the same direction for both)
    assert rep["loss"]["hip_vs_matched"] <= max(3.0 * rep["loss"]["floor"], 1e-3), rep["loss"]
