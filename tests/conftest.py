import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "phased-consistency-model_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")


@pytest.fixture(scope="session")
def golden():
    from safetensors.torch import load_file
    return load_file(os.path.join(ROOT, "tests", "golden", "pcm_math_golden.safetensors"))


@pytest.fixture(scope="session")
def golden_fm():
    """flow-matching (SD3 variant) PCM math + samplers, tests/golden/make_golden_sd3.py"""
    from safetensors.torch import load_file
    return load_file(os.path.join(ROOT, "tests", "golden", "pcm_fm_golden.safetensors"))
