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


@pytest.fixture(autouse=True)
def _poison_uninitialised_memory(monkeypatch):
    """PCM_POISON_EMPTY=1: every ``torch.empty`` / ``empty_like`` buffer the host code hands to a kernel starts as NaN (floats) or a
    large sentinel (ints), so a kernel that READS an output / workspace element before writing it -- partially written split-K slabs,
    a statistics arena that was not cleared, a tail the epilogue skips -- turns the result into NaN instead of passing by luck."""
    if os.environ.get("PCM_POISON_EMPTY") != "1":
        yield
        return
    import torch
    real_empty, real_empty_like = torch.empty, torch.empty_like

    def poison(t):
        if t.is_floating_point():
            t.fill_(float("nan"))
        elif t.dtype in (torch.int32, torch.int64):
            t.fill_(0x3f3f3f3f)
        elif t.dtype == torch.uint8:
            t.fill_(0xff)          # byte workspaces: 0xffff bf16 = NaN, 0xffffffff fp32 = NaN
        return t
    monkeypatch.setattr(torch, "empty", lambda *a, **k: poison(real_empty(*a, **k)))
    monkeypatch.setattr(torch, "empty_like", lambda *a, **k: poison(real_empty_like(*a, **k)))
    yield
