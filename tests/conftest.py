import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "phased-consistency-model_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (``-m "not gpu"``) is mostly the single-threaded wave64 host emulator stepping whole networks: one process takes about an
    hour, the machine's cores take ~15 minutes.  When the caller asked for the CPU suite and gave no ``-n`` (and pytest-xdist is there), run it
    on min(7, cores - 1) workers (what fits the build container's 64 GiB); PCM_TEST_WORKERS=N overrides, 0 keeps one process.  The ``-m gpu``
    suite is never parallelised: its tests time kernels and share one device."""
    opt = config.option
    if os.environ.get("PYTEST_XDIST_WORKER") or hasattr(config, "workerinput"):      # (a worker: it must not spawn workers of its own)
        return None
    if getattr(opt, "markexpr", "") != "not gpu" or getattr(opt, "numprocesses", 0) is not None or not config.pluginmanager.hasplugin("xdist"):
        return None
    if getattr(opt, "collectonly", False) or getattr(opt, "usepdb", False):
        return None
    n = int(os.environ.get("PCM_TEST_WORKERS", min(7, max(1, (os.cpu_count() or 2) - 1))))
    if n > 1:
        opt.numprocesses = n          # (pytest-xdist's own pytest_cmdline_main, called after this one, turns it into n popen workers, --dist load)
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")


@pytest.fixture(scope="session", autouse=True)
def _cap_cpu_threads_on_the_gpu_box():
    """The live oracle evaluations left in the -m gpu suite are narrow configs (thousands of tiny CPU ops); on the GPU box torch defaults to
    128 threads of a shared 256-core host, where a wide pool only adds hand-shakes and -- with busy neighbours -- stalls on its slowest
    thread: the same tests took 9 s on one box and 103 s on another (profiles/r04_k_pytest_gpu_final_tree.txt vs r04_z_pytest_gpu_final_tree.txt).
    16 threads are as fast on a quiet box and do not degrade on a busy one.  (No effect in the 8-core build container.)"""
    import torch
    if torch.cuda.is_available():
        torch.set_num_threads(min(torch.get_num_threads(), 16))
    yield


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
