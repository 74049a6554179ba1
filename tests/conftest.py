import ctypes
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emu")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def emu_lib():
    """The product sources compiled against the HIP emulator (kernel-logic checks on CPU)."""
    import build_emu
    from orbhip import _lib
    return _lib.bind(ctypes.CDLL(build_emu.build()))


@pytest.fixture(scope="session")
def hip_lib():
    """The real product library (GPU tier)."""
    from orbhip import _lib
    return _lib.load()
