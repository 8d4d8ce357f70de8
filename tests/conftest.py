import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests/` on a machine without a GPU skips the gpu-marked tests instead of failing them with
    RTMI_NO_DEVICE; `-m gpu` always runs them (and fails loudly when there is no device: there is no CPU fallback)."""
    import os
    if "gpu" in (config.getoption("-m") or "") or os.path.exists("/dev/kfd"):
        return
    skip = pytest.mark.skip(reason="no GPU visible (/dev/kfd absent); run with -m gpu on an MI355X")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import _oracle
    _oracle.lib()
    return _oracle
