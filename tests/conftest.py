import importlib
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def vra():
    """the product package (ctypes harness over libvr_core.so)"""
    pkg = importlib.import_module("volume-renderer_amd")
    if not pkg.LIB_PATH.exists():
        pkg.build_library()
    pkg.load_library()
    return pkg


@pytest.fixture(scope="session")
def oracle():
    import oracle as o  # test infrastructure

    return o
