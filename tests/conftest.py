import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fedd_lib():
    """The in-tree HIP library; built on demand (hipcc cross-compiles without a GPU)."""
    from feddlib_amd import build, capi
    if not os.path.exists(capi.LIB_PATH):
        build.build(verbose=False)
    return capi
