import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Builds the product libraries and the oracle once per session (hipcc cross-compiles without a GPU)."""
    import libjxl_amd
    import jxlo
    libjxl_amd.build()
    jxlo.build()
    return libjxl_amd
