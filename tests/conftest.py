import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

# The stream generator and the oracle are OpenMP code. A GPU box shows every core of its host (256) but gives the job a
# share of about 16: a team of 256 spinning threads per parallel region made every small case cost 2 - 3 s there
# (scripts/r03_test_overhead.py). Set before libgomp loads.
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(16, len(os.sched_getaffinity(0))))))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")


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
