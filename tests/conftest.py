import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        from robust_cvd_amd import api
        s = api.Solver(0)
        s.close()
        return True
    except Exception:
        return False


@pytest.fixture(scope="session")
def have_gpu():
    return _has_gpu()


@pytest.fixture(scope="session", autouse=True)
def _build_native():
    """Both native pieces are built in-tree before any test (seconds when up to date)."""
    from robust_cvd_amd import build as b
    b.build()
    b.build_lib_python()
    from oracle import oracle as o
    o.build()
