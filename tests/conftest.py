import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_addoption(parser):
    parser.addoption("--lib-variant", default=None,
                     help="development: run the suite against lib/libcvd_hip_<name>.so (e.g. `det`, the deterministic build) "
                          "instead of the product library")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    variant = config.getoption("--lib-variant")
    if variant:
        import torch  # (first, as the tests do: the process then holds ONE HIP runtime -- torch's)
        if torch.cuda.is_available():
            torch.cuda.init()
        from robust_cvd_amd import api, build as b
        if variant == "det":
            b.build_deterministic()
        api.load_library(variant=variant)  # (must be the first load of the library in the process)


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


def _try_build(what, fn):
    """Builds one native piece; on a machine without its toolchain the failure is remembered, not raised: tests that need the
    piece fail at their own import / load with the real error, pure-Python tests (sharding, synth, formats) still run."""
    try:
        fn()
    except (FileNotFoundError, OSError, RuntimeError, Exception) as e:  # noqa: BLE001 (subprocess / toolchain errors)
        _BUILD_ERRORS[what] = e


_BUILD_ERRORS = {}


@pytest.fixture(scope="session", autouse=True)
def _build_native():
    """Both native pieces are built in-tree before any test (seconds when up to date)."""
    from robust_cvd_amd import build as b
    _try_build("libcvd_hip", b.build)
    _try_build("lib_python", b.build_lib_python)
    from oracle import oracle as o
    _try_build("oracle", o.build)
    if _BUILD_ERRORS:
        import warnings
        warnings.warn("native build(s) failed, dependent tests will fail on load: " +
                      "; ".join(f"{k}: {v}" for k, v in _BUILD_ERRORS.items()))
