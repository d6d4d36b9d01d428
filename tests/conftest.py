import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session", autouse=True)
def _load_checker_first():
    """Load (and if needed build) the CPU checker before any test initialises the GPU: a
    process that has touched the GPU must not fork/exec on the GPU pool."""
    from oracle import oracle
    oracle.lib()
    oracle.ref_lib()


@pytest.fixture(scope="session")
def orc(_load_checker_first):
    """The CPU checker (test infrastructure, never the product path)."""
    from oracle import oracle
    return oracle


def has_gpu():
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        return False


def rel_err(a, b):
    """max relative error with NaN positions required identical and inf == inf."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape
    assert np.array_equal(np.isnan(a), np.isnan(b)), "NaN masks differ"
    m = ~np.isnan(a)
    a, b = a[m], b[m]
    inf = np.isinf(a) | np.isinf(b)
    assert np.array_equal(a[inf], b[inf]), "infinities differ"
    a, b = a[~inf], b[~inf]
    if a.size == 0:
        return 0.0
    den = np.maximum(np.abs(b), 1e-300)
    return float(np.max(np.abs(a - b) / den))
