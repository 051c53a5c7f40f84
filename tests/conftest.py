import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The oracle's C restatement is OpenMP code.  The tests make thousands of SMALL oracle calls; with one spinning
# thread per hardware thread of a 256-thread host each parallel region costs ~0.2 s (measured on the GPU box), i.e.
# minutes per test module.  A modest passive team keeps the calls at their ~1 ms compute time.  (Set before
# libgomp is loaded; results do not depend on the team size.  bench.py's cpu_baseline leg is not affected.)
os.environ.setdefault("OMP_NUM_THREADS", "16")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _gpu_available():
    try:
        from xugrid_amd import _lib

        return _lib.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with -m gpu; when they are selected on a box without a GPU they
    # must FAIL loudly (no silent skip), so nothing is done here on purpose.
    return


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))

    return load


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build()
    return O


@pytest.fixture
def xr_option():
    """``xr_option(name, value)`` sets a run-time option of the library (include/xugrid_amd.h: xr_set_option) for the rest of the
    test; ``value=None`` puts back what the option held when the test first touched it; everything is restored at teardown.
    Strings are the words the environment variables of rounds 1-5 took ("free", "csr", "device") or numerals."""
    from xugrid_amd import engine

    words = {"free": 1, "csr": 2, "device": 1}
    initial = {}

    def set_option(name, value):
        if name not in initial:
            initial[name] = engine.get_option(name)
        if value is None:
            value = initial[name]
        elif isinstance(value, str):
            value = words[value] if value in words else int(value)
        engine.set_option(name, int(value))

    yield set_option
    for name, value in initial.items():
        engine.set_option(name, value)


@pytest.fixture(scope="session")
def hip():
    """The HIP engine bound to device 0.  Fails (does not skip) without a device."""
    import xugrid_amd
    from xugrid_amd import engine

    engine.init(0)
    return xugrid_amd


def canon(q, s, a):
    o = np.lexsort((s, q))
    return q[o], s[o], a[o]


def same_or_nan(a, b):
    return (a == b) | (np.isnan(a) & np.isnan(b))
