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
